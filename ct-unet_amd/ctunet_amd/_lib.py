"""ctypes binding of libctunet_hip.so (the C ABI declared in include/ctunet_hip.h).

There is no fallback: if the library is missing or a symbol cannot be resolved the import of
the product path fails with a clear message.  Build with ``python __graft_entry__.py`` or
``make -C ct-unet_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CTUNET_HIP_LIB: load a differently built libctunet_hip.so (development A/B runs); there is no other fallback
LIB_PATH = os.environ.get("CTUNET_HIP_LIB") or os.path.join(_HERE, "libctunet_hip.so")

P = C.c_void_p          # device pointer / stream
I = C.c_int
L = C.c_int64
F = C.c_float
D = C.c_double
Z = C.c_size_t

class PackJob(C.Structure):
    """ctu_pack_job of include/ctunet_hip.h."""
    _fields_ = [("w", C.c_void_p), ("wp", C.c_void_p), ("cinv", C.c_void_p), ("kind", C.c_int), ("Co", C.c_int),
                ("Ci", C.c_int), ("k", C.c_int), ("rin_p", C.c_int), ("nout_p", C.c_int), ("mode", C.c_int),
                ("layout", C.c_int)]


class BnTail(C.Structure):
    """ctu_bn_tail of include/ctunet_hip.h."""
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("num_batches_tracked", C.c_void_p), ("counter", C.c_void_p), ("count", C.c_double),
                ("momentum", C.c_float), ("eps", C.c_float), ("C", C.c_int), ("n_updates", C.c_int)]


class BnBwdTail(C.Structure):
    """ctu_bn_bwd_tail of include/ctunet_hip.h."""
    _fields_ = [("gamma", C.c_void_p), ("invstd", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("coef", C.c_void_p), ("mean", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p), ("counter", C.c_void_p), ("count", C.c_double),
                ("momentum", C.c_float), ("eps", C.c_float), ("C", C.c_int)]


# name -> (restype, argtypes); mirrors include/ctunet_hip.h one to one
SIGNATURES = {
    "ctu_last_error": (C.c_char_p, []),
    "ctu_abi_version": (I, []),
    "ctu_arch": (C.c_char_p, []),
    "ctu_ncdhw_to_ndhwc": (I, [P, P, I, I, I, I, I, I, I, P]),
    "ctu_ndhwc_to_ncdhw": (I, [P, P, I, I, I, I, I, I, P]),
    "ctu_conv3d_layout": (I, [I, I, I]),
    "ctu_conv3d_fwd_kernel_name": (C.c_char_p, [I, I, I, I, I, I, I]),
    "ctu_conv3d_wgrad_kernel_name": (C.c_char_p, [I, I, I, I]),
    "ctu_conv3d_packed_floats": (Z, [I, I, I, I]),
    "ctu_conv3d_num_blocks": (I, [I, I, I, I, I, I, I]),
    "ctu_pack_conv3d_weight": (I, [P, P, I, I, I, P, I, I, I, I, P]),
    "ctu_pack_batch": (I, [P, I, P]),
    "ctu_conv3d_fwd": (I, [P, I, I, P, P, I, P, P, I, P, I, I, P, I, I, I, I, I, I, P, P]),
    "ctu_conv3d_wgrad_ws_floats": (Z, [I, I, I, I, I, I, I]),
    "ctu_conv3d_wgrad": (I, [P, I, I, P, P, I, P, I, I, P, P, I, I, P, P, I, I, I, I, I, P]),
    "ctu_conv3d_wgrad_bn_supported": (I, [I, I, I, I, I, I, I]),
    "ctu_conv3d_wgrad_bn": (I, [P, I, I, P, P, I, P, I, I, P, P, P, P, P, P, I, I, P, P, I, I, I, I, I, P]),
    "ctu_conv3d_first_supported": (I, [I, I, I, I]),
    "ctu_conv3d_first_num_blocks": (I, [I, I, I, I]),
    "ctu_conv3d_first_fwd": (I, [P, I, P, P, I, P, I, I, P, I, I, I, I, P, P]),
    "ctu_conv3d_first_bwd_data": (I, [P, I, P, I, I, P, I, I, I, I, P]),
    "ctu_conv3d_first_wgrad_ws_floats": (Z, [I, I, I, I, I]),
    "ctu_conv3d_first_wgrad": (I, [P, I, P, I, P, I, P, I, I, I, I, P]),
    "ctu_conv3d_first_wgrad_bn": (I, [P, I, P, I, P, P, P, P, P, P, I, P, I, I, I, I, P]),
    "ctu_bn_finalize": (I, [P, I, I, I, D, P, P, P, P, F, F, I, P, P, P, P, P, P]),
    "ctu_bn_eval_affine": (I, [P, P, P, P, F, I, I, P, P, P]),
    "ctu_bn_bwd_num_blocks": (I, [L]),
    "ctu_bn_relu_bwd_reduce": (I, [P, I, P, I, I, P, P, P, P, L, P, P, P]),
    "ctu_bn_bwd_finalize": (I, [P, I, I, I, D, P, P, P, P, P, P, P, P, F, F, P, P]),
    "ctu_bn_relu_bwd_apply": (I, [P, I, P, I, I, P, P, P, P, P, L, P]),
    "ctu_maxpool2_fwd": (I, [P, I, I, P, P, I, P, I, I, I, I, I, P]),
    "ctu_maxpool2_bwd": (I, [P, I, I, P, P, I, P, I, P, I, I, I, I, I, I, P]),
    "ctu_maxpool2_bwd_bn_num_blocks": (I, [I, I, I, I, I]),
    "ctu_maxpool2_bwd_bn": (I, [P, I, I, P, P, P, P, P, I, P, I, I, I, I, I, I, P, P, P]),
    "ctu_convt_packed_floats": (Z, [I, I]),
    "ctu_pack_convt_weight": (I, [P, P, I, I, P, I, I, I, P]),
    "ctu_convt2_fwd": (I, [P, I, I, P, P, I, P, P, I, P, I, I, I, I, I, I, P]),
    "ctu_convt2_bwd_data": (I, [P, I, I, P, P, I, I, I, I, I, I, P]),
    "ctu_convt2_wgrad_ws_floats": (Z, [I, I, I, I, I, I]),
    "ctu_convt2_wgrad": (I, [P, I, I, P, P, I, P, I, I, P, P, I, I, P, P, I, I, I, I, P]),
    "ctu_head_fwd": (I, [P, I, I, P, P, I, P, P, P, I, I, I, I, P, P, I, L, P]),
    "ctu_head_bwd_ws_floats": (Z, [I, L, I, I]),
    "ctu_head_bwd": (I, [P, I, I, P, P, I, P, P, P, I, I, I, I, P, P, P, I, P, P, P, I, L, P]),
    "ctu_head_bwd_num_blocks": (I, [I, L]),
    "ctu_head_bwd_bn": (I, [P, I, I, P, P, I, P, P, P, I, I, I, I, P, P, P, I, P, P, P, I, L, P, P, I, P, P, P]),
    "ctu_loss_ws_floats": (Z, [I, L]),
    "ctu_loss_fwd": (I, [P, P, I, L, F, F, I, P, P, P]),
    "ctu_loss_bwd": (I, [P, P, I, L, F, F, I, P, P, P, P, I, P]),
    "ctu_skip_add": (I, [P, I, P, P, I, P, I, P, P, I, P, I, I, L, P]),
    "ctu_upconv_fused_supported": (I, [I, I, I, I, I, I]),
    "ctu_upconv_fused_packed_floats": (Z, [I, I]),
    "ctu_upconv_fused_num_blocks": (I, [I, I, I, I, I]),
    "ctu_upconv_fused_pack_ws_floats": (Z, [I, I]),
    "ctu_upconv_fused_pack": (I, [P, P, P, I, I, P, I, I, P, P, P, P]),
    "ctu_upconv_fused_fwd": (I, [P, I, I, P, P, I, P, P, P, I, I, P, I, I, I, I, P, P]),
    "ctu_upconv_fused_wgrad_ws_floats": (Z, [I, I, I, I, I, I]),
    "ctu_upconv_fused_wgrad": (I, [P, I, I, P, P, I, P, I, I, P, P, I, I, I, I, P]),
    "ctu_upconv_fused_wgrad_bn_supported": (I, [I, I, I, I, I, I]),
    "ctu_upconv_fused_wgrad_bn": (I, [P, I, I, P, P, I, P, I, I, P, P, P, P, P, P, P, I, I, I, I, P]),
    "ctu_upconv_fused_project_ws_floats": (Z, [I, L]),
    "ctu_upconv_fused_project": (I, [P, P, I, I, I, I, I, I, P, P, P, I, I, I, P, P, P, P, P]),
    "ctu_upconv_fused_bwd_packed_floats": (Z, [I, I]),
    "ctu_upconv_fused_pack_bwd": (I, [P, I, I, P, P]),
    "ctu_upconv_fused_bwd_data": (I, [P, I, I, P, P, I, I, I, I, I, I, P]),
    "ctu_hard_segm": (I, [P, I, I, L, P, P]),
    "ctu_one_hot": (I, [P, I, I, L, P, P]),
    "ctu_hard_dice_ws_doubles": (Z, [I]),
    "ctu_hard_dice_counts": (I, [P, P, I, I, L, P, P, P]),
    "ctu_hausdorff_ws_bytes": (Z, [I, I, I, I, I]),
    "ctu_hausdorff": (I, [P, P, I, I, I, I, I, P, P, P]),
    "ctu_extract_patches": (I, [P, P, I, I, I, I, I, I, I, I, P, P]),
    "ctu_stitch_patches": (I, [P, P, I, I, I, I, I, I, I, I, P, P]),
    "ctu_lp_upconv_fused_supported": (I, [I, I, I, I, I, I]),
    "ctu_lp_upconv_fused_packed_elems": (Z, [I]),
    "ctu_lp_upconv_fused_num_blocks": (I, [I, I, I, I]),
    "ctu_lp_upconv_fused_pack": (I, [I, P, I, P, P]),
    "ctu_lp_upconv_fused_fwd": (I, [I, P, I, I, P, P, I, P, P, P, I, P, I, I, I, I, P]),
    "ctu_lp_upconv_fused_wgrad_ws_floats": (Z, [I, I, I, I, I]),
    "ctu_lp_upconv_fused_wgrad": (I, [I, P, I, I, P, P, I, P, I, P, P, I, I, I, I, P]),
    "ctu_lp_upconv_fused_wgrad_bn_supported": (I, [I, I, I, I, I]),
    "ctu_lp_upconv_fused_wgrad_bn": (I, [I, P, I, I, P, P, I, P, I, P, P, P, P, P, P, P, I, I, I, I, P]),
    "ctu_lp_conv3d_wgrad_bn_supported": (I, [I, I, I, I, I, I, I]),
    "ctu_lp_conv3d_wgrad_bn": (I, [I, P, I, I, P, P, I, P, I, I, P, P, P, P, P, P, I, I, P, P, I, I, I, I, I, P]),
    "ctu_lp_upconv_fused_project": (I, [I, P, P, I, I, I, I, I, I, P, P, P, I, I, I, P, P, P, P, P]),
    "ctu_lp_upconv_fused_bwd_data": (I, [I, P, I, P, P, I, I, I, I, I, I, P]),
    "ctu_lp_conv3d_layout": (I, [I, I, I, I]),
    "ctu_lp_conv3d_packed_elems": (Z, [I, I, I]),
    "ctu_lp_conv3d_num_blocks": (I, [I, I, I, I, I, I, I, I]),
    "ctu_lp_pack_conv3d_weight": (I, [I, P, P, I, I, I, P, I, I, I, I, P]),
    "ctu_lp_pack_batch": (I, [I, P, I, P]),
    "ctu_lp_conv3d_fwd": (I, [I, P, I, I, P, P, I, P, P, I, P, I, I, P, I, I, I, I, I, I, P, P]),
    "ctu_lp_conv3d_wgrad_ws_floats": (Z, [I, I, I, I, I, I, I]),
    "ctu_lp_conv3d_wgrad": (I, [I, P, I, I, P, P, I, P, I, I, P, I, I, P, P, I, I, I, I, I, P]),
    "ctu_lp_conv3d_first_fwd": (I, [I, P, I, P, P, I, P, I, I, P, I, I, I, I, P, P]),
    "ctu_lp_conv3d_first_bwd_data": (I, [I, P, I, P, I, I, P, I, I, I, I, P]),
    "ctu_lp_conv3d_first_bwd_data_pair_supported": (I, [I, I]),
    "ctu_lp_conv3d_fwd_kernel_name": (C.c_char_p, [I, I, I, I, I, I, I, I]),
    "ctu_lp_conv3d_wgrad_kernel_name": (C.c_char_p, [I, I, I, I, I, I]),
    "ctu_lp_conv3d_first_bwd_data_pair": (I, [I, P, I, P, I, P, I, I, I, I, P]),
    "ctu_lp_conv3d_first_wgrad": (I, [I, P, I, P, I, P, I, P, I, I, I, I, P]),
    "ctu_lp_convt_packed_elems": (Z, [I, I, I]),
    "ctu_lp_pack_convt_weight": (I, [I, P, P, I, I, P, I, I, I, P]),
    "ctu_lp_convt2_fwd": (I, [I, P, I, I, P, P, I, P, P, I, P, I, I, I, I, I, I, P]),
    "ctu_lp_convt2_bwd_data": (I, [I, P, I, I, P, P, I, I, I, I, I, I, P]),
    "ctu_lp_convt2_wgrad_ws_floats": (Z, [I, I, I, I, I, I]),
    "ctu_lp_convt2_wgrad": (I, [I, P, I, I, P, P, I, P, I, I, P, I, I, P, P, I, I, I, I, P]),
    "ctu_lp_ncdhw_to_ndhwc": (I, [I, P, P, I, I, I, I, I, I, I, P]),
    "ctu_lp_ndhwc_to_ncdhw": (I, [I, P, P, I, I, I, I, I, I, P]),
    "ctu_lp_bn_relu_bwd_reduce": (I, [I, P, I, P, I, I, P, P, P, P, L, P, P, P]),
    "ctu_lp_bn_relu_bwd_apply": (I, [I, P, I, P, I, I, P, P, P, P, P, L, P]),
    "ctu_lp_maxpool2_fwd": (I, [I, P, I, I, P, P, I, P, I, I, I, I, I, P]),
    "ctu_lp_maxpool2_bwd": (I, [I, P, I, I, P, P, I, P, I, P, I, I, I, I, I, I, P]),
    "ctu_lp_maxpool2_bwd_bn": (I, [I, P, I, I, P, P, P, P, P, I, P, I, I, I, I, I, I, P, P, P]),
    "ctu_lp_skip_add": (I, [I, P, I, P, P, I, P, I, P, P, I, P, I, I, L, P]),
    "ctu_lp_channel_sum": (I, [I, P, I, I, L, P, P, I, P]),
    "ctu_lp_head_fwd": (I, [I, P, I, I, P, P, I, P, P, P, I, I, I, I, P, P, I, L, P]),
    "ctu_lp_head_bwd_bn": (I, [I, P, I, I, P, P, I, P, P, P, I, I, I, I, P, P, P, I, P, P, P, I, L, P, P, I, P, P, F, P]),
    "ctu_scale_tensors": (I, [P, P, I, F, P, P]),
    "ctu_comm_available": (I, []),
    "ctu_comm_unique_id": (I, [P]),
    "ctu_comm_init": (I, [C.POINTER(C.c_void_p), I, I, P]),
    "ctu_comm_allreduce_f32": (I, [P, P, P, Z, I, P]),
    "ctu_comm_destroy": (I, [P]),
    "ctu_channel_sum_num_blocks": (I, [L]),
    "ctu_channel_sum": (I, [P, I, I, L, P, P, I, P]),
    "ctu_adam_amsgrad": (I, [P, P, I, P, D, D, D, D, D, I, P, P]),
}

_lib = None
ABI_VERSION = 7          # CTU_ABI_VERSION of the csrc/ this file mirrors (bumped on any signature change)


class CtuError(RuntimeError):
    """A libctunet_hip.so entry point returned a non-zero status."""


def load() -> C.CDLL:
    """Load the shared library once and attach the prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X kernels are not built. Run `python __graft_entry__.py` "
            "(or `make -C ct-unet_amd/csrc`). There is no CPU fallback for this path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ctu_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} is ABI v{lib.ctu_abi_version()}, this package binds v{ABI_VERSION}: "
                          "a stale build -- run `python __graft_entry__.py` (or `make -C ct-unet_amd/csrc`)")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().ctu_last_error().decode("utf-8", "replace")
        raise CtuError(f"{what} failed (status {status}): {msg}")
