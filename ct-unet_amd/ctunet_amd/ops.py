"""Thin tensor-level wrappers over the C ABI (include/ctunet_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation is a call into
libctunet_hip.so.  All functions require CUDA (ROCm) tensors and raise otherwise -- there is no
CPU path.

``CL`` describes a channels-last activation: a slice ``[c0, c0+cp)`` of the channels of a
contiguous ``[N, D, H, W, cs]`` fp32 buffer plus its lazy-BN transform (``scale``/``shift`` over
the slice, ``relu``): consumers see ``relu(raw * scale + shift)``.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


LP_CODE = {torch.bfloat16: 1, torch.float16: 2}          # CTU_BF16 / CTU_F16 of include/ctunet_hip.h
ACT_DTYPES = (torch.float32, torch.bfloat16, torch.float16)


def _need_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"ctunet_amd: {name} must live on the GPU (MI355X); this path has no CPU fallback")
    if t.dtype != torch.float32 and t.dtype != torch.int32:
        raise RuntimeError(f"ctunet_amd: {name} must be float32, got {t.dtype}")


def _ptr(t: Optional[torch.Tensor], off_elems: int = 0):
    if t is None:
        return None
    return t.data_ptr() + t.element_size() * off_elems


def _dual(code: int, name: str, *args) -> None:
    """ctu_<name>(args) for fp32 tensors, ctu_lp_<name>(dtype code, args) for 16-bit ones (same argument lists)."""
    lib = _lib.load()
    if code:
        _lib.check(getattr(lib, "ctu_lp_" + name)(code, *args), "lp_" + name)
    else:
        _lib.check(getattr(lib, "ctu_" + name)(*args), name)


def _tail_arg(tail):
    """ctypes argument of an optional ctu_bn_tail / ctu_bn_bwd_tail."""
    import ctypes
    return None if tail is None else ctypes.byref(tail)


def make_bn_tail(c: int, count, gamma, beta, rmean, rvar, momentum: float, eps: float, n_updates: int, vec4, nbt, counter):
    """ctu_bn_tail: the launch that writes the BatchNorm partial rows also finalizes them (no ctu_bn_finalize launch).
    vec4: [4, cp] rows receiving scale, shift, mean, invstd; counter: one zeroed int32 device word owned by the layer."""
    assert counter.dtype == torch.int32 and counter.is_cuda and counter.numel() == 1
    assert nbt is None or (nbt.dtype == torch.int64 and nbt.is_cuda and nbt.numel() == 1)
    t = _lib.BnTail()
    t.gamma, t.beta = gamma.data_ptr(), beta.data_ptr()
    t.running_mean, t.running_var = _ptr(rmean), _ptr(rvar)
    t.scale, t.shift, t.mean, t.invstd = (vec4[i].data_ptr() for i in range(4))
    t.num_batches_tracked = None if nbt is None else nbt.data_ptr()
    t.counter = counter.data_ptr()
    t.count, t.momentum, t.eps, t.C, t.n_updates = float(count), momentum, eps, c, n_updates
    return t


def make_bn_bwd_tail(c: int, count, gamma, vec4, dgb, coef, replay, counter):
    """ctu_bn_bwd_tail (replaces the ctu_bn_bwd_finalize launch); replay as in bn_relu_bwd."""
    assert counter is None or (counter.dtype == torch.int32 and counter.is_cuda and counter.numel() == 1)
    rm, rv, mom, eps, nbt = (tuple(replay) + (None,))[:5] if replay is not None else (None, None, 0.0, 0.0, None)
    assert nbt is None or (nbt.dtype == torch.int64 and nbt.is_cuda and nbt.numel() == 1)
    t = _lib.BnBwdTail()
    t.gamma, t.invstd, t.mean = gamma.data_ptr(), vec4[3].data_ptr(), vec4[2].data_ptr()
    t.dgamma, t.dbeta, t.coef = dgb[0].data_ptr(), dgb[1].data_ptr(), coef.data_ptr()
    t.running_mean, t.running_var = _ptr(rm), _ptr(rv)
    t.num_batches_tracked = None if nbt is None else nbt.data_ptr()
    t.counter = None if counter is None else counter.data_ptr()
    t.count, t.momentum, t.eps, t.C = float(count), mom, eps, c
    return t


def lp(t_or_dtype) -> int:
    """dtype code of the ctu_lp_* entry points for a 16-bit activation tensor / dtype; 0 for float32."""
    dt = t_or_dtype if isinstance(t_or_dtype, torch.dtype) else t_or_dtype.dtype
    return LP_CODE.get(dt, 0)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional HIP-event bracketing of individual launches (bench.py's roofline leg).

    Events are recorded on the stream the kernels are launched on (torch's current stream), so they
    time exactly one launch each.  ``records`` holds (tag, algorithmic_flops, algorithmic_bytes, start, end).
    """

    def __init__(self):
        self.records = []
        self.details = []
        self._empty = []           # event pairs around NOTHING, interleaved with the timed launches: the bracket's own cost

    def _overhead_ms(self) -> float:
        """Median elapsed time of the empty event pairs (recorded every 16th launch): what an event pair measures when
        there is no kernel between the two records (2-5 us on this stack) -- subtracted from every bracketed launch, so the
        per-launch figures agree with the kernel-trace durations of rocprofv3."""
        ts = sorted(a.elapsed_time(b) for a, b in self._empty)
        return ts[len(ts) // 2] if ts else 0.0

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, tag, flops, nbytes, start, detail=None):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.append((tag, flops, nbytes, start, ev))
        self.details.append(detail)
        if len(self.records) % 16 == 1:
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            b.record()
            self._empty.append((a, b))

    def by_layer(self):
        """{(tag, detail): dict(launches, total_ms, flops)} -- per-geometry split of ``summary`` (dev tool)."""
        out = {}
        ovh = self._overhead_ms()
        for (tag, fl, nb, a, b), det in zip(self.records, self.details):
            d = out.setdefault((tag, det), dict(launches=0, total_ms=0.0, flops=0.0))
            d["launches"] += 1
            d["total_ms"] += max(a.elapsed_time(b) - ovh, 1e-4)
            d["flops"] += fl
        return out

    def summary(self):
        """{tag: dict(launches, total_ms, avg_ms, flops, bytes)} -- call after a device synchronize.
        A launch site (tag, geometry) is represented by the MEDIAN of its timed launches: a one-off stall that lands between
        an event pair (an allocator hipMalloc, RCCL's lazy channel set-up in the first eager distributed step: 73 ms once,
        measured) would otherwise be booked on whatever kernel happened to be bracketed."""
        sites = {}
        ovh = self._overhead_ms()
        for (tag, fl, nb, a, b), det in zip(self.records, self.details):
            sites.setdefault((tag, det, fl, nb), []).append(max(a.elapsed_time(b) - ovh, 1e-4))
        out = {}
        for (tag, det, fl, nb), ts in sites.items():
            ts.sort()
            med = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
            d = out.setdefault(tag, dict(launches=0, total_ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += len(ts)
            d["total_ms"] += med * len(ts)
            d["flops"] += fl * len(ts)
            d["bytes"] += nb * len(ts)
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / d["launches"]
        return out


TIMER: Optional[KernelTimer] = None      # set by bench.py; None in normal operation
# 16-bit first-layer weight gradient: volumes with at least this many voxels go through the matrix pipe (conv_first_wgrad)
FIRST_WGRAD_MFMA_MIN_VOX = int(os.environ.get("CTUNET_FIRST_WGRAD_MFMA_MIN_VOX", "4000000"))


def _tile_tag(w: int) -> str:
    return "4_4_16" if w >= 16 else ("4_8_8" if w >= 8 else "4_4_4")


def _nt(nout_p: int) -> int:
    n16 = (nout_p + 15) // 16
    return 1 if n16 == 1 else (2 if n16 == 2 else 4)


@dataclass
class CL:
    buf: torch.Tensor                       # [N, D, H, W, cs] contiguous fp32
    c0: int = 0
    cp: int = 0                             # padded channels of this slice (multiple of 8)
    scale: Optional[torch.Tensor] = None    # [cp] (already sliced) or None
    shift: Optional[torch.Tensor] = None
    relu: bool = False

    def __post_init__(self):
        if self.cp == 0:
            self.cp = self.buf.shape[-1] - self.c0
        assert self.buf.dim() == 5 and self.buf.is_contiguous() and self.buf.dtype in ACT_DTYPES
        assert self.cp % 8 == 0 and self.c0 % 4 == 0 and self.c0 + self.cp <= self.buf.shape[-1]
        # 16-bit tensors: slices start on 16-byte boundaries and the voxel stride keeps them there
        assert self.buf.dtype == torch.float32 or (self.c0 % 8 == 0 and self.buf.shape[-1] % 8 == 0)

    @property
    def dtype(self) -> torch.dtype:
        return self.buf.dtype

    @property
    def lp(self) -> int:
        return LP_CODE.get(self.buf.dtype, 0)

    @property
    def cs(self) -> int:
        return self.buf.shape[-1]

    @property
    def dims(self) -> Tuple[int, int, int, int]:
        n, d, h, w, _ = self.buf.shape
        return n, d, h, w

    @property
    def nvox(self) -> int:
        n, d, h, w = self.dims
        return n * d * h * w

    @property
    def ptr(self):
        return _ptr(self.buf, self.c0)

    def slice(self, c0: int, cp: int) -> "CL":
        sc = None if self.scale is None else self.scale[c0:c0 + cp]
        sh = None if self.shift is None else self.shift[c0:c0 + cp]
        return CL(self.buf, self.c0 + c0, cp, sc, sh, self.relu)

    def raw(self) -> "CL":
        return CL(self.buf, self.c0, self.cp)

    def with_xf(self, scale, shift, relu=True) -> "CL":
        return CL(self.buf, self.c0, self.cp, scale, shift, relu)


def new_cl(n, d, h, w, cs, device, zero=False, dtype=torch.float32) -> torch.Tensor:
    f = torch.zeros if zero else torch.empty
    return f((n, d, h, w, cs), dtype=dtype, device=device)


# ---------------------------------------------------------------------------- layout
def ncdhw_to_cl(x: torch.Tensor, cp: Optional[int] = None, dtype=torch.float32) -> CL:
    _need_cuda(x, "input")
    x = x.contiguous()
    n, c, d, h, w = x.shape
    cp = cp or pad8(c)
    out = new_cl(n, d, h, w, cp, x.device, dtype=dtype)
    _dual(lp(dtype), "ncdhw_to_ndhwc", x.data_ptr(), out.data_ptr(), n, c, d, h, w, cp, cp, _stream())
    return CL(out, 0, cp)


def cl_to_ncdhw(a: CL, c: int) -> torch.Tensor:
    n, d, h, w = a.dims
    out = torch.empty((n, c, d, h, w), dtype=torch.float32, device=a.buf.device)
    _dual(a.lp, "ndhwc_to_ncdhw", a.ptr, out.data_ptr(), n, c, d, h, w, a.cs, _stream())
    return out


# ---------------------------------------------------------------------------- conv3d
def conv_layout(k: int, nout_p: int, w: int, dtype=torch.float32, rin_p: int = 0) -> int:
    """Packed-weight layout the forward kernel wants for this output width / volume width (16-bit path: depends on the
    padded input channel count too)."""
    if lp(dtype):
        return _lib.load().ctu_lp_conv3d_layout(k, rin_p, nout_p, w)
    return _lib.load().ctu_conv3d_layout(k, nout_p, w)


def pack_conv_w(w: torch.Tensor, cinv: Optional[torch.Tensor], rin_p: int, nout_p: int, mode: int,
                layout: int = 0) -> torch.Tensor:
    """cinv: int32 map padded input-channel position -> logical channel (-1 = padding), None = identity."""
    _need_cuda(w, "conv weight")
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    lib = _lib.load()
    n = lib.ctu_conv3d_packed_floats(k, rin_p, nout_p, layout)
    wp = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.check(lib.ctu_pack_conv3d_weight(w.contiguous().data_ptr(), wp.data_ptr(), co, ci, k, _ptr(cinv), rin_p,
                                          nout_p, mode, layout, _stream()), "pack_conv3d_weight")
    return wp


def packed_floats(kind: str, k: int, rin_p: int, nout_p: int, layout: int) -> int:
    lib = _lib.load()
    return lib.ctu_conv3d_packed_floats(k, rin_p, nout_p, layout) if kind == "conv" else \
        lib.ctu_convt_packed_floats(rin_p, nout_p)


def pack_batch(jobs) -> None:
    """jobs: list of (kind 'conv'|'convt', w, wp, cinv, rin_p, nout_p, mode, layout): every job in ONE launch."""
    if not jobs:
        return
    arr = (_lib.PackJob * len(jobs))()
    for a, (kind, w, wp, cinv, rin_p, nout_p, mode, layout) in zip(arr, jobs):
        _need_cuda(w, "weight")
        assert w.is_contiguous()
        a.w, a.wp, a.cinv = w.data_ptr(), wp.data_ptr(), (None if cinv is None else cinv.data_ptr())
        if kind == "conv":
            a.kind, a.Co, a.Ci, a.k = 0, w.shape[0], w.shape[1], w.shape[2]
        else:
            a.kind, a.Ci, a.Co, a.k = 1, w.shape[0], w.shape[1], 2
        a.rin_p, a.nout_p, a.mode, a.layout = rin_p, nout_p, mode, layout
    _lib.check(_lib.load().ctu_pack_batch(arr, len(jobs), _stream()), "pack_batch")


def pack_batch_lp(jobs, dtype: torch.dtype) -> None:
    """jobs as in pack_batch: every 16-bit weight copy in ONE launch."""
    if not jobs:
        return
    arr = (_lib.PackJob * len(jobs))()
    for a, (kind, w, wp, cinv, rin_p, nout_p, mode, layout) in zip(arr, jobs):
        _need_cuda(w, "weight")
        assert w.is_contiguous() and wp.dtype == dtype
        a.w, a.wp, a.cinv = w.data_ptr(), wp.data_ptr(), (None if cinv is None else cinv.data_ptr())
        if kind == "conv":
            a.kind, a.Co, a.Ci, a.k = 0, w.shape[0], w.shape[1], w.shape[2]
        else:
            a.kind, a.Ci, a.Co, a.k = 1, w.shape[0], w.shape[1], 2
        a.rin_p, a.nout_p, a.mode, a.layout = rin_p, nout_p, mode, (layout if kind == "conv" else 0)
    _lib.check(_lib.load().ctu_lp_pack_batch(LP_CODE[dtype], arr, len(jobs), _stream()), "lp_pack_batch")


def conv_num_blocks(dims, nout_p: int, layout: int = 0, k: int = 3, dtype=torch.float32, rin_p: int = 32) -> int:
    """Rows of the BN partial-sum buffer a conv3d_fwd call with this geometry writes (16-bit path: depends on the padded
    input channel count too, pass rin_p)."""
    n, d, h, w = dims
    if lp(dtype):
        return _lib.load().ctu_lp_conv3d_num_blocks(n, d, h, w, k, rin_p, nout_p, layout)
    return _lib.load().ctu_conv3d_num_blocks(n, d, h, w, k, nout_p, layout)


def conv3d_fwd(x: CL, wp: torch.Tensor, bias: Optional[torch.Tensor], out: CL, k: int,
               stats: Optional[torch.Tensor] = None, algo_ch: Optional[Tuple[int, int]] = None, layout: int = 0,
               tail=None) -> None:
    """algo_ch = (logical Cin, logical Cout) -- only used to count algorithmic FLOPs when timing.
    tail: make_bn_tail(...) -- the launch finalizes the BatchNorm of its output itself."""
    n, d, h, w = x.dims
    assert out.dims == x.dims
    lib = _lib.load()
    if x.lp:
        assert out.dtype == x.dtype and wp.dtype == x.dtype, (x.dtype, out.dtype, wp.dtype)
        t0 = TIMER.begin() if TIMER is not None else None
        _lib.check(lib.ctu_lp_conv3d_fwd(x.lp, x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), wp.data_ptr(),
                                         _ptr(bias), 0 if bias is None else bias.numel(), out.ptr, out.cs, out.cp,
                                         _ptr(stats), n, d, h, w, k, layout, _tail_arg(tail), _stream()), "lp_conv3d_fwd")
        if t0 is not None:
            ci, co = algo_ch if algo_ch is not None else (x.cp, out.cp)
            vox = n * d * h * w
            name = lib.ctu_lp_conv3d_fwd_kernel_name(n, d, h, w, k, x.cp, out.cp, layout).decode()
            TIMER.end(f"{name}<{'bf16' if x.lp == 1 else 'f16'}, {k}>", 2.0 * ci * co * k ** 3 * vox, 2.0 * vox * (ci + co), t0,
                      (w, x.cp, out.cp))
        return
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.check(lib.ctu_conv3d_fwd(x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), wp.data_ptr(),
                                  _ptr(bias), 0 if bias is None else bias.numel(), out.ptr, out.cs, out.cp, _ptr(stats),
                                  n, d, h, w, k, layout, _tail_arg(tail), _stream()), "conv3d_fwd")
    if t0 is not None:
        ci, co = algo_ch if algo_ch is not None else (x.cp, out.cp)
        vox = n * d * h * w
        TIMER.end(lib.ctu_conv3d_fwd_kernel_name(n, d, h, w, k, out.cp, layout).decode(), 2.0 * ci * co * k ** 3 * vox,
                  4.0 * vox * (ci + co), t0, (w, x.cp, out.cp))


def conv3d_wgrad(x: CL, g: CL, co: int, ci: int, k: int, cinv: Optional[torch.Tensor], ws: torch.Tensor,
                 want_bias: bool) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    n, d, h, w = x.dims
    lib = _lib.load()
    if x.lp:
        assert g.dtype == x.dtype
        need = lib.ctu_lp_conv3d_wgrad_ws_floats(n, d, h, w, k, x.cp, g.cp)
        assert ws.numel() >= need, (ws.numel(), need)
        dw = torch.empty((co, ci, k, k, k), dtype=torch.float32, device=x.buf.device)
        t0 = TIMER.begin() if TIMER is not None else None
        _lib.check(lib.ctu_lp_conv3d_wgrad(x.lp, x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), g.ptr, g.cs,
                                           g.cp, dw.data_ptr(), co, ci, _ptr(cinv), ws.data_ptr(), n, d, h, w, k, _stream()),
                   "lp_conv3d_wgrad")
        if t0 is not None:
            vox = n * d * h * w
            name = lib.ctu_lp_conv3d_wgrad_kernel_name(d, h, w, k, x.cp, g.cp).decode()
            TIMER.end(f"{name}<{'bf16' if x.lp == 1 else 'f16'}, {k}> (+slab reduce)",
                      2.0 * ci * co * k ** 3 * vox, 2.0 * vox * (ci + co), t0, (w, x.cp, g.cp))
        return dw, (channel_sum(g, co) if want_bias else None)
    need = lib.ctu_conv3d_wgrad_ws_floats(n, d, h, w, k, x.cp, g.cp)
    assert ws.numel() >= need, (ws.numel(), need)
    dw = torch.empty((co, ci, k, k, k), dtype=torch.float32, device=x.buf.device)
    db = torch.empty(co, dtype=torch.float32, device=x.buf.device) if want_bias else None
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.check(lib.ctu_conv3d_wgrad(x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), g.ptr, g.cs, g.cp,
                                    dw.data_ptr(), _ptr(db), co, ci, _ptr(cinv), ws.data_ptr(), n, d, h, w, k,
                                    _stream()), "conv3d_wgrad")
    if t0 is not None:
        vox = n * d * h * w
        TIMER.end(lib.ctu_conv3d_wgrad_kernel_name(w, k, x.cp, g.cp).decode() + " (+slab reduce)",
                  2.0 * ci * co * k ** 3 * vox, 4.0 * vox * (ci + co), t0, (w, x.cp, g.cp))
    return dw, db


def conv3d_wgrad_bn_supported(dims, k: int, cin_p: int, cout_p: int, dtype=torch.float32) -> bool:
    """Can conv3d_wgrad_bn take this layer (k = 3, full boxes [and channel tiles for fp32])?"""
    n, d, h, w = dims
    if lp(dtype):
        return bool(_lib.load().ctu_lp_conv3d_wgrad_bn_supported(n, d, h, w, k, cin_p, cout_p))
    return bool(_lib.load().ctu_conv3d_wgrad_bn_supported(n, d, h, w, k, cin_p, cout_p))


def conv3d_wgrad_bn(x: CL, ga: CL, y: CL, vec: torch.Tensor, coef: torch.Tensor, gy: CL, co: int, ci: int, k: int,
                    cinv: Optional[torch.Tensor], ws: torch.Tensor) -> torch.Tensor:
    """Weight gradient with the layer's BatchNorm + ReLU backward folded in: ga = gradient w.r.t. the ACTIVATED output, y =
    the raw conv output, vec = its [4, cp] BatchNorm vectors, coef = bn_relu_bwd(lazy=True)'s rows; gy (same buffer shape
    and channel offset as ga) receives the raw-output gradient for the data-gradient kernel."""
    n, d, h, w = x.dims
    lib = _lib.load()
    assert ga.dims == x.dims and y.dims == x.dims and gy.dims == x.dims and ga.dtype == x.dtype == y.dtype == gy.dtype
    assert y.cs == ga.cs == gy.cs and y.cp == ga.cp == gy.cp and gy.buf.data_ptr() != ga.buf.data_ptr()
    if x.lp:
        need = lib.ctu_lp_conv3d_wgrad_ws_floats(n, d, h, w, k, x.cp, ga.cp)
        assert ws.numel() >= need, (ws.numel(), need)
        dw = torch.empty((co, ci, k, k, k), dtype=torch.float32, device=x.buf.device)
        t0 = TIMER.begin() if TIMER is not None else None
        _lib.check(lib.ctu_lp_conv3d_wgrad_bn(x.lp, x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), ga.ptr, ga.cs,
                                              ga.cp, y.ptr, vec[0].data_ptr(), vec[1].data_ptr(), coef.data_ptr(), gy.ptr,
                                              dw.data_ptr(), co, ci, _ptr(cinv), ws.data_ptr(), n, d, h, w, k, _stream()),
                   "lp_conv3d_wgrad_bn")
        if t0 is not None:
            vox = n * d * h * w
            name = lib.ctu_lp_conv3d_wgrad_kernel_name(d, h, w, k, x.cp, ga.cp).decode()
            TIMER.end(f"{name}<{'bf16' if x.lp == 1 else 'f16'}, {k}> (+slab reduce)",
                      2.0 * ci * co * k ** 3 * vox, 2.0 * vox * (ci + co), t0, (w, x.cp, ga.cp))
        return dw
    need = lib.ctu_conv3d_wgrad_ws_floats(n, d, h, w, k, x.cp, ga.cp)
    assert ws.numel() >= need, (ws.numel(), need)
    dw = torch.empty((co, ci, k, k, k), dtype=torch.float32, device=x.buf.device)
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.check(lib.ctu_conv3d_wgrad_bn(x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), ga.ptr, ga.cs, ga.cp,
                                       y.ptr, vec[0].data_ptr(), vec[1].data_ptr(), coef.data_ptr(), gy.ptr,
                                       dw.data_ptr(), co, ci, _ptr(cinv), ws.data_ptr(), n, d, h, w, k, _stream()),
               "conv3d_wgrad_bn")
    if t0 is not None:
        vox = n * d * h * w
        TIMER.end(lib.ctu_conv3d_wgrad_kernel_name(w, k, x.cp, ga.cp).decode() + " (+slab reduce)",
                  2.0 * ci * co * k ** 3 * vox, 4.0 * vox * (ci + co), t0, (w, x.cp, ga.cp))
    return dw


def conv3d_wgrad_ws(dims, k, cin_p, cout_p, dtype=torch.float32) -> int:
    n, d, h, w = dims
    if lp(dtype):
        return _lib.load().ctu_lp_conv3d_wgrad_ws_floats(n, d, h, w, k, cin_p, cout_p)
    return _lib.load().ctu_conv3d_wgrad_ws_floats(n, d, h, w, k, cin_p, cout_p)


def pack_conv_w_lp(w: torch.Tensor, cinv: Optional[torch.Tensor], rin_p: int, nout_p: int, mode: int, dtype: torch.dtype,
                   into: Optional[torch.Tensor] = None, layout: int = 0) -> torch.Tensor:
    """16-bit MFMA-fragment-ordered copy of an fp32 master Conv3d weight (mode 0 forward, 1 data gradient; layout as
    conv_layout(..., dtype, rin_p) says for the launch that will read it)."""
    _need_cuda(w, "conv weight")
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    lib = _lib.load()
    n = lib.ctu_lp_conv3d_packed_elems(k, rin_p, nout_p)
    wp = into if into is not None else torch.empty(n, dtype=dtype, device=w.device)
    assert wp.numel() == n and wp.dtype == dtype
    _lib.check(lib.ctu_lp_pack_conv3d_weight(LP_CODE[dtype], w.contiguous().data_ptr(), wp.data_ptr(), co, ci, k, _ptr(cinv),
                                             rin_p, nout_p, mode, layout, _stream()), "lp_pack_conv3d_weight")
    return wp


# ---------------------------------------------------------------------------- first layer (C_in <= 2)
def conv_first_supported(k: int, cin: int, nout_p: int, w: int) -> bool:
    return bool(_lib.load().ctu_conv3d_first_supported(k, cin, nout_p, w))


def conv_first_num_blocks(dims) -> int:
    n, d, h, w = dims
    return _lib.load().ctu_conv3d_first_num_blocks(n, d, h, w)


def conv_first_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: CL,
                   stats: Optional[torch.Tensor], tail=None) -> None:
    """x: NCDHW float32 on the GPU (read in place); w: torch Conv3d weight [Co, cin, 3, 3, 3]."""
    _need_cuda(x, "input")
    n, cin, d, h, w_ = x.shape
    lib = _lib.load()
    t0 = TIMER.begin() if TIMER is not None else None
    _dual(out.lp, "conv3d_first_fwd", x.data_ptr(), cin, w.data_ptr(), _ptr(bias), 0 if bias is None else bias.numel(),
          out.ptr, out.cs, w.shape[0], _ptr(stats), n, d, h, w_, _tail_arg(tail), _stream())
    if t0 is not None:
        vox = n * d * h * w_
        TIMER.end(f"first_fwd_kernel<{cin}>", 2.0 * cin * w.shape[0] * 27 * vox, 4.0 * vox * (cin + w.shape[0]), t0)


def conv_first_bwd_data(g: CL, w: torch.Tensor, cin: int) -> torch.Tensor:
    """dx (float32 NCDHW) of the first encoder convolution.  16-bit gradients of volumes at least 32 wide go through the
    matrix pipe (the 8 -> 8 pair-layout kernel with a float32-plane epilogue; the weight's 9 KB fragment copy is made here,
    one tiny launch), everything else through the direct kernel."""
    n, d, h, w_ = g.dims
    dx = torch.empty((n, cin, d, h, w_), dtype=torch.float32, device=g.buf.device)
    lib = _lib.load()
    t0 = TIMER.begin() if TIMER is not None else None
    pair = bool(g.lp and g.cp == 8 and lib.ctu_lp_conv3d_first_bwd_data_pair_supported(cin, w_))
    if pair:
        wp = pack_conv_w_lp(w, None, 8, 8, 1, g.dtype, None, 1)
        _lib.check(lib.ctu_lp_conv3d_first_bwd_data_pair(g.lp, g.ptr, g.cs, wp.data_ptr(), cin, dx.data_ptr(), n, d, h, w_,
                                                         _stream()), "lp_conv3d_first_bwd_data_pair")
    else:
        _dual(g.lp, "conv3d_first_bwd_data", g.ptr, g.cs, w.data_ptr(), cin, w.shape[0], dx.data_ptr(), n, d, h, w_, _stream())
    if t0 is not None:
        vox = n * d * h * w_
        tag = f"lp_conv_fwd_pair_kernel<{'bf16' if g.lp == 1 else 'f16'}, dx of the first layer>" if pair else f"first_bwd_data_kernel<{cin}>"
        TIMER.end(tag, 2.0 * cin * w.shape[0] * 27 * vox, (2.0 * w.shape[0] if g.lp else 4.0 * w.shape[0]) * vox + 4.0 * vox * cin, t0)
    return dx


def conv_first_wgrad(x: torch.Tensor, g: CL, co: int, ws: torch.Tensor) -> torch.Tensor:
    n, cin, d, h, w_ = x.shape
    lib = _lib.load()
    if (g.lp and g.cp == 8 and n * d * h * w_ >= FIRST_WGRAD_MFMA_MIN_VOX
            and lib.ctu_lp_conv3d_wgrad_kernel_name(d, h, w_, 3, 8, 8) == b"lp_wgrad8_kernel"):
        # large 16-bit volumes: an 8-channel 16-bit copy of the input (one streaming pass) + the 8 -> 8 matrix-pipe weight
        # gradient instead of the direct VALU kernel (192^3 x 2 channels: 223 -> 75 + 45 us; 256^3: 520 -> 196 + 100; no gain at
        # 128^3 x 1, hence the threshold).  The input enters this product rounded to 16 bits, like every other layer's.
        x8 = ncdhw_to_cl(x, 8, g.dtype)
        need = lib.ctu_lp_conv3d_wgrad_ws_floats(n, d, h, w_, 3, 8, 8)
        return conv3d_wgrad(x8, g, co, cin, 3, None, ws if ws.numel() >= need else torch.empty(need, device=x.device), False)[0]
    assert ws.numel() >= lib.ctu_conv3d_first_wgrad_ws_floats(n, d, h, w_, cin)
    dw = torch.empty((co, cin, 3, 3, 3), dtype=torch.float32, device=x.device)
    t0 = TIMER.begin() if TIMER is not None else None
    _dual(g.lp, "conv3d_first_wgrad", x.data_ptr(), cin, g.ptr, g.cs, dw.data_ptr(), co, ws.data_ptr(), n, d, h, w_, _stream())
    if t0 is not None:
        vox = n * d * h * w_
        TIMER.end(f"first_wgrad_kernel<{cin}> (+slab reduce)", 2.0 * cin * co * 27 * vox, 4.0 * vox * (cin + co), t0)
    return dw


def conv_first_wgrad_bn(x: torch.Tensor, ga: CL, y: CL, vec: torch.Tensor, coef: torch.Tensor, gy: CL, co: int,
                        ws: torch.Tensor) -> torch.Tensor:
    """conv_first_wgrad with the BatchNorm + ReLU backward folded in (see conv3d_wgrad_bn); fp32 only."""
    n, cin, d, h, w_ = x.shape
    lib = _lib.load()
    assert not ga.lp and y.cs == ga.cs == gy.cs and ga.cp == 8 and gy.buf.data_ptr() != ga.buf.data_ptr()
    assert ws.numel() >= lib.ctu_conv3d_first_wgrad_ws_floats(n, d, h, w_, cin)
    dw = torch.empty((co, cin, 3, 3, 3), dtype=torch.float32, device=x.device)
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.check(lib.ctu_conv3d_first_wgrad_bn(x.data_ptr(), cin, ga.ptr, ga.cs, y.ptr, vec[0].data_ptr(), vec[1].data_ptr(),
                                             coef.data_ptr(), gy.ptr, dw.data_ptr(), co, ws.data_ptr(), n, d, h, w_, _stream()),
               "conv3d_first_wgrad_bn")
    if t0 is not None:
        vox = n * d * h * w_
        TIMER.end(f"first_wgrad_kernel<{cin}> (+slab reduce)", 2.0 * cin * co * 27 * vox, 4.0 * vox * (cin + co), t0)
    return dw


def conv_first_wgrad_ws(dims, cin: int) -> int:
    n, d, h, w = dims
    return _lib.load().ctu_conv3d_first_wgrad_ws_floats(n, d, h, w, cin)


# ---------------------------------------------------------------------------- batch norm
def bn_finalize_into(stats, nblocks, c, cp, count, gamma, beta, rmean, rvar, momentum, eps, n_updates, vec4, nbt=None):
    """vec4: [4, cp] view (rows may be strided) receiving scale, shift, mean, invstd.
    nbt: the BatchNorm's num_batches_tracked (int64 device scalar), advanced by n_updates in the same launch."""
    lib = _lib.load()
    assert nbt is None or (nbt.dtype == torch.int64 and nbt.is_cuda and nbt.numel() == 1)
    _lib.check(lib.ctu_bn_finalize(stats.data_ptr(), nblocks, c, cp, float(count), gamma.data_ptr(), beta.data_ptr(),
                                   _ptr(rmean), _ptr(rvar), momentum, eps, n_updates, vec4[0].data_ptr(),
                                   vec4[1].data_ptr(), vec4[2].data_ptr(), vec4[3].data_ptr(),
                                   None if nbt is None else nbt.data_ptr(), _stream()), "bn_finalize")


def bn_finalize(stats, nblocks, c, cp, count, gamma, beta, rmean, rvar, momentum, eps, n_updates):
    vec = torch.empty((4, cp), dtype=torch.float32, device=stats.device)     # scale, shift, mean, invstd
    bn_finalize_into(stats, nblocks, c, cp, count, gamma, beta, rmean, rvar, momentum, eps, n_updates, vec)
    return vec


def bn_eval_affine_into(gamma, beta, rmean, rvar, eps, c, cp, vec):
    lib = _lib.load()
    _lib.check(lib.ctu_bn_eval_affine(gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), eps, c, cp,
                                      vec[0].data_ptr(), vec[1].data_ptr(), _stream()), "bn_eval_affine")


def bn_eval_affine(gamma, beta, rmean, rvar, eps, c, cp):
    vec = torch.empty((2, cp), dtype=torch.float32, device=gamma.device)
    bn_eval_affine_into(gamma, beta, rmean, rvar, eps, c, cp, vec)
    return vec


def bn_relu_bwd(y: CL, ga: CL, vec: torch.Tensor, gamma: torch.Tensor, c: int, partials: torch.Tensor,
                replay=None, pre_reduced: Optional[int] = None, counter: Optional[torch.Tensor] = None, finalized=None,
                lazy: bool = False):
    """In place: ga <- gradient w.r.t. the raw conv output y.  Returns (dgamma, dbeta).
    replay = (running_mean, running_var, momentum, eps[, num_batches_tracked]): also apply the running-stat update a
    second time (the one torch.utils.checkpoint's recompute performs in backward, models.py:232-255).
    counter: the reduction launch finalizes itself (ctu_bn_bwd_tail) instead of a ctu_bn_bwd_finalize launch.
    finalized = (dgb, coef): the kernel that produced ga already reduced AND finalized (maxpool_bwd / head_bwd with fin=).
    lazy: reduce and finalize only -- ga stays the gradient w.r.t. the ACTIVATED output and the weight-gradient kernel
    applies the backward while it stages ga (conv3d_wgrad_bn / upconv_fused_wgrad_bn); returns (dgamma, dbeta, coef)."""
    lib = _lib.load()
    nvox = y.nvox
    cp = y.cp
    sc, sh, mu, istd = (vec[i].data_ptr() for i in range(4))
    st = _stream()
    if finalized is not None:
        dgb, coef = finalized
        if lazy:
            return dgb[0], dgb[1], coef
        _dual(y.lp, "bn_relu_bwd_apply", y.ptr, y.cs, ga.ptr, ga.cs, cp, sc, sh, mu, istd, coef.data_ptr(), nvox, st)
        return dgb[0], dgb[1]
    nb = lib.ctu_bn_bwd_num_blocks(nvox) if pre_reduced is None else pre_reduced
    assert partials.numel() >= nb * 2 * cp
    dgb = torch.empty((2, c), dtype=torch.float32, device=y.buf.device)
    coef = torch.empty((5, cp), dtype=torch.float32, device=y.buf.device)
    tail = make_bn_bwd_tail(c, nvox, gamma, vec, dgb, coef, replay, counter) if (counter is not None and pre_reduced is None) else None
    if pre_reduced is None:       # (else: the kernel that produced ga wrote the nb reduction rows, maxpool_bwd(bn=...))
        assert ga.dtype == y.dtype
        _dual(y.lp, "bn_relu_bwd_reduce", y.ptr, y.cs, ga.ptr, ga.cs, cp, sc, sh, mu, istd, nvox, partials.data_ptr(),
              _tail_arg(tail), st)
    if tail is None:
        rm, rv, mom, eps, nbt = (tuple(replay) + (None,))[:5] if replay is not None else (None, None, 0.0, 0.0, None)
        assert nbt is None or (nbt.dtype == torch.int64 and nbt.is_cuda and nbt.numel() == 1)
        _lib.check(lib.ctu_bn_bwd_finalize(partials.data_ptr(), nb, c, cp, float(nvox), gamma.data_ptr(), istd,
                                           dgb[0].data_ptr(), dgb[1].data_ptr(), coef.data_ptr(), mu, _ptr(rm), _ptr(rv),
                                           mom, eps, None if nbt is None else nbt.data_ptr(), st), "bn_bwd_finalize")
    if lazy:
        return dgb[0], dgb[1], coef
    _dual(y.lp, "bn_relu_bwd_apply", y.ptr, y.cs, ga.ptr, ga.cs, cp, sc, sh, mu, istd, coef.data_ptr(), nvox, st)
    return dgb[0], dgb[1]


def bn_bwd_partials_floats(nvox: int, cp: int) -> int:
    return _lib.load().ctu_bn_bwd_num_blocks(nvox) * 2 * cp


# ---------------------------------------------------------------------------- pool
def maxpool_fwd(x: CL, out: CL) -> None:
    n, d, h, w = x.dims
    lib = _lib.load()
    assert out.dtype == x.dtype
    _dual(x.lp, "maxpool2_fwd", x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), out.ptr, out.cs, n, d, h, w,
          _stream())


def maxpool_bwd_bn_blocks(dims, cp: int) -> int:
    n, d, h, w = dims
    return _lib.load().ctu_maxpool2_bwd_bn_num_blocks(n, d, h, w, cp)


def maxpool_bwd(x: CL, gout: CL, gin: CL, accumulate: bool, bn=None, fin=None):
    """bn = (vec [4, cp] scale/shift/mean/invstd of x's BatchNorm, partials): also emit that BatchNorm's backward
    reduction rows into partials; returns their number (pass it to bn_relu_bwd(pre_reduced=...)).
    fin = (gamma, c, replay, counter) with bn: the launch also finalizes that reduction (ctu_bn_bwd_tail); returns
    (rows, (dgb, coef)) -- pass the pair to bn_relu_bwd(finalized=...)."""
    n, d, h, w = x.dims
    lib = _lib.load()
    if bn is not None:
        vec, partials = bn
        nb = lib.ctu_maxpool2_bwd_bn_num_blocks(n, d, h, w, x.cp)
        assert x.scale is not None and x.relu and partials.numel() >= nb * 2 * x.cp
        assert vec[0].data_ptr() == x.scale.data_ptr() and vec[1].data_ptr() == x.shift.data_ptr()
        tail, done = None, None
        if fin is not None:
            gamma, c, replay, counter = fin
            dgb = torch.empty((2, c), dtype=torch.float32, device=x.buf.device)
            coef = torch.empty((5, x.cp), dtype=torch.float32, device=x.buf.device)
            tail, done = make_bn_bwd_tail(c, x.nvox, gamma, vec, dgb, coef, replay, counter), (dgb, coef)
        _dual(x.lp, "maxpool2_bwd_bn", x.ptr, x.cs, x.cp, vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
              vec[3].data_ptr(), gout.ptr, gout.cs, gin.ptr, gin.cs, int(accumulate), n, d, h, w, partials.data_ptr(),
              _tail_arg(tail), _stream())
        return nb if fin is None else (nb, done)
    _dual(x.lp, "maxpool2_bwd", x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), gout.ptr, gout.cs, gin.ptr,
          gin.cs, int(accumulate), n, d, h, w, _stream())


# ---------------------------------------------------------------------------- conv transpose
def pack_convt_w(w: torch.Tensor, cinv, rin_p: int, nout_p: int, mode: int) -> torch.Tensor:
    _need_cuda(w, "convT weight")
    ci, co = w.shape[0], w.shape[1]
    lib = _lib.load()
    wp = torch.empty(lib.ctu_convt_packed_floats(rin_p, nout_p), dtype=torch.float32, device=w.device)
    _lib.check(lib.ctu_pack_convt_weight(w.contiguous().data_ptr(), wp.data_ptr(), ci, co, _ptr(cinv), rin_p, nout_p,
                                         mode, _stream()), "pack_convt_weight")
    return wp


def pack_convt_w_lp(w: torch.Tensor, cinv, rin_p: int, nout_p: int, mode: int, dtype: torch.dtype,
                    into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """16-bit fragment-ordered copy of an fp32 master ConvTranspose3d weight [Ci,Co,2,2,2] (mode 0 forward, 1 data gradient)."""
    _need_cuda(w, "convT weight")
    ci, co = w.shape[0], w.shape[1]
    lib = _lib.load()
    n = lib.ctu_lp_convt_packed_elems(rin_p, nout_p, mode)
    wp = into if into is not None else torch.empty(n, dtype=dtype, device=w.device)
    assert wp.numel() == n and wp.dtype == dtype
    _lib.check(lib.ctu_lp_pack_convt_weight(LP_CODE[dtype], w.contiguous().data_ptr(), wp.data_ptr(), ci, co, _ptr(cinv), rin_p,
                                            nout_p, mode, _stream()), "lp_pack_convt_weight")
    return wp


def _timed_convt(tag: str, x_cp: int, o_cp: int, vox: int, w: int, elem: int):
    """Context for the ConvTranspose launches of the stage table: algorithmic 2*8*C*C FLOP and (C + 8C) elements per coarse voxel."""
    class _T:
        def __enter__(self_):
            self_.t0 = TIMER.begin() if TIMER is not None else None
        def __exit__(self_, *exc):
            if self_.t0 is not None and exc[0] is None:
                TIMER.end(tag, 16.0 * x_cp * o_cp * vox, float(elem) * vox * (x_cp + 8 * o_cp), self_.t0, (w, x_cp, o_cp))
    return _T()


def convt_fwd(x: CL, wp: torch.Tensor, bias: Optional[torch.Tensor], out: CL) -> None:
    n, d, h, w = x.dims
    lib = _lib.load()
    with _timed_convt("convt2_fwd" + ("_lp" if x.lp else ""), x.cp, out.cp, n * d * h * w, w, x.buf.element_size()):
        _convt_fwd(x, wp, bias, out, lib, n, d, h, w)


def _convt_fwd(x, wp, bias, out, lib, n, d, h, w) -> None:
    if x.lp:
        assert out.dtype == x.dtype and wp.dtype == x.dtype
        _lib.check(lib.ctu_lp_convt2_fwd(x.lp, x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), wp.data_ptr(),
                                         _ptr(bias), 0 if bias is None else bias.numel(), out.ptr, out.cs, out.cp, n, d, h, w,
                                         _stream()), "lp_convt2_fwd")
        return
    _lib.check(lib.ctu_convt2_fwd(x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), wp.data_ptr(),
                                  _ptr(bias), 0 if bias is None else bias.numel(), out.ptr, out.cs, out.cp, n, d, h, w,
                                  _stream()), "convt2_fwd")


def convt_bwd_data(gout: CL, wp: torch.Tensor, gin: CL) -> None:
    n, d, h, w = gin.dims
    lib = _lib.load()
    with _timed_convt("convt2_bwd_data" + ("_lp" if gout.lp else ""), gin.cp, gout.cp, n * d * h * w, w, gout.buf.element_size()):
        _convt_bwd_data(gout, wp, gin, lib, n, d, h, w)


def _convt_bwd_data(gout, wp, gin, lib, n, d, h, w) -> None:
    if gout.lp:
        assert gin.dtype == gout.dtype and wp.dtype == gout.dtype
        _lib.check(lib.ctu_lp_convt2_bwd_data(gout.lp, gout.ptr, gout.cs, gout.cp, wp.data_ptr(), gin.ptr, gin.cs, gin.cp, n, d,
                                              h, w, _stream()), "lp_convt2_bwd_data")
        return
    _lib.check(lib.ctu_convt2_bwd_data(gout.ptr, gout.cs, gout.cp, wp.data_ptr(), gin.ptr, gin.cs, gin.cp, n, d, h, w,
                                       _stream()), "convt2_bwd_data")


def convt_wgrad(x: CL, g: CL, ci: int, co: int, imap, ws: torch.Tensor):
    n, d, h, w = x.dims
    lib = _lib.load()
    with _timed_convt("convt2_wgrad" + ("_lp" if x.lp else ""), x.cp, g.cp, n * d * h * w, w, x.buf.element_size()):
        return _convt_wgrad(x, g, ci, co, imap, ws, lib, n, d, h, w)


def _convt_wgrad(x, g, ci, co, imap, ws, lib, n, d, h, w):
    if x.lp:
        assert g.dtype == x.dtype
        assert ws.numel() >= lib.ctu_lp_convt2_wgrad_ws_floats(n, d, h, w, x.cp, g.cp)
        dw = torch.empty((ci, co, 2, 2, 2), dtype=torch.float32, device=x.buf.device)
        _lib.check(lib.ctu_lp_convt2_wgrad(x.lp, x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), g.ptr, g.cs,
                                           g.cp, dw.data_ptr(), ci, co, _ptr(imap), ws.data_ptr(), n, d, h, w, _stream()),
                   "lp_convt2_wgrad")
        return dw, channel_sum(g, co)
    assert ws.numel() >= lib.ctu_convt2_wgrad_ws_floats(n, d, h, w, x.cp, g.cp)
    dw = torch.empty((ci, co, 2, 2, 2), dtype=torch.float32, device=x.buf.device)
    db = torch.empty(co, dtype=torch.float32, device=x.buf.device)
    _lib.check(lib.ctu_convt2_wgrad(x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), g.ptr, g.cs, g.cp,
                                    dw.data_ptr(), db.data_ptr(), ci, co, _ptr(imap), ws.data_ptr(), n, d, h, w,
                                    _stream()), "convt2_wgrad")
    return dw, db


def convt_wgrad_ws(dims, cin_p, cout_p, dtype=torch.float32) -> int:
    n, d, h, w = dims
    if lp(dtype):
        return _lib.load().ctu_lp_convt2_wgrad_ws_floats(n, d, h, w, cin_p, cout_p)
    return _lib.load().ctu_convt2_wgrad_ws_floats(n, d, h, w, cin_p, cout_p)


# ---------------------------------------------------------------------------- head
def head_fwd(x: CL, w: torch.Tensor, b: torch.Tensor, imap, act: int, head_mode: int):
    n, d, h, w_ = x.dims
    co, ci = w.shape[0], w.shape[1]
    v = d * h * w_
    dev = x.buf.device
    lib = _lib.load()
    if head_mode == 0:
        out0 = torch.empty((n, co, d, h, w_), dtype=torch.float32, device=dev)
        out1 = None
    else:
        out0 = torch.empty((n, 2, d, h, w_), dtype=torch.float32, device=dev)
        out1 = torch.empty((n, 2, d, h, w_), dtype=torch.float32, device=dev)
    _dual(x.lp, "head_fwd", x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), w.data_ptr(), b.data_ptr(),
          _ptr(imap), ci, co, act, head_mode, out0.data_ptr(), _ptr(out1), n, v, _stream())
    return out0, out1


def head_bwd(x: CL, w: torch.Tensor, b: torch.Tensor, imap, act: int, head_mode: int, g0: torch.Tensor,
             g1: Optional[torch.Tensor], gin: CL, bn=None, fin=None, gscale: float = 1.0):
    """bn = (vec [4, bn_cp] of the BatchNorm whose activated output is x's first bn_cp channels, partials): also emit
    that BatchNorm's backward reduction rows; returns (dw, db, rows) then (rows -> bn_relu_bwd(pre_reduced=...)).
    fin = (gamma, c, replay) with bn: the launch pair also finalizes that reduction; returns (dw, db, rows, (dgb, coef)).
    gscale (16-bit tensors): g0 / g1 are multiplied by it as they are read (float16 loss scale)."""
    assert gscale == 1.0 or x.lp, "gscale is the 16-bit path's loss scale"

    def call(*args):                                    # (the 16-bit entry carries gscale in front of the stream)
        if x.lp:
            _lib.check(_lib.load().ctu_lp_head_bwd_bn(x.lp, *args[:-1], float(gscale), args[-1]), "lp_head_bwd_bn")
        else:
            _lib.check(_lib.load().ctu_head_bwd_bn(*args), "head_bwd_bn")
    n, d, h, w_ = x.dims
    co, ci = w.shape[0], w.shape[1]
    v = d * h * w_
    dev = x.buf.device
    lib = _lib.load()
    ws = torch.empty(lib.ctu_head_bwd_ws_floats(n, v, x.cp, co), dtype=torch.float32, device=dev)
    dw = torch.empty_like(w)
    db = torch.empty_like(b)
    if bn is not None:
        vec, partials = bn
        bn_cp = vec.shape[1]
        rows = lib.ctu_head_bwd_num_blocks(n, v)
        assert x.scale is not None and x.relu and vec[0].data_ptr() == x.scale.data_ptr() and partials.numel() >= rows * 2 * bn_cp
        tail, done = None, None
        if fin is not None:
            gamma, c, replay = fin[:3]
            dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
            coef = torch.empty((5, bn_cp), dtype=torch.float32, device=dev)
            tail, done = make_bn_bwd_tail(c, n * v, gamma, vec, dgb, coef, replay, None), (dgb, coef)
        call(x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), w.data_ptr(), b.data_ptr(),
             _ptr(imap), ci, co, act, head_mode, g0.data_ptr(), _ptr(g1), gin.ptr, gin.cs, dw.data_ptr(), db.data_ptr(),
              ws.data_ptr(), n, v, vec[2].data_ptr(), vec[3].data_ptr(), bn_cp, partials.data_ptr(), _tail_arg(tail), _stream())
        return (dw, db, rows) if fin is None else (dw, db, rows, done)
    call(x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), w.data_ptr(), b.data_ptr(),
          _ptr(imap), ci, co, act, head_mode, g0.data_ptr(), _ptr(g1), gin.ptr, gin.cs, dw.data_ptr(), db.data_ptr(),
          ws.data_ptr(), n, v, None, None, 0, None, None, _stream())
    return dw, db


def head_bwd_blocks(dims) -> int:
    n, d, h, w_ = dims
    return _lib.load().ctu_head_bwd_num_blocks(n, d * h * w_)


# ---------------------------------------------------------------------------- loss
def loss_fwd(pred: torch.Tensor, target: torch.Tensor, ce_lambda: float, dice_lambda: float, dice_softmax: bool):
    _need_cuda(pred, "prediction")
    _need_cuda(target, "target")
    assert pred.shape == target.shape and pred.shape[1] == 2, "loss kernel handles 2-channel maps"
    pred, target = pred.contiguous(), target.contiguous()
    n = pred.shape[0]
    v = pred[0, 0].numel()
    lib = _lib.load()
    ws = torch.empty(lib.ctu_loss_ws_floats(n, v), dtype=torch.float32, device=pred.device)
    terms = torch.empty(2, dtype=torch.float32, device=pred.device)
    _lib.check(lib.ctu_loss_fwd(pred.data_ptr(), target.data_ptr(), n, v, ce_lambda, dice_lambda, int(dice_softmax),
                                terms.data_ptr(), ws.data_ptr(), _stream()), "loss_fwd")
    return terms, ws


def loss_bwd(pred, target, ce_lambda, dice_lambda, dice_softmax, ws, g_ce: Optional[torch.Tensor],
             g_dice: Optional[torch.Tensor], out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """g_ce / g_dice: the upstream gradients of the two terms (0-d float32 device tensors, read in place; None = 1)."""
    pred, target = pred.contiguous(), target.contiguous()
    for g_ in (g_ce, g_dice):
        assert g_ is None or (g_.is_cuda and g_.dtype == torch.float32 and g_.numel() == 1)
    n = pred.shape[0]
    v = pred[0, 0].numel()
    g = out if out is not None else torch.empty_like(pred)
    lib = _lib.load()
    _lib.check(lib.ctu_loss_bwd(pred.data_ptr(), target.data_ptr(), n, v, ce_lambda, dice_lambda, int(dice_softmax),
                                ws.data_ptr(), _ptr(g_ce), _ptr(g_dice), g.data_ptr(), int(accumulate), _stream()),
               "loss_bwd")
    return g


# ---------------------------------------------------------------------------- misc

def skip_add(a: CL, b: Optional[CL], out: CL) -> None:
    """out = act(a) + act(b) (additive skip, UNet(cat=False)); b None: out = act(a) (channel-slice copy)."""
    assert out.dims == a.dims and out.cp == a.cp and (b is None or (b.dims == a.dims and b.cp == a.cp))
    assert out.dtype == a.dtype and (b is None or b.dtype == a.dtype)
    _dual(a.lp, "skip_add", a.ptr, a.cs, _ptr(a.scale), _ptr(a.shift), int(a.relu), 0 if b is None else b.ptr,
          0 if b is None else b.cs, _ptr(b.scale) if b is not None else 0, _ptr(b.shift) if b is not None else 0,
          int(b.relu) if b is not None else 0, out.ptr, out.cs, out.cp, a.nvox, _stream())


def channel_sum(x: CL, c: int) -> torch.Tensor:
    lib = _lib.load()
    nb = lib.ctu_channel_sum_num_blocks(x.nvox)
    part = torch.empty(nb * x.cp, dtype=torch.float32, device=x.buf.device)
    out = torch.empty(c, dtype=torch.float32, device=x.buf.device)
    _dual(x.lp, "channel_sum", x.ptr, x.cs, x.cp, x.nvox, part.data_ptr(), out.data_ptr(), c, _stream())
    return out


# ---------------------------------------------------------------------------- fused up-convolution (decoder)
def upconv_fused_supported(dims, k: int, cin_p: int, nout_p: int) -> bool:
    n, d, h, w = dims
    return bool(_lib.load().ctu_upconv_fused_supported(k, d, h, w, cin_p, nout_p))


def upconv_fused_pack(wt: torch.Tensor, bt: torch.Tensor, w3: torch.Tensor, cinv, cin_p: int, nout_p: int,
                      into=None):
    """Composite weights (fragment order) and the 27 border-class biases of ConvTranspose3d(C,C,2,2) -> Conv3d(C,Co,3).
    into = (wp, beff): re-pack in place (stable pointers for graph replay)."""
    lib = _lib.load()
    c, co = wt.shape[0], w3.shape[0]
    assert wt.shape == (c, c, 2, 2, 2) and w3.shape == (co, c, 3, 3, 3) and bt.shape == (c,)
    if into is not None:
        wp, beff, ws = into
    else:
        wp = torch.empty(lib.ctu_upconv_fused_packed_floats(cin_p, nout_p), dtype=torch.float32, device=wt.device)
        beff = torch.empty((27, nout_p), dtype=torch.float32, device=wt.device)
        ws = torch.empty(lib.ctu_upconv_fused_pack_ws_floats(c, nout_p), dtype=torch.float32, device=wt.device)
    _lib.check(lib.ctu_upconv_fused_pack(wt.detach().contiguous().data_ptr(), bt.detach().contiguous().data_ptr(),
                                         w3.detach().contiguous().data_ptr(), c, co, _ptr(cinv), cin_p, nout_p, wp.data_ptr(),
                                         beff.data_ptr(), ws.data_ptr(), _stream()), "upconv_fused_pack")
    return wp, beff, ws


def upconv_fused_num_blocks(dims, nout_p: int) -> int:
    n, d, h, w = dims
    return _lib.load().ctu_upconv_fused_num_blocks(n, d, h, w, nout_p)


def upconv_fused_fwd(x: CL, wp: torch.Tensor, beff: torch.Tensor, out: CL, stats: Optional[torch.Tensor],
                     algo_ch: Optional[Tuple[int, int]] = None, tail=None) -> None:
    """out (fine grid, raw) = conv3(convT(act(x))) in one kernel; x is the COARSE input."""
    n, d, h, w = x.dims
    assert out.dims == (n, 2 * d, 2 * h, 2 * w)
    lib = _lib.load()
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.check(lib.ctu_upconv_fused_fwd(x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), wp.data_ptr(),
                                        beff.data_ptr(), out.ptr, out.cs, out.cp, _ptr(stats), n, d, h, w, _tail_arg(tail),
                                        _stream()),
               "upconv_fused_fwd")
    if t0 is not None:
        ci, co = algo_ch if algo_ch is not None else (x.cp, out.cp)
        vox = n * d * h * w
        # algorithmic work of the two layers it replaces: convT 2*ci*ci*8 per coarse voxel + conv 2*27*ci*co per fine voxel
        TIMER.end(f"upconv_fused_fwd_kernel<{out.cp}>", vox * (16.0 * ci * ci + 8 * 54.0 * ci * co), 4.0 * vox * (ci + 8 * co),
                  t0, (w, x.cp, out.cp))


def upconv_fused_wgrad_bn_supported(dims, cin_p: int, nout_p: int, dtype=torch.float32) -> bool:
    n, d, h, w = dims
    return dtype == torch.float32 and bool(_lib.load().ctu_upconv_fused_wgrad_bn_supported(n, d, h, w, cin_p, nout_p))


def upconv_fused_wgrad(x: CL, g: CL, c: int, co: int, bt: torch.Tensor, pack_ws: torch.Tensor, imap, lazy=None):
    """(dWT [C,C,2,2,2], dbT [C], dW3 [Co,C,3,3,3]) of the fused ConvTranspose3d -> Conv3d pair: x = COARSE input of the
    transposed conv, g = fine-grid gradient w.r.t. the conv's raw output, pack_ws = scratch of this step's
    upconv_fused_pack (transposed weights).
    lazy = (y, vec, coef, gy): g is the gradient w.r.t. the ACTIVATED output; the BatchNorm + ReLU backward is applied while
    the weight-gradient kernel stages it and the raw-output gradient lands in gy (read by the projections here and by
    upconv_fused_bwd_data afterwards) -- see conv3d_wgrad_bn."""
    n, d, h, w = x.dims
    assert g.dims == (n, 2 * d, 2 * h, 2 * w)
    lib = _lib.load()
    dev = x.buf.device
    dweff = torch.empty((8, 8, x.cp, g.cp), dtype=torch.float32, device=dev)
    ws = torch.empty(lib.ctu_upconv_fused_wgrad_ws_floats(n, d, h, w, x.cp, g.cp), dtype=torch.float32, device=dev)
    t0 = TIMER.begin() if TIMER is not None else None
    if lazy is not None:
        y, vec, coef, gy = lazy
        assert y.dims == g.dims and gy.dims == g.dims and y.cs == g.cs == gy.cs and y.cp == g.cp == gy.cp
        assert gy.buf.data_ptr() != g.buf.data_ptr()
        _lib.check(lib.ctu_upconv_fused_wgrad_bn(x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), g.ptr, g.cs,
                                                 g.cp, y.ptr, vec[0].data_ptr(), vec[1].data_ptr(), coef.data_ptr(), gy.ptr,
                                                 dweff.data_ptr(), ws.data_ptr(), n, d, h, w, _stream()),
                   "upconv_fused_wgrad_bn")
        g = gy
    else:
        _lib.check(lib.ctu_upconv_fused_wgrad(x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), g.ptr, g.cs, g.cp,
                                              dweff.data_ptr(), ws.data_ptr(), n, d, h, w, _stream()), "upconv_fused_wgrad")
    if t0 is not None:
        vox = n * d * h * w
        TIMER.end(f"upconv_fused_wgrad_kernel<{g.cp}> (+slab reduce)", vox * (16.0 * c * c + 8 * 54.0 * c * co),
                  4.0 * vox * (c + 8 * co), t0, (w, x.cp, g.cp))
    dwt = torch.empty((c, c, 2, 2, 2), dtype=torch.float32, device=dev)
    dbt = torch.empty(c, dtype=torch.float32, device=dev)
    dw3 = torch.empty((co, c, 3, 3, 3), dtype=torch.float32, device=dev)
    ws2 = torch.empty(lib.ctu_upconv_fused_project_ws_floats(g.cp, g.nvox), dtype=torch.float32, device=dev)
    _lib.check(lib.ctu_upconv_fused_project(dweff.data_ptr(), g.ptr, g.cs, g.cp, n, d, h, w, bt.detach().data_ptr(),
                                            pack_ws.data_ptr(), _ptr(imap), c, co, x.cp, dwt.data_ptr(), dbt.data_ptr(),
                                            dw3.data_ptr(), ws2.data_ptr(), _stream()), "upconv_fused_project")
    return dwt, dbt, dw3


# ---- 16-bit twins (upconv_lp.hip): 8 padded output channels, input channels a multiple of 32
def lp_upconv_fused_supported(dims, k: int, cin_p: int, nout_p: int) -> bool:
    n, d, h, w = dims
    return bool(_lib.load().ctu_lp_upconv_fused_supported(k, d, h, w, cin_p, nout_p))


def lp_upconv_fused_pack(wp32: torch.Tensor, cin_p: int, dtype: torch.dtype, into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """16-bit MFMA fragments (forward, then data gradient) of upconv_fused_pack's fp32 composite weights (nout_p = 8)."""
    lib = _lib.load()
    n = lib.ctu_lp_upconv_fused_packed_elems(cin_p)
    wp16 = into if into is not None else torch.empty(n, dtype=dtype, device=wp32.device)
    assert wp16.numel() == n and wp16.dtype == dtype
    _lib.check(lib.ctu_lp_upconv_fused_pack(LP_CODE[dtype], wp32.data_ptr(), cin_p, wp16.data_ptr(), _stream()), "lp_upconv_fused_pack")
    return wp16


def lp_upconv_fused_num_blocks(dims) -> int:
    n, d, h, w = dims
    return _lib.load().ctu_lp_upconv_fused_num_blocks(n, d, h, w)


def lp_upconv_fused_fwd(x: CL, wp16: torch.Tensor, beff: torch.Tensor, out: CL, stats: Optional[torch.Tensor],
                        algo_ch: Optional[Tuple[int, int]] = None) -> None:
    """out (fine grid, 16-bit, raw) = conv3(convT(act(x))) in one kernel; x = the COARSE 16-bit input."""
    n, d, h, w = x.dims
    assert out.dims == (n, 2 * d, 2 * h, 2 * w) and x.lp and out.dtype == x.dtype and wp16.dtype == x.dtype and out.cp == 8
    lib = _lib.load()
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.check(lib.ctu_lp_upconv_fused_fwd(x.lp, x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), wp16.data_ptr(),
                                           beff.data_ptr(), out.ptr, out.cs, _ptr(stats), n, d, h, w, _stream()),
               "lp_upconv_fused_fwd")
    if t0 is not None:
        ci, co = algo_ch if algo_ch is not None else (x.cp, out.cp)
        vox = n * d * h * w
        TIMER.end(f"lp_upconv_fwd_kernel<{'bf16' if x.lp == 1 else 'f16'}>", vox * (16.0 * ci * ci + 8 * 54.0 * ci * co),
                  2.0 * vox * (x.cp + 8 * out.cp), t0, (w, x.cp, out.cp))


def lp_upconv_fused_wgrad_bn_supported(dims, cin_p: int) -> bool:
    n, d, h, w = dims
    return bool(_lib.load().ctu_lp_upconv_fused_wgrad_bn_supported(n, d, h, w, cin_p))


def lp_upconv_fused_wgrad(x: CL, g: CL, c: int, co: int, bt: torch.Tensor, pack_ws: torch.Tensor, imap, lazy=None):
    """(dWT, dbT, dW3) as upconv_fused_wgrad, from 16-bit tensors: x = COARSE input, g = fine-grid raw-output gradient.
    lazy = (y, vec, coef, gy) as in upconv_fused_wgrad."""
    n, d, h, w = x.dims
    assert g.dims == (n, 2 * d, 2 * h, 2 * w) and x.lp and g.dtype == x.dtype and g.cp == 8 and g.cs == 8
    lib = _lib.load()
    dev = x.buf.device
    dweff = torch.empty((8, 8, x.cp, 8), dtype=torch.float32, device=dev)
    ws = torch.empty(lib.ctu_lp_upconv_fused_wgrad_ws_floats(n, d, h, w, x.cp), dtype=torch.float32, device=dev)
    t0 = TIMER.begin() if TIMER is not None else None
    if lazy is not None:
        y, vec, coef, gy = lazy
        assert y.dims == g.dims and gy.dims == g.dims and y.cs == g.cs == gy.cs and gy.buf.data_ptr() != g.buf.data_ptr()
        _lib.check(lib.ctu_lp_upconv_fused_wgrad_bn(x.lp, x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), g.ptr, g.cs,
                                                    y.ptr, vec[0].data_ptr(), vec[1].data_ptr(), coef.data_ptr(), gy.ptr,
                                                    dweff.data_ptr(), ws.data_ptr(), n, d, h, w, _stream()), "lp_upconv_fused_wgrad_bn")
        g = gy
    else:
        _lib.check(lib.ctu_lp_upconv_fused_wgrad(x.lp, x.ptr, x.cs, x.cp, _ptr(x.scale), _ptr(x.shift), int(x.relu), g.ptr, g.cs,
                                                 dweff.data_ptr(), ws.data_ptr(), n, d, h, w, _stream()), "lp_upconv_fused_wgrad")
    if t0 is not None:
        vox = n * d * h * w
        TIMER.end(f"lp_upconv_wgrad_kernel<{'bf16' if x.lp == 1 else 'f16'}> (+slab reduce)", vox * (16.0 * c * c + 8 * 54.0 * c * co),
                  2.0 * vox * (x.cp + 8 * g.cp), t0, (w, x.cp, g.cp))
    dwt = torch.empty((c, c, 2, 2, 2), dtype=torch.float32, device=dev)
    dbt = torch.empty(c, dtype=torch.float32, device=dev)
    dw3 = torch.empty((co, c, 3, 3, 3), dtype=torch.float32, device=dev)
    ws2 = torch.empty(lib.ctu_upconv_fused_project_ws_floats(g.cp, g.nvox), dtype=torch.float32, device=dev)
    _lib.check(lib.ctu_lp_upconv_fused_project(x.lp, dweff.data_ptr(), g.ptr, g.cs, g.cp, n, d, h, w, bt.detach().data_ptr(),
                                               pack_ws.data_ptr(), _ptr(imap), c, co, x.cp, dwt.data_ptr(), dbt.data_ptr(),
                                               dw3.data_ptr(), ws2.data_ptr(), _stream()), "lp_upconv_fused_project")
    return dwt, dbt, dw3


def lp_upconv_fused_bwd_data(g: CL, wp16: torch.Tensor, gin: CL, algo_ch: Optional[Tuple[int, int]] = None) -> None:
    """gin (coarse, 16-bit) <- gradient of the fused pair w.r.t. its activated input, from the fine-grid gradient g."""
    n, d, h, w = gin.dims
    assert g.dims == (n, 2 * d, 2 * h, 2 * w) and g.lp and gin.dtype == g.dtype and wp16.dtype == g.dtype and g.cp == 8
    lib = _lib.load()
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.check(lib.ctu_lp_upconv_fused_bwd_data(g.lp, g.ptr, g.cs, wp16.data_ptr(), gin.ptr, gin.cs, gin.cp, n, d, h, w, _stream()),
               "lp_upconv_fused_bwd_data")
    if t0 is not None:
        ci, co = algo_ch if algo_ch is not None else (gin.cp, g.cp)
        vox = n * d * h * w
        TIMER.end(f"lp_upconv_bwd_data_kernel<{'bf16' if g.lp == 1 else 'f16'}>", vox * (16.0 * ci * ci + 8 * 54.0 * ci * co),
                  2.0 * vox * (gin.cp + 8 * g.cp), t0, (w, gin.cp, g.cp))


def upconv_fused_pack_bwd(wp: torch.Tensor, cin_p: int, nout_p: int, into: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    wpd = into if into is not None else torch.empty(lib.ctu_upconv_fused_bwd_packed_floats(cin_p, nout_p),
                                                     dtype=torch.float32, device=wp.device)
    _lib.check(lib.ctu_upconv_fused_pack_bwd(wp.data_ptr(), cin_p, nout_p, wpd.data_ptr(), _stream()), "upconv_fused_pack_bwd")
    return wpd


def upconv_fused_bwd_data(g: CL, wpd: torch.Tensor, gin: CL, algo_ch: Optional[Tuple[int, int]] = None) -> None:
    """gin (coarse) <- gradient of the fused ConvTranspose3d -> Conv3d pair w.r.t. its (activated) input."""
    n, d, h, w = gin.dims
    assert g.dims == (n, 2 * d, 2 * h, 2 * w)
    lib = _lib.load()
    t0 = TIMER.begin() if TIMER is not None else None
    _lib.check(lib.ctu_upconv_fused_bwd_data(g.ptr, g.cs, g.cp, wpd.data_ptr(), gin.ptr, gin.cs, gin.cp, n, d, h, w, _stream()),
               "upconv_fused_bwd_data")
    if t0 is not None:
        ci, co = algo_ch if algo_ch is not None else (gin.cp, g.cp)
        vox = n * d * h * w
        TIMER.end(f"upconv_fused_bwd_data_kernel<{g.cp}>", vox * (16.0 * ci * ci + 8 * 54.0 * ci * co), 4.0 * vox * (ci + 8 * co),
                  t0, (w, gin.cp, g.cp))


# ---------------------------------------------------------------------------- inference tail / sample schema
def _ncv(t: torch.Tensor):
    """(N, C, V) of a contiguous fp32 NCDHW (5-D) or CDHW (4-D) CUDA map."""
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.dim() not in (4, 5):
        raise RuntimeError("ctunet_amd: expected a contiguous float32 CUDA map [N,C,D,H,W] or [C,D,H,W]")
    n = t.shape[0] if t.dim() == 5 else 1
    c = t.shape[1] if t.dim() == 5 else t.shape[0]
    return n, c, t.numel() // (n * c)


def hard_segm(prob: torch.Tensor) -> torch.Tensor:
    """argmax over the class dimension as float32 ([N,D,H,W] or [D,H,W]); first maximum wins."""
    n, c, v = _ncv(prob)
    out = torch.empty(prob.shape[:1] + prob.shape[2:] if prob.dim() == 5 else prob.shape[1:], dtype=torch.float32,
                      device=prob.device)
    _lib.check(_lib.load().ctu_hard_segm(prob.data_ptr(), n, c, v, out.data_ptr(), _stream()), "hard_segm")
    return out


def one_hot(label: torch.Tensor, num_classes: int) -> torch.Tensor:
    """one_hot(label.long(), C).movedim(-1, 1).float() for a float32 CUDA label volume [N,D,H,W]."""
    if label.dtype != torch.float32 or not label.is_cuda or not label.is_contiguous() or label.dim() != 4:
        raise RuntimeError("ctunet_amd: expected a contiguous float32 CUDA label volume [N,D,H,W]")
    n = label.shape[0]
    out = torch.empty((n, num_classes) + tuple(label.shape[1:]), dtype=torch.float32, device=label.device)
    _lib.check(_lib.load().ctu_one_hot(label.data_ptr(), n, num_classes, label.numel() // n, out.data_ptr(), _stream()),
               "one_hot")
    return out


def hard_dice_counts(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """float64 [N, C, 3]: per item and class {|hard & target|, |hard|, |target|}, hard = one_hot(argmax(pred, 1))."""
    n, c, v = _ncv(pred)
    if target.shape != pred.shape or _ncv(target) != (n, c, v):
        raise RuntimeError("ctunet_amd: prediction and one-hot target must have the same shape")
    lib = _lib.load()
    ws = torch.empty(lib.ctu_hard_dice_ws_doubles(n), dtype=torch.float64, device=pred.device)
    counts = torch.empty((n, c, 3), dtype=torch.float64, device=pred.device)
    _lib.check(lib.ctu_hard_dice_counts(pred.data_ptr(), target.data_ptr(), n, c, v, counts.data_ptr(), ws.data_ptr(),
                                        _stream()), "hard_dice_counts")
    return counts


def hausdorff(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """float32 [N, C-1]: symmetric Euclidean Hausdorff distance between the surfaces of argmax(pred, 1) == c and
    target[:, c] for c = 1..C-1 (NaN where either surface is empty)."""
    n, c, v = _ncv(pred)
    if pred.dim() != 5 or target.shape != pred.shape or _ncv(target) != (n, c, v):
        raise RuntimeError("ctunet_amd: hausdorff expects a [N,C,D,H,W] prediction and a one-hot target of the same shape")
    d, h, w = pred.shape[2:]
    lib = _lib.load()
    ws = torch.empty(lib.ctu_hausdorff_ws_bytes(n, c, d, h, w), dtype=torch.uint8, device=pred.device)
    out = torch.empty((n, c - 1), dtype=torch.float32, device=pred.device)
    _lib.check(lib.ctu_hausdorff(pred.data_ptr(), target.data_ptr(), n, c, d, h, w, out.data_ptr(), ws.data_ptr(), _stream()),
               "hausdorff")
    return out


def extract_patches(vol: torch.Tensor, coords: torch.Tensor, patch: Tuple[int, int, int]) -> torch.Tensor:
    """vol [C,D,H,W] float32 CUDA, coords int32 CUDA [P,3] (z0,y0,x0) -> [P,C,pd,ph,pw] (zero outside the volume)."""
    _need_cuda(vol, "volume")
    _need_cuda(coords, "patch coordinates")
    if vol.dim() != 4 or coords.dim() != 2 or coords.shape[1] != 3 or coords.dtype != torch.int32:
        raise RuntimeError("ctunet_amd: extract_patches expects vol [C,D,H,W] and int32 coords [P,3]")
    vol, coords = vol.contiguous(), coords.contiguous()
    c, d, h, w = vol.shape
    p = coords.shape[0]
    out = torch.empty((p, c) + tuple(patch), dtype=torch.float32, device=vol.device)
    _lib.check(_lib.load().ctu_extract_patches(vol.data_ptr(), coords.data_ptr(), p, c, d, h, w, patch[0], patch[1], patch[2],
                                               out.data_ptr(), _stream()), "extract_patches")
    return out


def stitch_patches(patches: torch.Tensor, coords: torch.Tensor, shape: Tuple[int, int, int]) -> torch.Tensor:
    """patches [P,C,pd,ph,pw] + coords -> [C,D,H,W]: mean over the patches covering each voxel."""
    _need_cuda(patches, "patches")
    _need_cuda(coords, "patch coordinates")
    if patches.dim() != 5 or coords.shape != (patches.shape[0], 3) or coords.dtype != torch.int32:
        raise RuntimeError("ctunet_amd: stitch_patches expects patches [P,C,pd,ph,pw] and int32 coords [P,3]")
    patches, coords = patches.contiguous(), coords.contiguous()
    p, c, pd, ph, pw = patches.shape
    out = torch.empty((c,) + tuple(shape), dtype=torch.float32, device=patches.device)
    _lib.check(_lib.load().ctu_stitch_patches(patches.data_ptr(), coords.data_ptr(), p, c, shape[0], shape[1], shape[2], pd, ph,
                                              pw, out.data_ptr(), _stream()), "stitch_patches")
    return out


def scale_tensors(tensors, s: float, nonfinite: Optional[torch.Tensor] = None) -> None:
    """Every float32 CUDA tensor of the list scaled in place by s, one launch (un-scaling of loss-scaled gradients).
    nonfinite: float32[1] device flag set to 1 when any scaled value is inf / NaN (fp16 overflow detection)."""
    import ctypes as C
    ts = [t for t in tensors if t is not None]
    if not ts:
        return
    for t in ts:
        _need_cuda(t, "tensor")
        assert t.is_contiguous() and t.dtype == torch.float32
    pa = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    sa = (C.c_int64 * len(ts))(*[t.numel() for t in ts])
    assert nonfinite is None or (nonfinite.is_cuda and nonfinite.dtype == torch.float32 and nonfinite.numel() == 1)
    _lib.check(_lib.load().ctu_scale_tensors(pa, sa, len(ts), float(s), _ptr(nonfinite), _stream()), "scale_tensors")
