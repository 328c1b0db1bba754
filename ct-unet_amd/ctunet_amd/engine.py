"""Kernel sequencing for the U-Net forward / backward on MI355X.

``UNetEngine`` turns a ``NetPlan`` (the channel plan of one of the reference's model classes)
into the sequence of libctunet_hip.so calls that computes ``UNet.forward``
(/root/reference/ctunet/pytorch/models.py:226-261, legacy :509-538) and its backward.  It is
the analogue of torch autograd + ATen dispatch for this one graph; all arithmetic happens in the
HIP kernels (``ops``), torch only owns memory and the stream.

Data flow (DESIGN.md has the picture):
  * activations are channels-last fp32, channels padded to 8;
  * every conv stores its RAW output and BatchNorm+ReLU is applied lazily by the consumer
    (per-channel scale/shift from ``bn_finalize``), so BN/ReLU never cost an HBM pass forward;
  * the skip concat is free: the decoder's second conv writes channels [0, Cp) and the encoder's
    second conv writes channels [Cp, 2Cp) of one buffer per resolution level
    (``torch.cat((ubl, d), 1)``, models.py:249 -- upsampled first, skip second);
  * backward re-reads the raw tensors and recomputes activations/pool argmax on the fly; nothing
    but raw conv outputs and per-channel statistics is saved.
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .ops import CL, pad8

BN_EPS = 1e-5
# decoder ConvTranspose3d -> Conv3d pairs run as one coarse-grid kernel (CTUNET_FUSE_UP=0: the two separate kernels)
FUSE_UP = os.environ.get("CTUNET_FUSE_UP", "1") != "0"
FUSE_UP_MAX_CO = int(os.environ.get("CTUNET_FUSE_UP_MAXCO", "16"))     # widest (padded) conv output that takes the fused path
# the max-pool backward also emits the BatchNorm-backward reduction of the layer it pools (CTUNET_POOL_BN=0: separate pass)
POOL_BN = os.environ.get("CTUNET_POOL_BN", "1") != "0"
# 16-bit path: the decoder's top-level up-convolution (8 padded output channels, input channels a multiple of 32) runs through the
# 16-bit fused kernels of upconv_lp.hip; every other level takes the unfused 16-bit ConvTranspose3d + Conv3d kernels.
# CTUNET_LP_FUSE_UP=0: the unfused kernels everywhere.
LP_FUSE_UP = os.environ.get("CTUNET_LP_FUSE_UP", "1") != "0"
BN_MOMENTUM = 0.1
# BatchNorm + ReLU backward applied by the weight-gradient kernel while it stages the gradient (ops.conv3d_wgrad_bn) instead
# of a separate in-place pass over the layer (CTUNET_LAZY_BN=0: the separate ctu_bn_relu_bwd_apply pass)
LAZY_BN = os.environ.get("CTUNET_LAZY_BN", "1") != "0"
# ... on the 16-bit path too (ops.conv3d_wgrad_bn with 16-bit tensors; every 16-bit weight-gradient kernel has the LZ variant).
# Off by default -- measured SLOWER at every size: the separate pass moves half the bytes it moves in fp32, while the folded
# form adds a read of y and a write of the second gradient buffer to HBM-bound kernels (UNet() 128^3 bf16 2.31 ms with it, 2.24
# without; UNetSP 192^3 bf16 5.92 vs 5.69; 256^3 fp16 11.96 vs 11.39)
LAZY_BN_LP = os.environ.get("CTUNET_LAZY_BN_LP", "0") != "0"
# the launch that writes a BatchNorm's partial rows also finalizes them (its last block: ctu_bn_tail / ctu_bn_bwd_tail)
# instead of a separate ctu_bn_finalize / ctu_bn_bwd_finalize launch (CTUNET_BN_TAIL=0: the separate launches)
BN_TAIL = os.environ.get("CTUNET_BN_TAIL", "0") != "0"
# the generic UNet's dead centre block (models.py:241) runs on a forked stream beside the decoder (CTUNET_CENTER_SIDE=0: in line)
CENTER_SIDE = os.environ.get("CTUNET_CENTER_SIDE", "0") != "0"


@dataclass
class BlockPlan:
    prefix: str          # state_dict prefix of the Sequential, e.g. "d_blocks.0.block"
    first: int           # index of the first Conv3d in it (1 when a ConvTranspose3d sits at 0)
    cin: int             # logical input channels of the first conv (= ConvTranspose channels for up blocks)
    cout: int


@dataclass
class NetPlan:
    k: int
    conv_bias: bool
    in_ch: int
    out_ch: int
    enc: List[BlockPlan]
    center: BlockPlan
    center_live: bool    # False: generic UNet drops the centre block's output (models.py:241)
    dec: List[BlockPlan]
    head: str
    act: int             # bit0 softmax, bit1 sigmoid
    head_mode: int       # 0 plain, 1 SP re-encoding, 2 SP + softmax
    skip: str = "cat"    # how a decoder level meets its encoder skip (models.py:247-253): "cat" | "add" | "none"


class _ConvRec:
    """What one conv+BN stage leaves behind for backward."""
    __slots__ = ("x", "y", "vec", "stats", "nblk", "conv", "bn", "cin", "cout", "imap", "bias", "first")


class UNetEngine:
    def __init__(self, plan: NetPlan, dtype: torch.dtype = torch.float32, loss_scale: Optional[float] = None):
        """dtype: storage type of every activation / activation-gradient tensor (float32, or bfloat16 / float16 for the
        reduced-precision path of BASELINE configs 4 / 5: 16-bit tensors in HBM, fp32 MFMA accumulation, fp32 BatchNorm
        statistics, fp32 master weights and weight gradients).  loss_scale: float16 only -- the factor the incoming output
        gradients are multiplied by before they enter the 16-bit backward (and every parameter gradient divided by at
        the end); None = a power of two near voxels / 16, which puts the per-voxel loss gradient of a mean-reduced loss
        (1 / voxels: 6e-8 for a 256^3 patch, below fp16's smallest normal 6e-5) at about 1/16."""
        if dtype not in ops.ACT_DTYPES:
            raise ValueError(f"ctunet_amd: activation dtype {dtype} not supported (float32, bfloat16, float16)")
        self.plan = plan
        self.dtype = dtype
        self.loss_scale = loss_scale
        self._pack_cache: Dict[Tuple, Tuple[int, torch.Tensor]] = {}
        self._replay_stats = False
        self._up_cache: Dict[str, Tuple] = {}
        self._imaps: Dict[Tuple, torch.Tensor] = {}
        self._tail_words: Dict[str, torch.Tensor] = {}      # device -> int32 ticket counters, one per (layer, direction)
        self._tail_slots: Dict[Tuple[str, str], int] = {}

    # ------------------------------------------------------------------ small helpers
    def overflow_flag(self, device) -> torch.Tensor:
        """float32[1] on the device: 1 after a float16 backward whose (un-scaled) gradients contained inf / NaN, else 0.  One
        tensor per engine and device for its whole life (stable pointer: captured graphs and the optimizer keep it)."""
        flags = self.__dict__.setdefault("_overflow", {})
        f = flags.get(str(device))
        if f is None:
            f = flags[str(device)] = torch.zeros(1, dtype=torch.float32, device=device)
        return f

    def _counter(self, layer: str, direction: str, device) -> Optional[torch.Tensor]:
        """The zero-initialised ticket word of a BatchNorm layer's in-launch finalize (None: BN_TAIL off).  The tail puts
        it back to zero itself, so the words are allocated once and live as long as the engine (graph replays included)."""
        if not BN_TAIL:
            return None
        words = self._tail_words.get(str(device))
        if words is None:
            words = self._tail_words[str(device)] = torch.zeros(1024, dtype=torch.int32, device=device)
        slot = self._tail_slots.setdefault((layer, direction), len(self._tail_slots))
        if slot >= words.numel():
            raise RuntimeError("ctunet_amd: more than 1024 BatchNorm tails in one engine")
        return words[slot:slot + 1]

    def _maps(self, segs: Tuple[Tuple[int, int], ...], cp: int, device):
        """segs: ((n_logical, padded_start), ...).  Returns (imap, cinv) int32 device tensors:
        imap[logical] -> padded position, cinv[padded position] -> logical or -1; (None, None) = identity."""
        idx: List[int] = []
        for n, start in segs:
            idx += list(range(start, start + n))
        if idx == list(range(len(idx))):
            return None, None
        key = (segs, cp, str(device))
        t = self._imaps.get(key)
        if t is None:
            inv = [-1] * cp
            for logical, pos in enumerate(idx):
                inv[pos] = logical
            t = (torch.tensor(idx, dtype=torch.int32, device=device), torch.tensor(inv, dtype=torch.int32, device=device))
            self._imaps[key] = t
        return t

    def _packed(self, name: str, w: torch.Tensor, kind: str, imap, rin_p: int, nout_p: int, mode: int,
                layout: int = 0) -> torch.Tensor:
        """MFMA-ordered copy of a weight tensor, cached by (tensor version, address).  The first request of a key
        packs it on the spot and records the job; refresh_packs() then re-packs every recorded job whose weight
        changed in ONE launch at the start of the next forward (into the same buffers: stable pointers)."""
        key = (name, kind, mode, rin_p, nout_p, layout, self.dtype)
        hit = self._pack_cache.get(key)
        ver = (w._version, w.data_ptr())
        if hit is not None and hit[0] == ver:
            return hit[1]
        wd = w.detach()
        if self.dtype != torch.float32:                # 16-bit fragment-ordered copies of the fp32 masters
            wp = self._pack_lp(kind, wd, imap, rin_p, nout_p, mode, hit[1] if hit is not None else None, layout)
        else:
            wp = hit[1] if hit is not None else torch.empty(ops.packed_floats(kind, w.shape[2], rin_p, nout_p, layout),
                                                            dtype=torch.float32, device=w.device)
            ops.pack_batch([(kind, wd.contiguous(), wp, imap, rin_p, nout_p, mode, layout)])
        self._pack_cache[key] = (ver, wp, imap)
        return wp

    def _pack_lp(self, kind, wd, imap, rin_p, nout_p, mode, into, layout=0):
        if kind == "conv":
            return ops.pack_conv_w_lp(wd, imap, rin_p, nout_p, mode, self.dtype, into, layout)
        return ops.pack_convt_w_lp(wd, imap, rin_p, nout_p, mode, self.dtype, into)

    def refresh_packs(self, P: Dict[str, torch.Tensor]) -> None:
        # under graph capture every copy is re-packed unconditionally: the replayed graph must refresh them after each
        # optimizer step, whether or not the weights happened to change between the last warm-up step and the capture
        force = P and next(iter(P.values())).is_cuda and torch.cuda.is_current_stream_capturing()
        jobs = []
        for key, ent in self._pack_cache.items():
            name, kind, mode, rin_p, nout_p, layout, dt = key
            w = P.get(name + ".weight")
            if w is None or w.device != ent[1].device or dt != self.dtype:
                continue
            ver = (w._version, w.data_ptr())
            if force or ent[0] != ver:
                jobs.append((kind, w.detach().contiguous(), ent[1], ent[2], rin_p, nout_p, mode, layout))
                self._pack_cache[key] = (ver, ent[1], ent[2])
        if self.dtype != torch.float32:
            ops.pack_batch_lp(jobs, self.dtype)
        else:
            ops.pack_batch(jobs)

    # ------------------------------------------------------------------ forward pieces
    def _fwd_tail(self, P, bn: str, c: int, nvox: int, n_upd: int, vec4: torch.Tensor, device):
        counter = self._counter(bn, "fwd", device)
        if counter is None:
            return None
        return ops.make_bn_tail(c, nvox, P[bn + ".weight"], P[bn + ".bias"], P[bn + ".running_mean"], P[bn + ".running_var"],
                                BN_MOMENTUM, BN_EPS, n_upd, vec4, P.get(bn + ".num_batches_tracked") if n_upd else None,
                                counter)

    def _conv_bn(self, P, x: CL, conv: str, bn: str, cin: int, cout: int, imap, out: CL, vec4: torch.Tensor,
                 training: bool, n_upd: int, save: bool) -> Tuple[CL, Optional[_ConvRec]]:
        k = self.plan.k
        w = P[conv + ".weight"]
        bias = P.get(conv + ".bias")
        lay = ops.conv_layout(k, out.cp, x.dims[3], self.dtype, x.cp)
        if bias is not None and self.dtype != torch.float32:
            lay = 0                                  # (the 16-bit pair-layout kernel carries no bias)
        wp = self._packed(conv, w, "conv", imap, x.cp, out.cp, 0, lay)
        bias_p = None if bias is None else bias.detach()
        dims = x.dims
        c = cout
        if training:
            nblk = ops.conv_num_blocks(dims, out.cp, lay, k, self.dtype, x.cp)
            stats = torch.empty((nblk, 2, out.cp), dtype=torch.float32, device=x.buf.device)
            tail = self._fwd_tail(P, bn, c, x.nvox, n_upd, vec4, x.buf.device)
            ops.conv3d_fwd(x, wp, bias_p, out, k, stats, (cin, cout), lay, tail)
            if tail is None:
                ops.bn_finalize_into(stats, nblk, c, out.cp, x.nvox, P[bn + ".weight"], P[bn + ".bias"],
                                     P[bn + ".running_mean"], P[bn + ".running_var"], BN_MOMENTUM, BN_EPS, n_upd, vec4,
                                     P.get(bn + ".num_batches_tracked") if n_upd else None)
        else:
            stats, nblk = None, 0
            ops.conv3d_fwd(x, wp, bias_p, out, k, None, (cin, cout), lay)
            ops.bn_eval_affine_into(P[bn + ".weight"], P[bn + ".bias"], P[bn + ".running_mean"],
                                    P[bn + ".running_var"], BN_EPS, c, out.cp, vec4)
        y = out.with_xf(vec4[0], vec4[1], True)
        rec = None
        if save:
            rec = _ConvRec()
            rec.first = None
            rec.x, rec.y, rec.vec, rec.stats, rec.nblk = x, out.raw(), vec4, stats, nblk
            rec.conv, rec.bn, rec.cin, rec.cout, rec.imap, rec.bias = conv, bn, cin, cout, imap, bias is not None
        return y, rec

    def _fuse_up(self, x: CL, nout_p: int) -> bool:
        """Run ConvTranspose3d -> Conv3d of a decoder block as one coarse-grid kernel (ops.upconv_fused_fwd)?
        Levels whose first conv has at most 16 (padded) output channels: they hold the FLOPs and have enough boxes."""
        if not (self.plan.k == 3 and not self.plan.conv_bias and nout_p <= FUSE_UP_MAX_CO and FUSE_UP):
            return False
        if self.dtype == torch.float32:
            return ops.upconv_fused_supported(x.dims, 3, x.cp, nout_p)
        return LP_FUSE_UP and ops.lp_upconv_fused_supported(x.dims, 3, x.cp, nout_p)

    def _upconv_bn(self, P, x: CL, prefix: str, ct: int, cout: int, cinv, out: CL, vec4: torch.Tensor, training: bool,
                   n_upd: int, save: bool) -> Tuple[CL, Optional[_ConvRec]]:
        """{prefix}.0 (ConvTranspose3d) + {prefix}.1 (Conv3d) fused, then {prefix}.2 (BatchNorm) as in _conv_bn.
        The record's input (the transposed conv's output) is NOT materialised: backward recomputes it (rec.x None)."""
        conv, bn = f"{prefix}.1", f"{prefix}.2"
        wt, bt, w3 = P[f"{prefix}.0.weight"], P[f"{prefix}.0.bias"], P[conv + ".weight"]
        lp = bool(x.lp)                    # 16-bit: the composite is still built in fp32 and rounded once into 16-bit fragments
        ver = tuple((t._version, t.data_ptr()) for t in (wt, bt, w3))
        hit = self._up_cache.get(prefix)
        if hit is None or hit[0] != ver or hit[3] != (x.cp, out.cp, self.dtype) or torch.cuda.is_current_stream_capturing():
            same = hit is not None and hit[3] == (x.cp, out.cp, self.dtype)
            wp, beff, pws = ops.upconv_fused_pack(wt, bt, w3, cinv, x.cp, out.cp, (hit[1], hit[2], hit[4]) if same else None)
            if lp:
                wpd = ops.lp_upconv_fused_pack(wp, x.cp, self.dtype, hit[5] if same else None)      # forward + data-gradient fragments
            else:
                wpd = ops.upconv_fused_pack_bwd(wp, x.cp, out.cp, hit[5] if same else None)          # same weights, data-gradient order
            self._up_cache[prefix] = (ver, wp, beff, (x.cp, out.cp, self.dtype), pws, wpd)
        else:
            wp, beff, wpd = hit[1], hit[2], hit[5]
        nvox = 8 * x.nvox
        if training:
            nblk = ops.lp_upconv_fused_num_blocks(x.dims) if lp else ops.upconv_fused_num_blocks(x.dims, out.cp)
            stats = torch.empty((nblk, 2, out.cp), dtype=torch.float32, device=x.buf.device)
            tail = None if lp else self._fwd_tail(P, bn, cout, nvox, n_upd, vec4, x.buf.device)
            if lp:
                ops.lp_upconv_fused_fwd(x, wpd, beff, out, stats, (ct, cout))
            else:
                ops.upconv_fused_fwd(x, wp, beff, out, stats, (ct, cout), tail)
            if tail is None:
                ops.bn_finalize_into(stats, nblk, cout, out.cp, nvox, P[bn + ".weight"], P[bn + ".bias"],
                                     P[bn + ".running_mean"], P[bn + ".running_var"], BN_MOMENTUM, BN_EPS, n_upd, vec4,
                                     P.get(bn + ".num_batches_tracked") if n_upd else None)
        else:
            stats, nblk = None, 0
            if lp:
                ops.lp_upconv_fused_fwd(x, wpd, beff, out, None, (ct, cout))
            else:
                ops.upconv_fused_fwd(x, wp, beff, out, None, (ct, cout))
            ops.bn_eval_affine_into(P[bn + ".weight"], P[bn + ".bias"], P[bn + ".running_mean"],
                                    P[bn + ".running_var"], BN_EPS, cout, out.cp, vec4)
        y = out.with_xf(vec4[0], vec4[1], True)
        rec = None
        if save:
            rec = _ConvRec()
            rec.first = None
            rec.x, rec.y, rec.vec, rec.stats, rec.nblk = None, out.raw(), vec4, stats, nblk
            rec.conv, rec.bn, rec.cin, rec.cout, rec.imap, rec.bias = conv, bn, ct, cout, None, False
        return y, rec

    def _first_conv_bn(self, P, x: torch.Tensor, conv: str, bn: str, cin: int, cout: int, out: CL, vec4: torch.Tensor,
                       training: bool, n_upd: int, save: bool):
        """First encoder conv through the direct C_in <= 2 kernels (reads the NCDHW input in place)."""
        w = P[conv + ".weight"].detach()
        bias = P.get(conv + ".bias")
        bias_p = None if bias is None else bias.detach()
        dims = out.dims
        nvox = dims[0] * dims[1] * dims[2] * dims[3]
        if training:
            nblk = ops.conv_first_num_blocks(dims)
            stats = torch.empty((nblk, 2, out.cp), dtype=torch.float32, device=x.device)
            tail = self._fwd_tail(P, bn, cout, nvox, n_upd, vec4, x.device)
            ops.conv_first_fwd(x, w, bias_p, out, stats, tail)
            if tail is None:
                ops.bn_finalize_into(stats, nblk, cout, out.cp, nvox, P[bn + ".weight"], P[bn + ".bias"],
                                     P[bn + ".running_mean"], P[bn + ".running_var"], BN_MOMENTUM, BN_EPS, n_upd, vec4,
                                     P.get(bn + ".num_batches_tracked") if n_upd else None)
        else:
            stats, nblk = None, 0
            ops.conv_first_fwd(x, w, bias_p, out, None)
            ops.bn_eval_affine_into(P[bn + ".weight"], P[bn + ".bias"], P[bn + ".running_mean"],
                                    P[bn + ".running_var"], BN_EPS, cout, out.cp, vec4)
        y = out.with_xf(vec4[0], vec4[1], True)
        rec = None
        if save:
            rec = _ConvRec()
            rec.first = x
            rec.x, rec.y, rec.vec, rec.stats, rec.nblk = None, out.raw(), vec4, stats, nblk
            rec.conv, rec.bn, rec.cin, rec.cout, rec.imap, rec.bias = conv, bn, cin, cout, None, bias is not None
        return y, rec

    def forward(self, P: Dict[str, torch.Tensor], x: torch.Tensor, training: bool, save: bool, chk: bool):
        """Returns (out0, out1 | None, ctx | None)."""
        plan = self.plan
        dev = x.device
        n, cin, d, h, w = x.shape
        nlev = len(plan.enc)
        if cin != plan.in_ch:
            raise RuntimeError(f"ctunet_amd: expected {plan.in_ch} input channels, got {cin}")
        if any(s % (1 << nlev) for s in (d, h, w)):
            raise RuntimeError(f"ctunet_amd: spatial size {(d, h, w)} must be divisible by {1 << nlev}")
        if training and n * (d >> nlev) * (h >> nlev) * (w >> nlev) <= 1:
            # the reference's BatchNorm3d raises here too ("Expected more than 1 value per channel")
            raise ValueError("Expected more than 1 value per channel when training (centre block)")
        ctx = {"recs": {}, "levels": [], "training": training, "chk": chk} if save else None
        n_upd = 1 if training else 0
        self.refresh_packs(P)

        x = x.contiguous()
        e0 = plan.enc[0]
        first_direct = plan.conv_bias is False and ops.conv_first_supported(plan.k, cin, pad8(e0.cout), w)
        adt = self.dtype
        cur = None if first_direct else ops.ncdhw_to_cl(x, dtype=adt)
        x_cl = cur
        cat: List[torch.Tensor] = []     # concat buffer per level
        xf: List[torch.Tensor] = []      # [4, 2Cp] scale/shift/mean/invstd of the concat buffer
        pooled: List[CL] = []
        dskip: List[CL] = []
        dd, hh, ww = d, h, w
        recs = {}
        # one zero-filled allocation for every level's [4, 2Cp] vector block (one fill launch instead of one per level)
        xf_sizes = [8 * pad8(blk.cout) for blk in plan.enc]
        xf_all = torch.zeros(sum(xf_sizes), dtype=torch.float32, device=dev)
        xf_off = 0
        for i, blk in enumerate(plan.enc):
            cp = pad8(blk.cout)
            cat.append(torch.empty((n, dd, hh, ww, 2 * cp), dtype=adt, device=dev))
            xf.append(xf_all[xf_off:xf_off + 8 * cp].view(4, 2 * cp))
            xf_off += 8 * cp
            t1 = CL(torch.empty((n, dd, hh, ww, cp), dtype=adt, device=dev), 0, cp)
            v1 = torch.empty((4, cp), dtype=torch.float32, device=dev)
            imap = None
            if i == 0 and first_direct:
                a1, recs[(blk.prefix, 1)] = self._first_conv_bn(P, x, f"{blk.prefix}.{blk.first}",
                                                                f"{blk.prefix}.{blk.first + 1}", blk.cin, blk.cout, t1,
                                                                v1, training, n_upd, save)
            else:
                a1, recs[(blk.prefix, 1)] = self._conv_bn(P, cur, f"{blk.prefix}.{blk.first}",
                                                          f"{blk.prefix}.{blk.first + 1}", blk.cin, blk.cout, imap, t1,
                                                          v1, training, n_upd, save)
            a2, recs[(blk.prefix, 2)] = self._conv_bn(P, a1, f"{blk.prefix}.{blk.first + 3}", f"{blk.prefix}.{blk.first + 4}",
                                                      blk.cout, blk.cout, None, CL(cat[i], cp, cp), xf[i][:, cp:],
                                                      training, n_upd, save)
            dskip.append(a2)
            dd, hh, ww = dd // 2, hh // 2, ww // 2
            pl = CL(torch.empty((n, dd, hh, ww, cp), dtype=adt, device=dev), 0, cp)
            ops.maxpool_fwd(a2, pl)
            pooled.append(pl)
            cur = pl
        # ---- centre block
        cb = plan.center
        cpc = pad8(cb.cout)
        center_out = None
        side = None
        if plan.center_live or training:
            live = plan.center_live
            # The generic UNet drops the centre block's output (models.py:241): in train mode it runs only to move its
            # BatchNorm buffers, two latency-bound 8^3 launches (~80 us) nothing downstream waits for -- on a forked
            # stream beside the decoder, joined at the end of forward (inside a captured graph: a fork/join branch)
            if not live and CENTER_SIDE:
                side = self.__dict__.get("_side")
                if side is None or side.device != dev:
                    side = self.__dict__["_side"] = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                c1 = CL(torch.empty((n, dd, hh, ww, cpc), dtype=adt, device=dev), 0, cpc)
                c2 = CL(torch.empty((n, dd, hh, ww, cpc), dtype=adt, device=dev), 0, cpc)
                v1 = torch.empty((4, cpc), dtype=torch.float32, device=dev)
                v2 = torch.empty((4, cpc), dtype=torch.float32, device=dev)
                a1, r1 = self._conv_bn(P, cur, f"{cb.prefix}.{cb.first}", f"{cb.prefix}.{cb.first + 1}", cb.cin, cb.cout,
                                       None, c1, v1, training, n_upd, save and live)
                a2, r2 = self._conv_bn(P, a1, f"{cb.prefix}.{cb.first + 3}", f"{cb.prefix}.{cb.first + 4}", cb.cout, cb.cout,
                                       None, c2, v2, training, n_upd, save and live)
            if live:
                recs[(cb.prefix, 1)], recs[(cb.prefix, 2)] = r1, r2
                center_out = a2
        cur = center_out if plan.center_live else pooled[-1]
        cur_segs = ((cb.cout if plan.center_live else plan.enc[-1].cout, 0),)
        # ---- decoder
        dec_in: List[CL] = []
        ups: List[CL] = []
        for j, blk in enumerate(plan.dec):
            i = nlev - 1 - j
            cp = pad8(blk.cout)
            dd, hh, ww = dd * 2, hh * 2, ww * 2
            ct = blk.cin                                  # ConvTranspose3d(C, C)
            ctp = pad8(ct)
            imap_t, cinv_t = self._maps(cur_segs, cur.cp, dev)
            wt = P[f"{blk.prefix}.0.weight"]
            dec_in.append(cur)
            t1 = CL(torch.empty((n, dd, hh, ww, cp), dtype=adt, device=dev), 0, cp)
            v1 = torch.empty((4, cp), dtype=torch.float32, device=dev)
            if self._fuse_up(cur, cp):
                ups.append(None)                      # never materialised in forward; backward recomputes it
                a1, recs[(blk.prefix, 1)] = self._upconv_bn(P, cur, blk.prefix, ct, blk.cout, cinv_t, t1, v1, training,
                                                            n_upd, save)
            else:
                up = CL(torch.empty((n, dd, hh, ww, ctp), dtype=adt, device=dev), 0, ctp)
                wpt = self._packed(f"{blk.prefix}.0", wt, "convt", cinv_t, cur.cp, ctp, 0)
                ops.convt_fwd(cur, wpt, P[f"{blk.prefix}.0.bias"].detach(), up)
                ups.append(up)
                a1, recs[(blk.prefix, 1)] = self._conv_bn(P, up, f"{blk.prefix}.1", f"{blk.prefix}.2", ct, blk.cout, None,
                                                          t1, v1, training, n_upd, save)
            a2, recs[(blk.prefix, 2)] = self._conv_bn(P, a1, f"{blk.prefix}.4", f"{blk.prefix}.5", blk.cout, blk.cout, None,
                                                      CL(cat[i], 0, cp), xf[i][:, :cp], training, n_upd, save)
            if plan.skip == "cat":               # free concat: both producers wrote channel slices of cat[i]
                cur = CL(cat[i], 0, 2 * cp, xf[i][0], xf[i][1], True)
                cur_segs = ((blk.cout, 0), (blk.cout, cp))
            elif plan.skip == "none":
                cur = CL(cat[i], 0, cp, xf[i][0][:cp], xf[i][1][:cp], True)
                cur_segs = ((blk.cout, 0),)
            else:                                # "add": the sum of two differently normalised tensors is materialised
                cur = CL(torch.empty((n, dd, hh, ww, cp), dtype=adt, device=dev), 0, cp)
                ops.skip_add(CL(cat[i], 0, cp, xf[i][0][:cp], xf[i][1][:cp], True),
                             CL(cat[i], cp, cp, xf[i][0][cp:], xf[i][1][cp:], True), cur)
                cur_segs = ((blk.cout, 0),)
        # ---- head
        imap_h, _ = self._maps(cur_segs, cur.cp, dev)
        wl, bl = P[plan.head + ".weight"], P[plan.head + ".bias"]
        w2 = wl.detach().reshape(wl.shape[0], wl.shape[1])
        out0, out1 = ops.head_fwd(cur, w2, bl.detach(), imap_h, plan.act, plan.head_mode)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        if save:
            ctx.update(recs=recs, x_cl=x_cl, cat=cat, xf=xf, pooled=pooled, dskip=dskip, dec_in=dec_in, ups=ups,
                       head_in=cur, imap_h=imap_h, center_out=center_out, dims=(n, d, h, w))
        return out0, out1, ctx

    # ------------------------------------------------------------------ backward pieces
    def _replay(self, P, rec: _ConvRec):
        # use_checkpoint=True: the recompute in backward repeats every live BN's running-stat update
        # (models.py:232-255) -- folded into this BN's backward finalize
        return ((P[rec.bn + ".running_mean"], P[rec.bn + ".running_var"], BN_MOMENTUM, BN_EPS,
                 P.get(rec.bn + ".num_batches_tracked")) if self._replay_stats else None)

    def _bwd_fin(self, P, rec: _ConvRec, device):
        """fin= of ops.maxpool_bwd / ops.head_bwd: their launch also finalizes rec's BatchNorm backward (None: BN_TAIL off)."""
        counter = self._counter(rec.bn, "bwd", device)
        return None if counter is None else (P[rec.bn + ".weight"].detach(), rec.cout, self._replay(P, rec), counter)

    def _gy_like(self, ga: CL) -> CL:
        """Where a lazy BatchNorm backward writes the raw-output gradient: a buffer of ga's shape (one per ga buffer and
        backward pass -- the two convs that share a concat-level gradient buffer use disjoint channel halves of it)."""
        ent = self._gy_bufs.get(id(ga.buf))          # (the entry keeps ga's buffer alive, so its id cannot be recycled)
        if ent is None:
            ent = self._gy_bufs[id(ga.buf)] = (ga.buf, torch.empty_like(ga.buf))
        return CL(ent[1], ga.c0, ga.cp)

    def _conv_bn_bwd(self, P, rec: _ConvRec, ga: CL, gin: Optional[CL], grads: Dict[str, torch.Tensor], ws, part,
                     pre_reduced: Optional[int] = None, finalized=None, up_in: Optional[CL] = None):
        """ga: gradient w.r.t. the ACTIVATED output (overwritten with the raw-output gradient unless the BatchNorm backward
        is folded into the weight-gradient kernel, ops.conv3d_wgrad_bn: then the raw-output gradient goes to a second buffer).
        gin: where to write the gradient w.r.t. this conv's activated input (None: not needed).
        up_in: rec is a fused up-convolution and this is its coarse input -- only the BatchNorm part runs here; returns what
        the caller hands to ops.upconv_fused_wgrad(lazy=...) (None: ga already holds the raw-output gradient)."""
        k = self.plan.k
        can = LAZY_BN and ga.cs == rec.y.cs and (not ga.lp or LAZY_BN_LP)
        if up_in is not None:
            lazy = can and (ops.lp_upconv_fused_wgrad_bn_supported(up_in.dims, up_in.cp) if ga.lp else
                            ops.upconv_fused_wgrad_bn_supported(up_in.dims, up_in.cp, ga.cp))
        elif rec.first is not None:
            lazy = can and not rec.bias and not ga.lp
        else:
            lazy = can and not rec.bias and rec.x is not None and ops.conv3d_wgrad_bn_supported(ga.dims, k, rec.x.cp, ga.cp, self.dtype)
        res = ops.bn_relu_bwd(rec.y, ga, rec.vec, P[rec.bn + ".weight"].detach(), rec.cout, part, self._replay(P, rec),
                              pre_reduced, self._counter(rec.bn, "bwd", ga.buf.device), finalized, lazy)
        grads[rec.bn + ".weight"], grads[rec.bn + ".bias"] = res[0], res[1]
        if up_in is not None:
            return (rec.y, rec.vec, res[2], self._gy_like(ga)) if lazy else None
        if rec.first is not None:                  # direct C_in <= 2 kernels; gin is a request flag here
            if lazy:
                gy = self._gy_like(ga)
                grads[rec.conv + ".weight"] = ops.conv_first_wgrad_bn(rec.first, ga, rec.y, rec.vec, res[2], gy, rec.cout, ws)
                ga = gy
            else:
                grads[rec.conv + ".weight"] = ops.conv_first_wgrad(rec.first, ga, rec.cout, ws)
            return ops.conv_first_bwd_data(ga, P[rec.conv + ".weight"].detach(), rec.cin) if gin is not None else None
        if lazy:
            gy = self._gy_like(ga)
            grads[rec.conv + ".weight"] = ops.conv3d_wgrad_bn(rec.x, ga, rec.y, rec.vec, res[2], gy, rec.cout, rec.cin, k,
                                                             rec.imap, ws)
            ga = gy
        elif rec.x is not None:                    # (None: fused up-convolution, its weight gradients come from the caller)
            dw, dbias = ops.conv3d_wgrad(rec.x, ga, rec.cout, rec.cin, k, rec.imap, ws, rec.bias)
            grads[rec.conv + ".weight"] = dw
            if rec.bias:
                grads[rec.conv + ".bias"] = dbias
        if gin is not None:
            lay = ops.conv_layout(k, gin.cp, ga.dims[3], self.dtype, ga.cp)
            wpd = self._packed(rec.conv, P[rec.conv + ".weight"], "conv", rec.imap, ga.cp, gin.cp, 1, lay)
            ops.conv3d_fwd(ga, wpd, None, gin, k, None, (rec.cout, rec.cin), lay)

    def backward(self, P: Dict[str, torch.Tensor], ctx, g0: torch.Tensor, g1: Optional[torch.Tensor],
                 need_dx: bool, sync=None):
        """sync: optional parallel.GradSync -- receives each block's weight gradients as soon as their
        kernels are enqueued (head, decoder top->bottom, centre, encoder bottom->top)."""
        plan = self.plan
        recs = ctx["recs"]
        n, d, h, w = ctx["dims"]
        dev = g0.device
        nlev = len(plan.enc)
        # float16 gradients: scale what comes in (the per-voxel gradient of a mean-reduced loss underflows fp16), un-scale
        # every parameter gradient and dx on the way out; bf16 / fp32 need none
        gs = 1.0
        flag = None
        if self.dtype == torch.float16:
            gs = self.loss_scale if self.loss_scale else float(2 ** max(0, (n * d * h * w).bit_length() - 5))
            # (the head backward multiplies g0 / g1 by gs as it reads them: no scaled copy of the output-sized maps)
            # overflow guard: the un-scaling launches below set this flag when a gradient came out inf / NaN (a static loss
            # scale can overflow the 16-bit activation gradients); the fused optimizer skips the step when it is set
            flag = self.overflow_flag(dev)
            flag.zero_()
        grads: Dict[str, torch.Tensor] = {}
        emitted: set = set()
        self._gy_bufs: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}

        def emit():
            """Block boundary: the gradients produced since the last one are final -- un-scale them (float16 loss
            scaling) and hand them to the gradient exchange."""
            new = [(nm, g) for nm, g in grads.items() if nm not in emitted and g is not None]
            emitted.update(nm for nm, _ in new)
            if gs != 1.0 and new:
                ops.scale_tensors([g for _, g in new], 1.0 / gs, flag)
            if sync is not None:
                sync.push(new)
        k = plan.k
        # workspaces sized for the largest layer
        ws_n, part_n = 1, 1
        for r in recs.values():
            if r is None:
                continue
            if r.first is not None:
                ws_n = max(ws_n, ops.conv_first_wgrad_ws(r.y.dims, r.cin))
                part_n = max(part_n, ops.bn_bwd_partials_floats(r.y.nvox, r.y.cp))
                continue
            ws_n = max(ws_n, ops.conv3d_wgrad_ws(r.y.dims, k, r.x.cp if r.x is not None else pad8(r.cin), r.y.cp, self.dtype))
            part_n = max(part_n, ops.bn_bwd_partials_floats(r.y.nvox, r.y.cp))
        for x_in, blk in zip(ctx["dec_in"], plan.dec):
            ws_n = max(ws_n, ops.convt_wgrad_ws(x_in.dims, x_in.cp, pad8(blk.cin), self.dtype))
        for dsk in ctx["dskip"]:
            part_n = max(part_n, ops.maxpool_bwd_bn_blocks(dsk.dims, dsk.cp) * 2 * dsk.cp)
        part_n = max(part_n, ops.head_bwd_blocks(ctx["head_in"].dims) * 2 * pad8(plan.dec[-1].cout))
        ws = torch.empty(ws_n, dtype=torch.float32, device=dev)
        part = torch.empty(part_n, dtype=torch.float32, device=dev)

        # the second running-stat update that torch.utils.checkpoint's recompute performs
        # (models.py:232-255) rides on each live BN's backward finalize; the dead centre block is
        # never recomputed and has no backward.
        self._replay_stats = bool(ctx["training"] and ctx["chk"])

        cat, head_in = ctx["cat"], ctx["head_in"]
        gcat = [torch.empty_like(c) for c in cat]
        wl, bl = P[plan.head + ".weight"], P[plan.head + ".bias"]
        w2 = wl.detach().reshape(wl.shape[0], wl.shape[1])
        def g_skip_target(level: int) -> CL:
            """Where the gradient w.r.t. the tensor a decoder level hands on (cat / sum / plain output) is written."""
            full = gcat[level].shape[-1]
            return CL(gcat[level], 0, full if plan.skip == "cat" else full // 2)

        def g_skip_fanout(level: int):
            """additive skip: d(sum) reaches both addends -- copy it into the encoder half before the decoder's
            BatchNorm backward overwrites the first half in place"""
            if plan.skip == "add":
                half = gcat[level].shape[-1] // 2
                ops.skip_add(CL(gcat[level], 0, half), None, CL(gcat[level], half, half))

        # the head's input gradient completes the activated-output gradient of the last decoder conv: its BatchNorm-backward
        # reduction rides on the head backward (as the encoder ones ride on the max-pool backward)
        r_last = recs[(plan.dec[-1].prefix, 2)]
        head_rows, head_fin = None, None
        if (POOL_BN and head_in.scale is not None and head_in.relu and head_in.scale.data_ptr() == r_last.vec[0].data_ptr()
                and head_in.c0 == r_last.y.c0 and head_in.buf is r_last.y.buf):
            fin = self._bwd_fin(P, r_last, head_in.buf.device)
            res = ops.head_bwd(head_in, w2, bl.detach(), ctx["imap_h"], plan.act, plan.head_mode, g0.contiguous(),
                               None if g1 is None else g1.contiguous(), g_skip_target(0), (r_last.vec, part), fin, gscale=gs)
            dwl, dbl, head_rows = res[:3]
            head_fin = res[3] if fin is not None else None
        else:
            dwl, dbl = ops.head_bwd(head_in, w2, bl.detach(), ctx["imap_h"], plan.act, plan.head_mode, g0.contiguous(),
                                    None if g1 is None else g1.contiguous(), g_skip_target(0), gscale=gs)
        g_skip_fanout(0)
        grads[plan.head + ".weight"], grads[plan.head + ".bias"] = dwl.reshape(wl.shape), dbl
        emit()

        g_deep: Optional[CL] = None
        for j in range(nlev - 1, -1, -1):           # decoder blocks, top level first
            blk = plan.dec[j]
            i = nlev - 1 - j
            cp = pad8(blk.cout)
            r1, r2 = recs[(blk.prefix, 1)], recs[(blk.prefix, 2)]
            g_u2 = CL(gcat[i], 0, cp)
            g_u1 = CL(torch.empty_like(r1.y.buf), 0, cp)
            self._conv_bn_bwd(P, r2, g_u2, g_u1, grads, ws, part, head_rows if j == nlev - 1 else None,
                              head_fin if j == nlev - 1 else None)
            x_in = ctx["dec_in"][j]
            ct = blk.cin
            if j > 0 and plan.skip == "cat":
                segs = ((plan.dec[j - 1].cout, 0), (plan.dec[j - 1].cout, pad8(plan.dec[j - 1].cout)))
            else:
                segs = ((ct, 0),)
            imap_t, cinv_t = self._maps(segs, x_in.cp, dev)
            wt = P[f"{blk.prefix}.0.weight"]
            up = ctx["ups"][j]
            if j > 0:
                gin = g_skip_target(i + 1)
            else:
                gin = CL(torch.empty_like(x_in.buf), 0, x_in.cp)
                g_deep = gin
            if up is None:
                # ConvTranspose3d -> Conv3d ran fused (the transposed conv's output never existed): BatchNorm backward,
                # then the gradients of BOTH layers' parameters from the composite-weight gradient of (coarse input, g_u1)
                # and the data gradient straight back to the coarse grid
                lz = self._conv_bn_bwd(P, r1, g_u1, None, grads, ws, part, up_in=x_in)
                gq = g_u1
                up_wgrad = ops.lp_upconv_fused_wgrad if g_u1.lp else ops.upconv_fused_wgrad      # 16-bit: upconv_lp.hip
                dwt, dbt, dw3 = up_wgrad(x_in, gq, ct, blk.cout, P[f"{blk.prefix}.0.bias"], self._up_cache[blk.prefix][4], imap_t, lz)
                if lz is not None:
                    gq = lz[3]                        # the raw-output gradient the weight-gradient kernel wrote
                grads[f"{blk.prefix}.1.weight"] = dw3
                grads[f"{blk.prefix}.0.weight"], grads[f"{blk.prefix}.0.bias"] = dwt, dbt
                if g_u1.lp:
                    ops.lp_upconv_fused_bwd_data(gq, self._up_cache[blk.prefix][5], gin, (ct, blk.cout))
                else:
                    ops.upconv_fused_bwd_data(gq, self._up_cache[blk.prefix][5], gin, (ct, blk.cout))
            else:
                g_up = CL(torch.empty_like(up.buf), 0, up.cp)
                self._conv_bn_bwd(P, r1, g_u1, g_up, grads, ws, part)
                dwt, dbt = ops.convt_wgrad(x_in, g_up, ct, ct, imap_t, ws)
                grads[f"{blk.prefix}.0.weight"], grads[f"{blk.prefix}.0.bias"] = dwt, dbt
                wpd = self._packed(f"{blk.prefix}.0", wt, "convt", cinv_t, g_up.cp, x_in.cp, 1)
                ops.convt_bwd_data(g_up, wpd, gin)
            if j > 0:
                g_skip_fanout(i + 1)
            emit()
        g_pool = g_deep
        if plan.center_live:
            cb = plan.center
            r1, r2 = recs[(cb.prefix, 1)], recs[(cb.prefix, 2)]
            g_c1 = CL(torch.empty_like(r1.y.buf), 0, r1.y.cp)
            self._conv_bn_bwd(P, r2, g_deep, g_c1, grads, ws, part)
            g_pool = CL(torch.empty_like(ctx["pooled"][-1].buf), 0, ctx["pooled"][-1].cp)
            self._conv_bn_bwd(P, r1, g_c1, g_pool, grads, ws, part)
            emit()
        dx = None
        for i in range(nlev - 1, -1, -1):           # encoder blocks, deepest first
            blk = plan.enc[i]
            cp = pad8(blk.cout)
            r1, r2 = recs[(blk.prefix, 1)], recs[(blk.prefix, 2)]
            g_d2 = CL(gcat[i], cp, cp)
            # accumulate onto the skip's gradient; g_d2 is then complete, so the pass also carries the reduction of r2's
            # BatchNorm backward (the pooled tensor is that BatchNorm's activated output)
            dsk = ctx["dskip"][i]
            fuse = (POOL_BN and dsk.scale is not None and dsk.relu and dsk.scale.data_ptr() == r2.vec[0].data_ptr()
                    and ops.maxpool_bwd_bn_blocks(dsk.dims, dsk.cp) > 0)
            fin = self._bwd_fin(P, r2, dsk.buf.device) if fuse else None
            nb = ops.maxpool_bwd(dsk, g_pool, g_d2, plan.skip != "none", (r2.vec, part) if fuse else None, fin)
            nb, done = nb if fin is not None else (nb, None)
            g_d1 = CL(torch.empty_like(r1.y.buf), 0, cp)
            self._conv_bn_bwd(P, r2, g_d2, g_d1, grads, ws, part, nb, done)
            if i > 0:
                g_pool = CL(torch.empty_like(ctx["pooled"][i - 1].buf), 0, ctx["pooled"][i - 1].cp)
                self._conv_bn_bwd(P, r1, g_d1, g_pool, grads, ws, part)
            elif need_dx and r1.first is not None:
                dx = self._conv_bn_bwd(P, r1, g_d1, True, grads, ws, part)
            elif need_dx:
                g_x = CL(torch.empty_like(ctx["x_cl"].buf), 0, ctx["x_cl"].cp)
                self._conv_bn_bwd(P, r1, g_d1, g_x, grads, ws, part)
                dx = ops.cl_to_ncdhw(g_x, plan.in_ch)
            else:
                self._conv_bn_bwd(P, r1, g_d1, None, grads, ws, part)
            emit()
        emit()
        if gs != 1.0 and dx is not None:
            ops.scale_tensors([dx], 1.0 / gs, flag)
        if sync is not None:
            if flag is not None:
                # N > 1: the overflow flag travels with the last gradient bucket, so that every rank skips the step when ANY
                # rank overflowed (the mean of the ranks' 0 / 1 flags); the caller copies grads["__overflow__"] back into
                # overflow_flag() once the exchange has completed (fold_overflow)
                sync.push([("__overflow__", flag.clone())])
            grads.update(sync.finish())
        self._gy_bufs = {}
        return grads, dx

    def fold_overflow(self, grads: Dict[str, torch.Tensor], device) -> None:
        """After a gradient exchange: the ranks' combined overflow flag (see backward) replaces this rank's own."""
        red = grads.pop("__overflow__", None)
        if red is not None:
            self.overflow_flag(device).copy_(red.reshape(1))


# ----------------------------------------------------------------------------- autograd glue
class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, module, engine, names, x_req, x, *params):
        P = _tensor_dict(module)
        out0, out1, ctx = engine.forward(P, x, module.training, True, bool(getattr(module, "chk", False)))
        fctx.engine, fctx.module, fctx.names, fctx.ctx = engine, module, names, ctx
        fctx.two = out1 is not None
        fctx.x_req = x_req
        if out1 is None:
            return out0
        return out0, out1

    @staticmethod
    def backward(fctx, *gouts):
        if fctx.ctx is None:
            raise RuntimeError("ctunet_amd: backward through the same forward twice is not supported")
        g0 = gouts[0]
        g1 = gouts[1] if fctx.two else None
        ctx = fctx.ctx
        if g0 is None:
            g0 = torch.zeros(_out_shape(ctx, fctx, 0), device=ctx["cat"][0].device)
        if fctx.two and g1 is None:
            g1 = torch.zeros_like(g0)
        P = _tensor_dict(fctx.module)
        from .parallel import make_sync
        with torch.no_grad():
            grads, dx = fctx.engine.backward(P, ctx, g0, g1, fctx.x_req, make_sync(fctx.module))
            fctx.engine.fold_overflow(grads, g0.device)        # (GradSync.finish has waited for every bucket)
        fctx.ctx = None
        # parameters the graph never touches (the dead centre block, models.py:241) get None,
        # exactly as torch autograd leaves them in the reference
        return (None, None, None, None, dx) + tuple(grads.get(nm) for nm in fctx.names)


def _out_shape(ctx, fctx, idx):
    n, d, h, w = ctx["dims"]
    c = 2 if fctx.two else fctx.engine.plan.out_ch
    return (n, c, d, h, w)


def _tensor_dict(module) -> Dict[str, torch.Tensor]:
    P = dict(module.named_parameters())
    P.update(dict(module.named_buffers()))
    return P


def run_network(module, engine: UNetEngine, x: torch.Tensor):
    if not isinstance(x, torch.Tensor) or x.dim() != 5:
        raise RuntimeError("ctunet_amd: expected a [N, C, D, H, W] tensor")
    if not x.is_cuda:
        raise RuntimeError("ctunet_amd: this model runs on the MI355X only (input is on %s); there is no CPU path"
                           % x.device)
    if x.dtype != torch.float32:
        raise RuntimeError(f"ctunet_amd: fp32 input expected, got {x.dtype}")
    names, params = zip(*module.named_parameters())
    for p in params:
        if p.device != x.device:
            raise RuntimeError("ctunet_amd: module parameters and input must be on the same GPU (call .to(device))")
    need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    if not need_grad:
        with torch.no_grad():
            out0, out1, _ = engine.forward(_tensor_dict(module), x, module.training, False, False)
        return out0 if out1 is None else (out0, out1)
    return _UNetFn.apply(module, engine, names, x.requires_grad, x, *params)
