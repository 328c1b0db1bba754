"""Drop-in replacements for the model classes of ``ctunet.pytorch.models``.

Same class names, constructor signatures, ``state_dict`` keys/shapes, default initialisation
(so ``torch.manual_seed`` gives identical weights) and NCDHW-contiguous outputs as the
reference -- but ``forward`` runs the hand-written gfx950 kernels of libctunet_hip.so through
``engine.UNetEngine`` instead of a torch.nn graph.  The ``torch.nn`` objects inside are
parameter holders only (they give the reference's key names); none of them is ever called.

Reference: /root/reference/ctunet/pytorch/models.py
  UNetBlock :9-49, CenterBlock :52-97, UNet :158-261, UNet4b2i3o/5b2i3o/4b1i3o :272-296,
  UNetSP :299-330, UNetSPSmall :333-365, UNetDO :368-387, down/up_block_cr :393-438,
  recAE_v2_fixed :441-538, UNet4_2IC :541-557.

Not supported (dead or unreachable in the reference): ``residual=True``,
``fc_layer`` -- both crash in the reference itself -- and ``dropout_p>0`` (no shipped class sets
it; its random mask could not be compared with the reference's anyway); they raise
``NotImplementedError`` here instead of silently computing something else.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn
from torch.nn import init

from .engine import BlockPlan, NetPlan, UNetEngine, run_network


# ----------------------------------------------------------------------------- holders
class _ConvParams(nn.Module):
    """weight [Co,Ci,k,k,k] (+bias): initialised exactly like nn.Conv3d.reset_parameters."""

    def __init__(self, cin, cout, k, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            init.uniform_(self.bias, -bound, bound)


class _ConvTParams(nn.Module):
    """ConvTranspose3d(C, C, 2, 2) parameters: weight [Ci,Co,2,2,2] + bias."""

    def __init__(self, cin, cout, k=2):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, k, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        init.uniform_(self.bias, -bound, bound)


class _BNParams(nn.Module):
    """BatchNorm3d state: gamma/beta + running_mean/var/num_batches_tracked (eps 1e-5, momentum 0.1)."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Slot(nn.Module):
    """Index placeholder for the parameter-free ReLU / Dropout3d entries of the reference's Sequentials."""


def _double_conv(cin, cout, k, bias, first_extra: Optional[nn.Module] = None) -> nn.Sequential:
    mods: List[nn.Module] = [] if first_extra is None else [first_extra]
    mods += [_ConvParams(cin, cout, k, bias), _BNParams(cout), _Slot(),
             _ConvParams(cout, cout, k, bias), _BNParams(cout), _Slot(), _Slot()]
    return nn.Sequential(*mods)


class UNetBlock(nn.Module):
    """Parameters of one encoder / decoder block (reference ``UNetBlock``, models.py:9-49)."""

    def __init__(self, in_c, out_c, kern_s_conv=5, kern_s_uconv=2, pad=2, stride_c=1, stride_upc=2, dropout_p=0,
                 up_block=False):
        super().__init__()
        _check_geometry(kern_s_conv, pad, stride_c, dropout_p)
        if up_block and (kern_s_uconv != 2 or stride_upc != 2):
            raise NotImplementedError("ctunet_amd: only ConvTranspose3d(kernel 2, stride 2) is implemented")
        self.block = _double_conv(in_c, out_c, kern_s_conv, False, _ConvTParams(in_c, in_c) if up_block else None)


class CenterBlock(nn.Module):
    """Parameters of the centre block (reference ``CenterBlock``, models.py:52-97; conv form only)."""

    def __init__(self, input_channels, output_channels, kern_sz_conv, padding, dropout_p, fc_block=False):
        super().__init__()
        if fc_block:
            raise NotImplementedError("ctunet_amd: CenterBlock(fc_block=...) is dead code in the reference "
                                      "(channel plan mismatch, SURVEY 2.1) and is not implemented")
        _check_geometry(kern_sz_conv, padding, 1, dropout_p)
        self.block = _double_conv(input_channels, output_channels, kern_sz_conv, False)


def _check_geometry(k, pad, stride, dropout_p):
    if k not in (3, 5) or pad != (k - 1) // 2 or stride != 1:
        raise NotImplementedError(f"ctunet_amd: conv kernel {k} / padding {pad} / stride {stride} not implemented "
                                  "(the shipped classes use k3 p1 or k5 p2, stride 1)")
    if dropout_p != 0:
        raise NotImplementedError("ctunet_amd: Dropout3d with p > 0 is not implemented (p = 0 in every shipped class)")


class _HipNet(nn.Module):
    """Common forward plumbing: plan -> engine -> kernels."""

    _plan: NetPlan

    def _engine(self) -> UNetEngine:
        eng = self.__dict__.get("_eng")
        dt = self.__dict__.get("_act_dtype", torch.float32)
        if eng is None or eng.dtype != dt:
            eng = UNetEngine(self._plan, dt, self.__dict__.get("_loss_scale"))
            self.__dict__["_eng"] = eng       # not a submodule / not in state_dict
        return eng

    def set_precision(self, dtype=torch.float32, loss_scale=None):
        """Storage type of the activations between the kernels: ``torch.float32`` (default: the reference's arithmetic,
        1e-4 parity), ``torch.bfloat16`` or ``torch.float16`` (BASELINE configs 4 / 5: 16-bit tensors in HBM,
        v_mfma_f32_16x16x32 with fp32 accumulation, fp32 BatchNorm statistics, fp32 master weights / gradients /
        optimizer -- what ``torch.autocast`` would give the reference).  Inputs, outputs, parameters and their gradients
        stay float32 either way.  ``loss_scale`` (float16 only): see ``UNetEngine``.  Accepts the strings
        "fp32" / "bf16" / "fp16" too.  Returns self."""
        names = {"fp32": torch.float32, "f32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16}
        dt = names.get(dtype, dtype) if isinstance(dtype, str) else dtype
        if dt not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError(f"ctunet_amd: unsupported precision {dtype}")
        self.__dict__["_act_dtype"] = dt
        self.__dict__["_loss_scale"] = loss_scale
        eng = self.__dict__.get("_eng")
        if eng is not None and eng.dtype == dt:
            eng.loss_scale = loss_scale            # (the engine is only rebuilt when the storage type changes)
        return self

    def overflow_flag(self) -> torch.Tensor:
        """float32[1] on the model's GPU: 1 after a float16 backward whose un-scaled parameter gradients contained inf / NaN
        (the static loss scale overflowed the 16-bit activation gradients), 0 otherwise; cleared at the start of every float16
        backward.  Hand it to the fused optimizer (``optim.Adam.guard(model)``): an overflowed step then changes nothing --
        parameters, moments and step counter -- also inside a replayed HIP graph.  Call again after ``set_precision``."""
        return self._engine().overflow_flag(next(self.parameters()).device)

    def overflowed(self) -> bool:
        """Did the last float16 backward overflow?  (Synchronises; a trainer lowers ``loss_scale`` when it says yes.)"""
        return bool(self.overflow_flag().item() != 0)

    def _run(self, x):
        return run_network(self, self._engine(), x)


class UNet(_HipNet):
    """Generic n-block 3D U-Net (reference ``UNet``, models.py:158-261)."""

    def __init__(self, input_channels=1, out_channels=2, n_blocks=4, kern_sz_conv=3, kern_sz_upconv=2, stride_conv=1,
                 stride_upconv=2, i_size=8, padding=1, dropout_p=0, use_checkpoint=True, fc_layer=None,
                 use_skip_connections=True, apply_softmax=False, apply_sigmoid=True, cat=True, residual=False):
        super().__init__()
        if residual:
            raise NotImplementedError("ctunet_amd: residual=True crashes in the reference itself (SURVEY 2.1)")
        if fc_layer:
            raise NotImplementedError("ctunet_amd: fc_layer is dead code in the reference (SURVEY 2.1)")
        if out_channels > 4:
            raise NotImplementedError("ctunet_amd: the output head supports at most 4 channels")
        self.chk = use_checkpoint
        self.skip = use_skip_connections
        self.apply_softmax = apply_softmax
        self.apply_sigmoid = apply_sigmoid
        self.fc_layer = fc_layer
        self.cat = cat

        n = n_blocks
        widths = [i_size * 2 ** i for i in range(n + 1)]
        self.d_blocks = nn.ModuleList(
            UNetBlock(input_channels if i == 0 else widths[i - 1], widths[i], kern_sz_conv, 0, padding, stride_conv, 0,
                      dropout_p) for i in range(n))
        self.cblock = CenterBlock(widths[n - 1], widths[n], kern_sz_conv, padding, dropout_p, fc_layer)
        ups = []
        for i in range(n - 1, -1, -1):
            # deepest up-block takes the pooled encoder output (NOT the centre block, models.py:241);
            # the others take [previous up-block | encoder skip] = 4 * widths[i] channels.
            # (models.py:209-217: half of that when the skip is added instead of concatenated, or absent)
            c_in = widths[i] if i == n - 1 else (4 if (use_skip_connections and cat) else 2) * widths[i]
            ups.append(UNetBlock(c_in, widths[i], kern_sz_conv, kern_sz_upconv, padding, stride_conv, stride_upconv,
                                 dropout_p, True))
        self.u_blocks = nn.ModuleList(ups)
        self.last_conv = _ConvParams(2 * i_size if (use_skip_connections and cat) else i_size, out_channels, 1, True)

        enc = [BlockPlan(f"d_blocks.{i}.block", 0, input_channels if i == 0 else widths[i - 1], widths[i])
               for i in range(n)]
        fan = 4 if (use_skip_connections and cat) else 2
        dec = [BlockPlan(f"u_blocks.{j}.block", 1, widths[n - 1 - j] if j == 0 else fan * widths[n - 1 - j],
                         widths[n - 1 - j]) for j in range(n)]
        self._plan = NetPlan(k=kern_sz_conv, conv_bias=False, in_ch=input_channels, out_ch=out_channels, enc=enc,
                             center=BlockPlan("cblock.block", 0, widths[n - 1], widths[n]), center_live=False, dec=dec,
                             head="last_conv", act=(1 if apply_softmax else 0) | (2 if apply_sigmoid else 0),
                             head_mode=0, skip="none" if not use_skip_connections else ("cat" if cat else "add"))

    def forward(self, x):
        return self._run(x)


class UNet4b2i3o(UNet):
    """Three-channel output UNet with Shape Priors (models.py:272-278)."""

    def __init__(self):
        super().__init__(i_size=7, input_channels=2, out_channels=3, use_checkpoint=True)


class UNet5b2i3o(UNet):
    """models.py:281-287."""

    def __init__(self):
        super().__init__(i_size=4, input_channels=2, out_channels=3, n_blocks=5, use_checkpoint=True)


class UNet4b1i3o(UNet):
    """models.py:290-296."""

    def __init__(self):
        super().__init__(i_size=7, input_channels=1, out_channels=3, use_checkpoint=True)


class _SPHead:
    """(bg, flap, full) -> ([bg, flap+full], [1-flap, flap]) fused into the head kernel (models.py:317-330)."""
    _head_mode = 1

    def _set_head(self):
        self._plan.head_mode = self._head_mode


class UNetSP(UNet4b2i3o, _SPHead):
    def __init__(self):
        super().__init__()
        self._set_head()


class UNetSPSmall(UNet5b2i3o, _SPHead):
    _head_mode = 2          # + softmax of each pair (models.py:364-365)

    def __init__(self):
        super().__init__()
        self._set_head()


class UNetDO(UNet4b1i3o, _SPHead):
    def __init__(self):
        super().__init__()
        self._set_head()


# ----------------------------------------------------------------------------- legacy
def down_block_cr(in_c, out_c, kern_s, pad, dropout_p=0.5):
    """Parameters of the legacy encoder block: k5 convs WITH bias (models.py:393-411)."""
    _check_geometry(kern_s, pad, 1, dropout_p)
    return _double_conv(in_c, out_c, kern_s, True)


def up_block_cr(in_c, out_c, kern_s_conv, kern_s_uconv, pad, stride_uc, dropout_p=0.5):
    """Parameters of the legacy decoder block (models.py:414-438)."""
    _check_geometry(kern_s_conv, pad, 1, dropout_p)
    if kern_s_uconv != 2 or stride_uc != 2:
        raise NotImplementedError("ctunet_amd: only ConvTranspose3d(kernel 2, stride 2) is implemented")
    return _double_conv(in_c, out_c, kern_s_conv, True, _ConvTParams(in_c, in_c))


class recAE_v2_fixed(_HipNet):
    """Legacy fixed 4-level U-Net: k5 p2 convs with bias, live centre block, softmax output
    (reference ``recAE_v2_fixed``, models.py:441-538)."""

    def __init__(self, input_channels=1, kern_sz_conv=5, kern_sz_upconv=2, stride_upconv=2, i_size=8, padding=2,
                 dropout_p=0, use_checkpoint=True):
        super().__init__()
        self.chk = use_checkpoint
        fms = [i_size * 2 ** n for n in range(5)]
        k, p = kern_sz_conv, padding
        self.dblock1 = down_block_cr(input_channels, fms[0], kern_s=k, pad=p, dropout_p=dropout_p)
        self.dblock2 = down_block_cr(fms[0], fms[1], kern_s=k, pad=p, dropout_p=dropout_p)
        self.dblock3 = down_block_cr(fms[1], fms[2], kern_s=k, pad=p, dropout_p=dropout_p)
        self.dblock4 = down_block_cr(fms[2], fms[3], kern_s=k, pad=p, dropout_p=dropout_p)
        self.cblock_center = _double_conv(fms[3], fms[4], k, True)
        self.ublock1 = up_block_cr(fms[4], fms[3], k, kern_sz_upconv, p, stride_upconv, dropout_p)
        self.ublock2 = up_block_cr(2 * fms[3], fms[2], k, kern_sz_upconv, p, stride_upconv, dropout_p)
        self.ublock3 = up_block_cr(2 * fms[2], fms[1], k, kern_sz_upconv, p, stride_upconv, dropout_p)
        self.ublock4 = up_block_cr(2 * fms[1], fms[0], k, kern_sz_upconv, p, stride_upconv, dropout_p)
        self.last_conv = _ConvParams(2 * fms[0], 2, 1, True)

        enc = [BlockPlan(f"dblock{i + 1}", 0, input_channels if i == 0 else fms[i - 1], fms[i]) for i in range(4)]
        dec = [BlockPlan(f"ublock{j + 1}", 1, fms[4] if j == 0 else 2 * fms[4 - j], fms[3 - j]) for j in range(4)]
        self._plan = NetPlan(k=k, conv_bias=True, in_ch=input_channels, out_ch=2, enc=enc,
                             center=BlockPlan("cblock_center", 0, fms[3], fms[4]), center_live=True, dec=dec,
                             head="last_conv", act=1, head_mode=0)

    def forward(self, x):
        return self._run(x)


class UNet4_2IC(recAE_v2_fixed):
    """models.py:541-557."""

    def __init__(self):
        super().__init__(i_size=7, input_channels=2)
