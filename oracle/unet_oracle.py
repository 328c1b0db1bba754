"""CPU oracle for the ctunet 3D U-Net hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``ct-unet_amd/`` imports it, and the product path
raises when the HIP library is missing instead of falling back here.

It restates, as flat functions over a ``state_dict``, the graph that the
reference assembles from ``torch.nn`` modules.  The arithmetic itself lives in
PyTorch ATen (third-party, unpinned in the reference's ``setup.py:6``); here it
is reached through ``torch.nn.functional`` on CPU tensors, torch 2.10.0.

Parity pin: ``tests/golden/make_golden.py`` imports the reference's
``ctunet/pytorch/models.py`` / ``utilities.py`` / ``ProblemHandler.py`` in the
build container and stores outputs, gradients and post-step buffers in
``tests/golden/*.npz|json``; ``tests/test_oracle_golden.py`` checks this file
against them.  The reference ships no tests or known-answer vectors of its own
(``tox.ini`` points at a missing ``tests/``), so those fixtures are the pin.

Reference lines followed (all under /root/reference):
  UNetBlock / CenterBlock ....... ctunet/pytorch/models.py:9-49, 52-97
  UNet.__init__ channel plan .... ctunet/pytorch/models.py:175-224
  UNet.forward .................. ctunet/pytorch/models.py:226-261
  UNetSP/UNetSPSmall/UNetDO ..... ctunet/pytorch/models.py:299-387
  down_block_cr / up_block_cr ... ctunet/pytorch/models.py:393-438
  recAE_v2_fixed.forward ........ ctunet/pytorch/models.py:509-538
  dice_loss ..................... ctunet/utilities.py:35-50
  comp_losses_metrics (single) .. ctunet/pytorch/ProblemHandler.py:44-102
  comp_losses_metrics (double) .. ctunet/pytorch/ProblemHandler.py:213-309
  hard_segm_from_tensor ......... ctunet/utilities.py:103-124
  dice_coeff (monai, absent) .... ctunet/utilities.py:53-59  -> "parity unpinned"
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# network description
# --------------------------------------------------------------------------
@dataclass
class NetSpec:
    """Architecture constants of one of the reference's model classes."""

    family: str = "generic"        # "generic" (UNet) | "legacy" (recAE_v2_fixed)
    in_ch: int = 1
    out_ch: int = 2
    n_blocks: int = 4
    i_size: int = 8
    k: int = 3
    pad: int = 1
    skip: bool = True
    cat: bool = True
    apply_softmax: bool = False
    apply_sigmoid: bool = True
    head: str = "plain"            # "plain" | "sp" (UNetSP/UNetDO) | "sp_softmax" (UNetSPSmall)


SPECS: Dict[str, NetSpec] = {
    # models.py:175-180 defaults
    "UNet": NetSpec(),
    # models.py:272-296
    "UNet4b2i3o": NetSpec(in_ch=2, out_ch=3, i_size=7),
    "UNet5b2i3o": NetSpec(in_ch=2, out_ch=3, i_size=4, n_blocks=5),
    "UNet4b1i3o": NetSpec(in_ch=1, out_ch=3, i_size=7),
    # models.py:299-387
    "UNetSP": NetSpec(in_ch=2, out_ch=3, i_size=7, head="sp"),
    "UNetSPSmall": NetSpec(in_ch=2, out_ch=3, i_size=4, n_blocks=5, head="sp_softmax"),
    "UNetDO": NetSpec(in_ch=1, out_ch=3, i_size=7, head="sp"),
    # models.py:441-557
    "recAE_v2_fixed": NetSpec(family="legacy", in_ch=1, out_ch=2, i_size=8, k=5, pad=2),
    "UNet4_2IC": NetSpec(family="legacy", in_ch=2, out_ch=2, i_size=7, k=5, pad=2),
}


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def _bn(x, sd, prefix, training, stat_updates):
    """BatchNorm3d, eps 1e-5, momentum 0.1."""
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training and stat_updates is not None:
        # F.batch_norm updates rm/rv in place; num_batches_tracked is the
        # module's job in torch, so it is restated here.
        y = F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"],
                         True, BN_MOMENTUM, BN_EPS)
        sd[prefix + ".num_batches_tracked"] += 1
        stat_updates.append(prefix)
        return y
    if training:
        return F.batch_norm(x, None, None, sd[prefix + ".weight"], sd[prefix + ".bias"],
                            True, BN_MOMENTUM, BN_EPS)
    return F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"],
                        False, BN_MOMENTUM, BN_EPS)


def _conv_bn_relu(x, sd, conv, bn, pad, training, stat_updates):
    y = F.conv3d(x, sd[conv + ".weight"], sd.get(conv + ".bias"), 1, pad)
    return F.relu(_bn(y, sd, bn, training, stat_updates))


def double_conv(x, sd, prefix, first, pad, training, stat_updates):
    """[Conv3d -> BN -> ReLU] x2, Sequential indices first,first+1 / first+3,first+4
    (models.py:25-34, 36-46, 70-82).  Dropout3d(p=0) is the identity."""
    y = _conv_bn_relu(x, sd, f"{prefix}.{first}", f"{prefix}.{first + 1}", pad, training, stat_updates)
    return _conv_bn_relu(y, sd, f"{prefix}.{first + 3}", f"{prefix}.{first + 4}", pad, training, stat_updates)


def up_block(x, sd, prefix, pad, training, stat_updates):
    """ConvTranspose3d(k2,s2,bias) then the double conv (models.py:36-46, 413-438)."""
    y = F.conv_transpose3d(x, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"], stride=2)
    return double_conv(y, sd, prefix, 1, pad, training, stat_updates)


def pool(x):
    """MaxPool3d(2, stride 2), indices discarded (models.py:190-191, 233)."""
    return F.max_pool3d(x, 2, 2)


def sp_head(y3):
    """(bg, flap, full) -> ([bg, flap+full], [1-flap, flap]) (models.py:317-330)."""
    bg, flap, full = y3[:, 0:1], y3[:, 1:2], y3[:, 2:3]
    return torch.cat((bg, flap + full), 1), torch.cat((1 - flap, flap), 1)


# --------------------------------------------------------------------------
# whole-network forward
# --------------------------------------------------------------------------
def forward(spec: NetSpec, sd: Dict[str, torch.Tensor], x: torch.Tensor,
            training: bool = False, update_stats: bool = True,
            return_logits: bool = False):
    """Forward of the network described by ``spec`` with parameters/buffers ``sd``.

    ``sd`` tensors that require grad make the result differentiable.  In training
    mode with ``update_stats`` the BN running buffers in ``sd`` are updated in
    place exactly once per BN (the reference's default ``use_checkpoint=True``
    applies that update twice per step, see ``bn_checkpoint_replay``).
    """
    upd: Optional[List[str]] = [] if (training and update_stats) else None
    if spec.family == "legacy":
        out = _forward_legacy(spec, sd, x, training, upd, return_logits)
    else:
        out = _forward_generic(spec, sd, x, training, upd, return_logits)
    return out


def _forward_generic(spec, sd, x, training, upd, return_logits):
    n = spec.n_blocks
    d, mps = [], []
    for i in range(n):
        src = x if i == 0 else mps[-1]
        d.append(double_conv(src, sd, f"d_blocks.{i}.block", 0, spec.pad, training, upd))
        mps.append(pool(d[-1]))
    # models.py:238-241 -- the centre block runs and its result is dropped
    # (fc_layer is None in every shipped class); only its BN buffers observe it.
    if training and upd is not None:
        with torch.no_grad():
            double_conv(mps[-1], sd, "cblock.block", 0, spec.pad, True, upd)
    cur = mps[-1]
    for i in range(n):
        ubl = up_block(cur, sd, f"u_blocks.{i}.block", spec.pad, training, upd)
        if spec.skip:
            cur = torch.cat((ubl, d[-i - 1]), 1) if spec.cat else ubl + d[-i - 1]
        else:
            cur = ubl
    lc = F.conv3d(cur, sd["last_conv.weight"], sd["last_conv.bias"])
    if return_logits:
        return lc
    out = F.softmax(lc, 1) if spec.apply_softmax else lc
    out = torch.sigmoid(out) if spec.apply_sigmoid else out
    if spec.head == "plain":
        return out
    sk, fl = sp_head(out)
    if spec.head == "sp_softmax":           # models.py:364-365
        return F.softmax(sk, 1), F.softmax(fl, 1)
    return sk, fl


def _forward_legacy(spec, sd, x, training, upd, return_logits):
    downs, cur = [], x
    for i in range(1, 5):
        downs.append(double_conv(cur, sd, f"dblock{i}", 0, spec.pad, training, upd))
        cur = pool(downs[-1])
    cur = double_conv(cur, sd, "cblock_center", 0, spec.pad, training, upd)
    for i in range(1, 5):
        up = up_block(cur, sd, f"ublock{i}", spec.pad, training, upd)
        cur = torch.cat((up, downs[-i]), 1)
    lc = F.conv3d(cur, sd["last_conv.weight"], sd["last_conv.bias"])
    if return_logits:
        return lc
    return F.softmax(lc, 1)


def bn_checkpoint_replay(spec: NetSpec, sd: Dict[str, torch.Tensor], batch_stats: Dict[str, Tuple[torch.Tensor, torch.Tensor, int]]):
    """Second running-stat update that ``torch.utils.checkpoint`` causes.

    With the reference default ``use_checkpoint=True`` every block is re-run in
    backward, so each BN applies its momentum update twice per step with the same
    batch statistics; the dead centre block is never recomputed.
    ``batch_stats[prefix] = (mean, biased_var, n)``.
    """
    for prefix, (mean, var, n) in batch_stats.items():
        if prefix.startswith("cblock."):
            continue
        sd[prefix + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
        sd[prefix + ".running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / (n - 1))
        sd[prefix + ".num_batches_tracked"] += 1


# --------------------------------------------------------------------------
# losses and metrics
# --------------------------------------------------------------------------
def dice_loss(probs: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
    """utilities.py:39-50: per batch item over all C*D*H*W, eps 1e-7."""
    b = masks.size(0)
    p, m = probs.reshape(b, -1), masks.reshape(b, -1)
    num = (p * m).sum(1)
    den = (p * p).sum(1) + (m * m).sum(1)
    eps = 0.0000001
    return 1 - 2 * torch.mean((num + eps) / (den + eps))


def cross_entropy(pred: torch.Tensor, onehot_target: torch.Tensor) -> torch.Tensor:
    """ProblemHandler.py:67-69 / 247-256: CE(mean) on the map as logits, target argmax."""
    return F.cross_entropy(pred, argmax1(onehot_target))


def loss_single(pred, target, ce_lambda: float, dice_lambda: float):
    """ProblemHandler.comp_losses_metrics, ProblemHandler.py:59-88.
    Returns (total, {'ce':..,'dice_loss':..}) in the reference's list order."""
    terms, parts = [], {}
    if ce_lambda != 0:
        terms.append(ce_lambda * cross_entropy(pred, target))
        parts["ce"] = terms[-1]
    if dice_lambda != 0:
        terms.append(dice_lambda * dice_loss(pred, target))
        parts["dice_loss"] = terms[-1]
    return sum(terms), parts


def loss_double(pred: Sequence[torch.Tensor], target: Sequence[torch.Tensor],
                ce_lambda: float, dice_lambda: float):
    """FlapRecWithShapePriorDoubleOut.comp_losses_metrics, ProblemHandler.py:228-298."""
    sk_p, fl_p = pred
    sk_t, fl_t = target
    terms, parts = [], {}
    if ce_lambda != 0:
        terms.append(ce_lambda * cross_entropy(sk_p, sk_t)); parts["ce_sk"] = terms[-1]
        terms.append(ce_lambda * cross_entropy(fl_p, fl_t)); parts["ce_fl"] = terms[-1]
    if dice_lambda != 0:
        sk_sm, fl_sm = F.softmax(sk_p, 1), F.softmax(fl_p, 1)
        terms.append(dice_lambda * dice_loss(sk_sm, sk_t)); parts["dice_loss_sk"] = terms[-1]
        terms.append(dice_lambda * dice_loss(fl_sm, fl_t)); parts["dice_loss_fl"] = terms[-1]
    return sum(terms), parts


def argmax1(t: torch.Tensor) -> torch.Tensor:
    """torch.argmax(t, 1) (first maximum wins) as contiguous elementwise passes over the class planes: ATen-CPU's strided
    argmax over a 2-channel 256^3 map takes 6 s per call (60 of the 139 s of the 256^3 full-size test), this takes 0.1 s."""
    idx = torch.zeros_like(t[:, 0], dtype=torch.long)
    best = t[:, 0]
    for c in range(1, t.shape[1]):
        m = t[:, c] > best
        idx = torch.where(m, torch.full_like(idx, c), idx)
        best = torch.where(m, t[:, c], best)
    return idx


def hard_segmentation(prob_map: torch.Tensor) -> torch.Tensor:
    """utilities.py:118-119: argmax over the class dim, as float."""
    return argmax1(prob_map).float()


def hard_dice(pred: torch.Tensor, target_onehot: torch.Tensor) -> torch.Tensor:
    """Foreground hard Dice of argmax(pred) vs a one-hot target.

    PARITY UNPINNED: the reference calls monai.metrics.compute_meandice
    (utilities.py:53-59); monai is un-vendored, unpinned and absent here.  This
    restates its published definition 2|A.B|/(|A|+|B|) on channels >= 1, mean over
    batch/classes; empty-vs-empty is defined as 1.0 (monai yields NaN there).
    """
    c = pred.shape[1]
    hard = F.one_hot(argmax1(pred), c).movedim(-1, 1).to(target_onehot.dtype)
    vals = []
    for ch in range(1, c):
        a, b = hard[:, ch].flatten(1), target_onehot[:, ch].flatten(1)
        inter, tot = (a * b).sum(1), a.sum(1) + b.sum(1)
        vals.append(torch.where(tot > 0, 2 * inter / tot.clamp_min(1), torch.ones_like(tot)))
    return torch.stack(vals, 1).mean()


# --------------------------------------------------------------------------
# helpers for tests / baselines
# --------------------------------------------------------------------------
def grads(spec: NetSpec, sd: Dict[str, torch.Tensor], x: torch.Tensor, loss_fn,
          training: bool = True):
    """Forward + backward through the oracle.  Returns (out, loss, {name: grad}, dx)."""
    leaf = {}
    for k_, v in sd.items():
        if v.is_floating_point() and not ("running_" in k_):
            leaf[k_] = v.detach().clone().requires_grad_(True)
        else:
            leaf[k_] = v
    xi = x.detach().clone().requires_grad_(True)
    out = forward(spec, leaf, xi, training=training)
    loss = loss_fn(out)
    loss.backward()
    g = {k_: v.grad for k_, v in leaf.items() if isinstance(v, torch.Tensor) and v.requires_grad}
    return out, loss.detach(), g, xi.grad
