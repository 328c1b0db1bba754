"""Dice / cross-entropy loss of the ctunet path on the fused gfx950 loss kernels.

API mirrors of
  ``ctunet.utilities.dice_loss``                                   utilities.py:35-50
  ``ProblemHandler.comp_losses_metrics``                            ProblemHandler.py:44-102
  ``FlapRecWithShapePriorDoubleOut.comp_losses_metrics``            ProblemHandler.py:213-309
(the reference's "binary cross entropy" is nn.CrossEntropyLoss on the 2-channel map taken as
logits with target argmax(one_hot, 1), ProblemHandler.py:67-69,247-256).

One kernel pass computes CE and Dice of a 2-channel map, one more its gradient; the reference
makes ~10 elementwise/reduction passes and 5 host syncs per batch for the same numbers.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops


class _FusedLoss(torch.autograd.Function):
    """(ce_lambda * CE, dice_lambda * Dice) of one [N,2,D,H,W] map as two 0-d tensors."""

    @staticmethod
    def forward(ctx, pred, target, ce_lambda, dice_lambda, dice_softmax):
        pred_c, target_c = pred.contiguous(), target.contiguous()
        terms, ws = ops.loss_fwd(pred_c, target_c, float(ce_lambda), float(dice_lambda), bool(dice_softmax))
        ctx.save_for_backward(pred_c, target_c, ws)
        ctx.cfg = (float(ce_lambda), float(dice_lambda), bool(dice_softmax))
        return terms[0], terms[1]

    @staticmethod
    def backward(ctx, g_ce, g_dice):
        pred, target, ws = ctx.saved_tensors
        ce, dc, sm = ctx.cfg
        # a term nobody differentiated contributes nothing (lambda 0); the other upstream scalars are read in place
        gp = ops.loss_bwd(pred, target, ce if g_ce is not None else 0.0, dc if g_dice is not None else 0.0, sm, ws,
                          None if g_ce is None else g_ce.float(), None if g_dice is None else g_dice.float())
        return gp, None, None, None, None


def _check_map(pred: torch.Tensor, target: torch.Tensor) -> None:
    if pred.dim() != 5 or pred.shape[1] != 2 or pred.shape != target.shape:
        raise RuntimeError("ctunet_amd: the fused loss handles [N,2,D,H,W] prediction / one-hot target pairs "
                           f"(got {tuple(pred.shape)} / {tuple(target.shape)})")


def fused_ce_dice(pred, target, ce_lambda, dice_lambda, dice_softmax) -> Tuple[torch.Tensor, torch.Tensor]:
    _check_map(pred, target)
    return _FusedLoss.apply(pred, target, ce_lambda, dice_lambda, dice_softmax)


class dice_loss(nn.Module):
    """Soft Dice loss, same call signature as ``ctunet.utilities.dice_loss`` (2-channel maps)."""

    def forward(self, output, masks):
        return fused_ce_dice(output, masks, 0.0, 1.0, False)[1]


def _total(terms: Sequence[torch.Tensor], like: torch.Tensor = None) -> torch.Tensor:
    """terms[0] + terms[1] + ... (the builtin sum() starts from int 0: one more elementwise launch per step).
    No term at all (both lambdas 0): the reference's ``sum([])`` is the int 0 (ProblemHandler.py:91) -- a zero scalar
    here, so that ``float(model.pt_loss)`` and the epoch_loss log behave the same."""
    if not terms:
        return torch.zeros((), dtype=torch.float32, device=None if like is None else like.device)
    total = terms[0]
    for t in terms[1:]:
        total = total + t
    return total


def _append(lm: Dict[str, List], key: str, value) -> None:
    lm.setdefault(key, []).append(value)


def _publish(model, keys: Sequence[str], tensors: Sequence[torch.Tensor], idx, n_imgs, verbose=True) -> None:
    """One device->host copy for all logged scalars (the reference syncs once per float())."""
    total = model.pt_loss
    vals = torch.stack([t.detach() for t in tensors] + [total.detach()]).tolist()
    lm = model.losses_and_metrics
    for k_, v in zip(keys, vals[:-1]):
        _append(lm, k_, v)
    _append(lm, "epoch_loss", vals[-1])
    if verbose:
        print("    Batch {}/{} ({:.0f}%)\tLoss: {:.6f}".format(idx + 1, n_imgs, 100.0 * (idx + 1) / n_imgs, vals[-1]))


def comp_losses_metrics_single(model, prediction, target, idx, n_imgs):
    """``ProblemHandler.comp_losses_metrics`` (ProblemHandler.py:44-102): CE + Dice on the raw map."""
    ce_l, dc_l = model.params["ce_lambda"], model.params["dice_lambda"]
    if target.dim() != 5:
        # integer class labels [N,D,H,W] are used as they are by the cross entropy (ProblemHandler.py:67-68); the Dice
        # term of the reference cannot take them (dice_loss multiplies [N,2V] by [N,V], utilities.py:45-47: RuntimeError)
        if dc_l != 0:
            raise RuntimeError("ctunet_amd: dice_loss needs a one-hot target of the prediction's shape "
                               f"(got {tuple(target.shape)} for {tuple(prediction.shape)}), as in the reference")
        if target.dim() != 4 or target.shape != prediction.shape[:1] + prediction.shape[2:]:
            raise RuntimeError(f"ctunet_amd: class-index target {tuple(target.shape)} does not match {tuple(prediction.shape)}")
        from . import ops
        target = ops.one_hot(target.float().contiguous(), prediction.shape[1])
    ce, dc = fused_ce_dice(prediction, target, ce_l or 0.0, dc_l or 0.0, False)
    keys, terms = [], []
    if ce_l != 0:
        keys.append("ce"); terms.append(ce)
    if dc_l != 0:
        keys.append("dice_loss"); terms.append(dc)
    model.pt_loss = _total(terms, prediction)
    _metrics(model, [("dice_coef", prediction, target)])
    _publish(model, keys, terms, idx, n_imgs, getattr(model, "verbose", True))


def comp_losses_metrics_double(model, prediction, target, idx, n_imgs):
    """``FlapRecWithShapePriorDoubleOut.comp_losses_metrics`` (ProblemHandler.py:213-309): CE on the
    raw maps, Dice on their softmax, list order ce_sk, ce_fl, dice_loss_sk, dice_loss_fl."""
    sk_p, fl_p = prediction
    sk_t, fl_t = target
    ce_l, dc_l = model.params["ce_lambda"], model.params["dice_lambda"]
    ce_s, dc_s = fused_ce_dice(sk_p, sk_t, ce_l or 0.0, dc_l or 0.0, True)
    ce_f, dc_f = fused_ce_dice(fl_p, fl_t, ce_l or 0.0, dc_l or 0.0, True)
    keys, terms = [], []
    if ce_l != 0:
        keys += ["ce_sk", "ce_fl"]; terms += [ce_s, ce_f]
    if dc_l != 0:
        keys += ["dice_loss_sk", "dice_loss_fl"]; terms += [dc_s, dc_f]
    model.pt_loss = _total(terms, sk_p)
    _metrics(model, [("dice_coef_sk", sk_p, sk_t), ("dice_coef_fl", fl_p, fl_t)], hd=True)
    _publish(model, keys, terms, idx, n_imgs, getattr(model, "verbose", True))


def _metrics(model, items, hd: bool = False) -> None:
    if model.params.get("save_dice_plots") is True:
        from .utilities import dice_coeff
        for key, p, t in items:
            _append(model.losses_and_metrics, key, dice_coeff(p, t))
    # Hausdorff: only the double-output handler logs it (ProblemHandler.py:287-295: keys hd_coef_sk / hd_coef_fl)
    if hd and model.params.get("save_hd_plots") is True:
        from .utilities import hausdorff
        for key, p, t in items:
            hk = key.replace("dice_coef", "hd_coef")
            _append(model.losses_and_metrics, hk, hausdorff(p, t))
