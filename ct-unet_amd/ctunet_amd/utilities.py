"""Host-side mirrors of the ``ctunet.utilities`` helpers the hot path touches.

  set_cfg_params / load_params ... utilities.py:215-256, ctunet/__init__.py:1-2 (ini -> typed dict)
  hard_segm_from_tensor .......... utilities.py:103-124
  dice_coeff ..................... utilities.py:53-59 (monai.compute_meandice; PARITY UNPINNED, see below)
"""
from __future__ import annotations

import configparser
import os

import torch

from .losses import dice_loss  # noqa: F401  (same import location as the reference: utils.dice_loss)


def set_cfg_params(cfg_file=None, default_dict=None):
    """INI -> dict; key prefixes ``i_ f_ b_ s_`` select int/float/bool/str and are stripped,
    sections are flattened, un-prefixed keys stay strings (utilities.py:215-256)."""
    if cfg_file is None:
        return None
    out = default_dict if default_dict is not None else dict()
    if not os.path.exists(cfg_file):
        raise FileNotFoundError(f"The provided cfg file does not exist ({cfg_file}).")
    cfg = configparser.ConfigParser()
    cfg.read(cfg_file)
    conv = {"i_": lambda sec, k: sec.getint(k), "f_": lambda sec, k: sec.getfloat(k),
            "b_": lambda sec, k: sec.getboolean(k), "s_": lambda sec, k: sec[k]}
    for name in cfg.sections():
        sec = cfg[name]
        for key in sec:
            f = conv.get(key[:2])
            if f is None:
                out[key] = sec[key]
            else:
                out[key[2:]] = f(sec, key)
    return out


def load_params(cfg_file, default_dict=None):
    """``ctunet.load_params`` (ctunet/__init__.py): alias of set_cfg_params."""
    return set_cfg_params(cfg_file, {} if default_dict is None else default_dict)


def hard_segm_from_tensor(prob_map, keep_dims=False):
    """argmax over the class dimension as float (utilities.py:103-124), one pass on the GPU."""
    from . import ops
    dim = 1 if prob_map.dim() == 5 else 0
    seg = ops.hard_segm(prob_map.detach().float().contiguous())
    return seg.unsqueeze(dim) if keep_dims else seg


def dice_coeff(pred, target):
    """Foreground hard Dice of argmax(pred) against a one-hot target, mean over batch and classes.

    The reference calls ``monai.metrics.compute_meandice(one_hot(argmax(pred,1)), target,
    include_background=False)`` (utilities.py:53-59).  monai is an unpinned third-party package that
    is not available here, so this follows its published definition 2|A.B| / (|A| + |B|);
    empty-vs-empty is defined as 1.0 (monai returns NaN).  PARITY UNPINNED.
    The three counts come from one pass over the maps (``ctu_hard_dice_counts``, exact integers).
    """
    from . import ops
    cnt = ops.hard_dice_counts(pred.detach().float().contiguous(), target.detach().float().contiguous())[:, 1:]
    inter, tot = cnt[..., 0], cnt[..., 1] + cnt[..., 2]
    return torch.where(tot > 0, 2 * inter / tot.clamp_min(1), torch.ones_like(tot)).mean().float()


def hausdorff(result_b, reference_b):
    """Hausdorff distance of argmax(result_b) against a one-hot reference, mean over batch and foreground classes.

    The reference calls ``monai.metrics.compute_hausdorff_distance(one_hot(argmax(result_b, 1)), reference_b)`` and maps
    nan / inf to ``max(reference_b.shape)`` (utilities.py:62-70).  monai is unpinned, un-vendored and absent here, so
    the surface definition (mask & ~erode(mask), 6-neighbourhood), the Euclidean metric and the symmetric maximum follow
    its published algorithm; PARITY UNPINNED (tests compare with the same definition on scipy.ndimage).
    One exact integer distance transform per (item, class, side) on the GPU (``ctu_hausdorff``).
    """
    from . import ops
    inf_alt = float(max(reference_b.shape))
    hd = ops.hausdorff(result_b.detach().float().contiguous(), reference_b.detach().float().contiguous())
    return torch.nan_to_num(hd, nan=inf_alt, posinf=inf_alt, neginf=inf_alt).mean()
