"""Checkpoint interchange with the reference trainer.

``Model.save_main_model`` writes ``torch.save(model.state_dict(), path)`` (Model.py:266-296) and
``Model.load_model`` accepts a pickled module or an OrderedDict (Model.py:448-472); under
``nn.DataParallel`` the keys carry a ``module.`` prefix (Model.py:282,486).  The drop-in classes keep the
reference's key names, so a plain ``load_state_dict`` works; this helper additionally strips the prefix
and only ever uses loaders that execute nothing from the file (``weights_only=True``).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Mapping

import torch


def strip_module_prefix(sd: Mapping[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in sd.items())


def load_state(model: torch.nn.Module, source, strict: bool = True) -> torch.nn.Module:
    """source: a state-dict mapping or a path to a ``.pt`` written by ``torch.save(state_dict)``."""
    if isinstance(source, (str, bytes)):
        source = torch.load(source, map_location="cpu", weights_only=True)
    model.load_state_dict(strip_module_prefix(source), strict=strict)
    return model


def save_state(model: torch.nn.Module, path: str) -> None:
    """Same format as the reference: a bare state_dict the reference's ``load_model`` can read."""
    torch.save(OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items()), path)
