"""The caller side of the hot path: what ``ctunet.pytorch.Model`` does between reading an ini and the
train step -- name resolution of model / handler classes, optimizer construction, one pass over a data loader --
for this package's classes, so the parameter dicts of the reference's example inis drive the MI355X path unchanged.

Mirrors (not copies): ``Model.new_model`` (Model.py:474-491: ``eval(params["model_class"])()``), ``Model.__init__``'s
handler resolution (Model.py:101), ``initialize_optimizer`` (Model.py:510-546: Adam/AdamW with amsgrad, RMSprop, SGD,
ReduceLROnPlateau) and ``forward_pass`` (Model.py:324-374).  Workspace folders, TensorBoard, checkpoint rotation,
NIfTI writers and the CLI stay the reference's (out of scope).  The host object carries the attributes
``comp_losses_metrics`` writes to (``pt_loss``, ``losses_and_metrics``, ``params``), as ``Model`` does.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ProblemHandler as _handlers
from . import models as _models
from . import optim as _optim


def resolve_model(name: str) -> torch.nn.Module:
    """``eval(self.params["model_class"])()`` over this package's model namespace (zero-argument constructors)."""
    cls = getattr(_models, name, None)
    if not (isinstance(cls, type) and issubclass(cls, torch.nn.Module)) or name.startswith("_"):
        raise NameError(f"name '{name}' is not defined")          # what eval() raises in the reference
    return cls()


def resolve_handler(name: str):
    cls = getattr(_handlers, name, None)
    if not isinstance(cls, type):
        raise NameError(f"name '{name}' is not defined")
    return cls()


class StepRunner:
    """Holds one model + handler + optimizer and runs ``forward_pass`` phases over loaders of the reference's sample
    schema ({"image", "target", "filepath"}; ``ctunet_amd.datasets``)."""

    verbose = False

    def __init__(self, params: Dict):
        self.params = dict(params)
        self.params.setdefault("save_dice_plots", False)
        self.params.setdefault("save_hd_plots", False)
        self.losses_and_metrics: Dict[str, list] = {}
        self.pt_loss: Optional[torch.Tensor] = None
        self.problem_handler = resolve_handler(self.params["problem_handler"])
        self.comp_losses_metrics = type(self.problem_handler).comp_losses_metrics
        self.models = {"main": resolve_model(self.params["model_class"]).to(self.params.get("device", "cuda"))}
        self.initialize_optimizer()

    def initialize_optimizer(self) -> None:
        p, net = self.params, self.models["main"]
        name = p.get("optimizer", "adam")
        lr, wd = p["learning_rate"], p.get("weight_decay", 0.0) or 0.0
        if name == "adam":                      # fused multi-tensor kernels (same update rule as optim.Adam(amsgrad=True))
            p["optimizer"] = _optim.Adam(net.parameters(), lr=lr, weight_decay=wd, amsgrad=True)
        elif name == "adamw":
            p["optimizer"] = _optim.AdamW(net.parameters(), lr=lr, weight_decay=wd, amsgrad=True)
        elif name == "rmsprop":
            p["optimizer"] = torch.optim.RMSprop(net.parameters(), lr=lr, weight_decay=wd, momentum=p.get("momentum", 0) or 0)
        elif name == "sgd":
            p["optimizer"] = torch.optim.SGD(net.parameters(), lr=lr, momentum=p.get("momentum", 0) or 0, weight_decay=wd)
        elif not isinstance(name, torch.optim.Optimizer):
            raise ValueError(f"ctunet_amd: unknown optimizer '{name}'")
        if hasattr(p["optimizer"], "guard"):
            p["optimizer"].guard(net)          # float16 activations: steps whose backward overflowed are skipped
        if "scheduler" in p:      # built whenever the key exists, whatever its value (Model.py:544-546)
            p["scheduler"] = torch.optim.lr_scheduler.ReduceLROnPlateau(p["optimizer"])

    def forward_pass(self, phase: str, data_loader) -> None:
        """One pass over ``data_loader``: 'train' updates the parameters, 'validation'/'val' only evaluates the
        losses; 'test' (prediction writing) is outside the accelerated path."""
        if phase == "test":
            raise NotImplementedError("ctunet_amd: the test phase writes NIfTI predictions (reference side)")
        if phase not in ("train", "validation", "val"):
            raise ValueError(f"unknown phase '{phase}'")
        net, dev = self.models["main"], self.params.get("device", "cuda")
        train = phase == "train"
        net.train(train)
        with torch.set_grad_enabled(train):
            for batch_idx, sample in enumerate(data_loader):
                input_img = sample["image"].to(dev)
                target = sample["target"]
                target = [e.to(dev) for e in target] if isinstance(target, (list, tuple)) else target.to(dev)
                if train:
                    input_img.requires_grad_()
                model_out = net(input_img)
                self.comp_losses_metrics(self, model_out, target, batch_idx, len(data_loader))
                if train:
                    self.pt_loss.backward()
                    self.params["optimizer"].step()
                    if self.params.get("scheduler") is not None:
                        self.params["scheduler"].step(self.pt_loss)
                    for param in net.parameters():
                        param.grad = None

    def epoch_averages(self, reset: bool = True) -> Dict[str, float]:
        """Mean of every logged key, as ``update_plots_tensorboard_avg`` computes it (Model.py:382-405)."""
        out = {k: sum(v) / len(v) for k, v in self.losses_and_metrics.items() if v}
        if reset:
            for k in self.losses_and_metrics:
                self.losses_and_metrics[k] = []
        return out
