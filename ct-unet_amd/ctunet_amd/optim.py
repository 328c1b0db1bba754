"""Fused Adam / AdamW with amsgrad on the MI355X (one kernel launch for all parameter tensors).

Drop-in for ``optim.Adam(params, lr, weight_decay, amsgrad=True)`` / ``optim.AdamW(...)`` as the reference
builds them (ctunet/pytorch/Model.py:514-527): same hyper-parameter names and defaults, same state entries
(``exp_avg``, ``exp_avg_sq``, ``max_exp_avg_sq``, ``step``), same update rule; parameters whose ``.grad`` is
``None`` (the dead centre block) are skipped and keep no state, as in torch.  torch's own capturable
amsgrad path issues two elementwise kernels per parameter tensor (116 launches per step for UNet()); this
issues two launches in total and is safe to capture in a HIP graph (the step counter lives on the device).
Deviation from torch: ONE step counter per parameter group (``group["step_t"]``, also what ``state[p]["step"]``
refers to), where torch keeps one per tensor -- a parameter that receives its first gradient later than the others
of its group therefore shares their bias correction (no shipped model has such a parameter: the dead centre block
never gets a gradient at all).
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True,
                 decoupled_weight_decay=False):
        if not amsgrad:
            raise NotImplementedError("ctunet_amd.optim.Adam implements the amsgrad variant the reference uses")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=True,
                                      decoupled_weight_decay=decoupled_weight_decay))
        # fp16 training: a float32[1] device flag (model.overflow_flag()) that the backward sets when a loss-scaled gradient
        # overflowed; a step that sees it set changes nothing (torch.amp.GradScaler.step's found_inf), also inside a replayed
        # HIP graph.  None: every step is applied.
        self.skip_flag = None

    def guard(self, model) -> "Adam":
        """Skip every step whose float16 backward overflowed (``model.overflow_flag()``); no-op for fp32 / bf16 models."""
        fp16 = model.__dict__.get("_act_dtype", torch.float32) == torch.float16
        self.skip_flag = model.overflow_flag() if fp16 and hasattr(model, "overflow_flag") else None
        return self

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            live: List[torch.Tensor] = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            if "step_t" not in group:
                group["step_t"] = torch.zeros(1, dtype=torch.float32, device=live[0].device)
            elif group["step_t"].device != live[0].device:
                # load_state_dict casts per-parameter state to the parameter's device but leaves param_groups values
                # where the checkpoint had them (map_location="cpu"): the kernel must never see a host pointer
                group["step_t"] = group["step_t"].to(live[0].device)
            ptrs, sizes = [], []
            for p in live:
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("ctunet_amd.optim.Adam: parameters must be float32 on the GPU (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = group["step_t"]                    # shared device counter (torch keeps one per tensor)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ptrs += [p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                         st["max_exp_avg_sq"].data_ptr()]
                sizes.append(p.numel())
            n = len(live)
            pa = (C.c_void_p * (5 * n))(*ptrs)
            sa = (C.c_int64 * n)(*sizes)
            b1, b2 = group["betas"]
            _lib.check(lib.ctu_adam_amsgrad(pa, sa, n, group["step_t"].data_ptr(), float(group["lr"]), float(b1),
                                            float(b2), float(group["eps"]), float(group["weight_decay"]),
                                            int(bool(group["decoupled_weight_decay"])),
                                            None if self.skip_flag is None else self.skip_flag.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "adam_amsgrad")
            # the kernel wrote the parameters through raw pointers: tell autograd / every (version-keyed) cache of
            # derived data -- the engine's MFMA-ordered weight copies -- that they changed
            torch.autograd.graph.increment_version(live)
        return loss


class AdamW(Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=True):
        super().__init__(params, lr, betas, eps, weight_decay, amsgrad, decoupled_weight_decay=True)
