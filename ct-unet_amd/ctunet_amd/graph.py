"""Whole-step HIP graph: forward + loss + backward + optimizer of one fixed-shape batch, captured once
and replayed, so the ~270 kernel launches of a step cost one graph launch (MI355X guide: "capture
launch-bound inner loops in hipGraphs").

The step is the reference's ``Model.forward_pass`` train branch (ctunet/pytorch/Model.py:343-374) minus its
host round trips: the per-term ``float(loss)`` syncs become ONE device->host copy after the replay.
With ``process_group`` set (one process per GPU) the captured graph holds forward + loss + backward only;
the gradient mean over ranks (one flat RCCL all-reduce, 3.3 MB for UNet()) and the optimizer step are
launched eagerly after each replay -- no collective is ever captured.  The bucketed, backward-overlapped
``parallel.GradSync`` path remains the eager alternative.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from .losses import fused_ce_dice


class GraphedTrainStep:
    """model: a ctunet_amd model on the GPU; optimizer: torch.optim.Adam/AdamW built with capturable=True."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, example_input: torch.Tensor,
                 example_targets: Sequence[torch.Tensor], ce_lambda: float, dice_lambda: float,
                 input_requires_grad: bool = True, warmup: int = 3, distributed: bool = False, process_group=None):
        self.model, self.opt = model, optimizer
        self.distributed, self.group = distributed, process_group
        if distributed and model.__dict__.get("_grad_sync_cfg") is not None:
            raise RuntimeError("GraphedTrainStep(distributed=True) does its own all-reduce: do not also call "
                               "parallel.distribute() on the model (use parallel.broadcast_parameters)")
        self.ce, self.dice = float(ce_lambda), float(dice_lambda)
        self.x = example_input.detach().clone()
        self.targets = [t.detach().clone() for t in example_targets]
        self.x_req = input_requires_grad
        self.double = len(self.targets) == 2
        self.values: Optional[torch.Tensor] = None
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph):
            self.values = self._step()
        self.keys = self._keys()
        if distributed:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            self.live = [p for p in model.parameters() if p.grad is not None]     # static .grad tensors of the graph
            self.sizes = [p.grad.numel() for p in self.live]

    def _keys(self) -> List[str]:
        k: List[str] = []
        if self.double:
            if self.ce:
                k += ["ce_sk", "ce_fl"]
            if self.dice:
                k += ["dice_loss_sk", "dice_loss_fl"]
        else:
            if self.ce:
                k += ["ce"]
            if self.dice:
                k += ["dice_loss"]
        return k + ["epoch_loss"]

    def _step(self) -> torch.Tensor:
        xi = self.x.requires_grad_(self.x_req)
        xi.grad = None
        out = self.model(xi)
        terms: List[torch.Tensor] = []
        if self.double:
            ce_s, dc_s = fused_ce_dice(out[0], self.targets[0], self.ce, self.dice, True)
            ce_f, dc_f = fused_ce_dice(out[1], self.targets[1], self.ce, self.dice, True)
            if self.ce:
                terms += [ce_s, ce_f]
            if self.dice:
                terms += [dc_s, dc_f]
        else:
            ce, dc = fused_ce_dice(out, self.targets[0], self.ce, self.dice, False)
            if self.ce:
                terms.append(ce)
            if self.dice:
                terms.append(dc)
        loss = sum(terms)
        loss.backward()
        if not self.distributed:
            self.opt.step()
            for p in self.model.parameters():
                p.grad = None
        return torch.stack([t.detach() for t in terms] + [loss.detach()])

    def _reduce_and_step(self) -> None:
        import torch.distributed as dist
        grads = [p.grad for p in self.live]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split(self.sizes), grads)])
        self.opt.step()

    def __call__(self, x: Optional[torch.Tensor] = None, targets: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        """Copies the batch into the captured buffers, replays the step, returns the loss terms (device tensor,
        order ``self.keys``); call ``.tolist()`` on it for the reference's logged floats (one sync)."""
        if x is not None:
            self.x.detach().copy_(x)
        if targets is not None:
            for dst, src in zip(self.targets, targets):
                dst.copy_(src)
        self.graph.replay()
        if self.distributed:
            self._reduce_and_step()
        return self.values
