"""Whole-step HIP graph: forward + loss + backward + optimizer of one fixed-shape batch, captured once
and replayed, so the ~270 kernel launches of a step cost one graph launch (MI355X guide: "capture
launch-bound inner loops in hipGraphs").

The step is the reference's ``Model.forward_pass`` train branch (ctunet/pytorch/Model.py:343-374) minus its
host round trips: the per-term ``float(loss)`` syncs become ONE device->host copy after the replay.
With ``distributed`` set (one process per GPU) the step is TWO graphs around one eagerly launched collective:
graph 1 = forward + loss + backward + flattening of the live gradients into one static buffer; then the gradient
sum over ranks (one flat RCCL all-reduce, 3.3 MB for UNet()) -- no collective is ever captured --; graph 2 = the fused
optimizer step reading the gradients straight from the flat buffer.  The bucketed,
backward-overlapped ``parallel.GradSync`` path remains the eager alternative.
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
        self._params = [p for p in model.parameters()]
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
        if distributed:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
                if distributed:
                    self._eager_reduce_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # capture_error_mode: with a process group alive, RCCL's watchdog THREAD polls its work events at any time; under
        # the default "global" mode such a query from another thread invalidates the capture (and the watchdog aborts the
        # process) -- a race that depends on when the last warm-up all-reduce retires.  Only this thread's calls matter.
        mode = "thread_local" if distributed else "global"
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.values = self._step()
        self.keys = self._keys()
        if distributed:
            if warmup == 0:                                   # the optimizer state must exist before graph 2 is captured
                self.graph.replay()
                self._eager_reduce_step()
                torch.cuda.synchronize()
            # graph 2: the optimizer reads its gradients from views of the static flat buffer graph 1 fills
            for p, v in zip(self._live, self.flat.split([p.numel() for p in self._live])):
                p.grad = v.view_as(p)
            self.graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph2, capture_error_mode=mode):
                self._scale_and_step()

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
        loss = terms[0]
        for t in terms[1:]:
            loss = loss + t
        loss.backward()
        if not self.distributed:
            self.opt.step()
            for p in self.model.parameters():
                p.grad = None
        else:
            # every live gradient into one flat buffer (static once captured); grads are left to the next backward,
            # which overwrites them (parameters the graph never touches keep .grad None on every rank)
            self._live = [p for p in self.model.parameters() if p.grad is not None]
            self.flat = torch.cat([p.grad.reshape(-1) for p in self._live])
            for p in self._live:
                p.grad = None
        return torch.stack([t.detach() for t in terms] + [loss.detach()])

    def _scale_and_step(self) -> None:
        self.opt.step()                 # the collective already averaged (ncclAvg)

    def _allreduce(self) -> None:
        """Mean over ranks of the flat gradient buffer: RCCL through the C ABI on the current stream (never captured)."""
        from .parallel import get_communicator
        get_communicator(self.group).allreduce_(self.flat, average=True)

    def _eager_reduce_step(self) -> None:
        """One eagerly launched all-reduce + optimizer step on the flat buffer the last _step() produced."""
        for p, v in zip(self._live, self.flat.split([p.numel() for p in self._live])):
            p.grad = v.view_as(p)
        self._allreduce()
        self._scale_and_step()
        for p in self._live:
            p.grad = None

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
            self._allreduce()
            self.graph2.replay()
        # the replayed optimizer kernel wrote the parameters through raw pointers: bump their version counters, so that
        # an eagerly launched forward after this replay (the reference's train-then-validate epoch loop,
        # Model.py:240-243) re-packs the engine's MFMA-ordered weight copies instead of hitting a stale cache entry
        torch.autograd.graph.increment_version(self._params)
        return self.values
