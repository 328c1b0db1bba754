"""Patch-level data parallelism: one process per GPU, gradient all-reduce over RCCL/xGMI.

The reference's only multi-GPU mechanism is single-process ``nn.DataParallel``
(/root/reference/ctunet/pytorch/Model.py:481-487), which cannot even split the batch of 1 that
every example ini uses.  This replaces it (it does not mirror it): each rank runs the full net on
its own patch; the only exchange is the mean of the parameter gradients before the optimizer step.

Design for MI355X (8 GPUs, point-to-point xGMI, 7 links x ~153 GB/s): the payload is tiny
(3.3 MB for UNet(), 27 MB for recAE_v2_fixed) so the collective is latency-bound; gradients are
flattened into a few large buckets in the order backward produces them (head -> decoder ->
encoder) and each bucket's ``all_reduce`` is issued on a side stream as soon as its last weight
gradient kernel has been enqueued, overlapping the remaining backward convolutions.
Parameters the graph never touches (the dead centre block of the generic UNet, models.py:241) never
enter a bucket on any rank; their ``.grad`` stays ``None`` as in the reference.
BatchNorm statistics stay per-rank (no SyncBN), matching batch-1-per-GPU reference semantics.

backend "nccl" IS RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class GradSync:
    """Bucketed, overlapped mean-all-reduce of gradients produced block by block."""

    def __init__(self, process_group=None, bucket_bytes: int = 8 << 20):
        self.group = process_group
        self.bucket_bytes = int(bucket_bytes)
        self._pending: List[Tuple[str, torch.Tensor]] = []
        self._pending_bytes = 0
        self._inflight: List[Tuple[torch.Tensor, List[Tuple[str, torch.Tensor, int]], object]] = []
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1

    # -- called by the engine as soon as a block's gradients have been enqueued
    def push(self, named_grads: Sequence[Tuple[str, torch.Tensor]]) -> None:
        if self.world == 1:
            return
        for name, g in named_grads:
            self._pending.append((name, g))
            self._pending_bytes += g.numel() * g.element_size()
        if self._pending_bytes >= self.bucket_bytes:
            self._launch()

    def _launch(self) -> None:
        if not self._pending:
            return
        items = self._pending
        self._pending, self._pending_bytes = [], 0
        total = sum(g.numel() for _, g in items)
        ref = items[0][1]
        flat = torch.empty(total, dtype=ref.dtype, device=ref.device)
        layout, off = [], 0
        for name, g in items:
            n = g.numel()
            flat[off:off + n].copy_(g.reshape(-1))
            layout.append((name, g, off))
            off += n
        if flat.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            flat.record_stream(self._comm_stream)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((flat, layout, work))

    def finish(self) -> Dict[str, torch.Tensor]:
        """Flush, wait for every bucket, and return {name: averaged gradient} as views of the buckets."""
        if self.world == 1:
            return {}
        self._launch()
        out: Dict[str, torch.Tensor] = {}
        inv = 1.0 / self.world
        for flat, layout, work in self._inflight:
            work.wait()                      # on CUDA: makes the current stream wait for the collective
            flat.mul_(inv)
            for name, g, off in layout:
                out[name] = flat[off:off + g.numel()].view_as(g)
        self._inflight = []
        return out


def distribute(module: torch.nn.Module, process_group=None, bucket_bytes: int = 8 << 20,
               broadcast: bool = True) -> torch.nn.Module:
    """Make ``module`` (a ctunet_amd model) data-parallel across the ranks of ``process_group``.

    Broadcasts parameters and buffers from rank 0 once, then averages gradients inside every
    backward.  The module is returned unwrapped: ``state_dict`` keys keep the reference's names
    (no ``module.`` prefix).
    """
    if not dist.is_initialized():
        raise RuntimeError("ctunet_amd.parallel.distribute: torch.distributed is not initialised")
    if broadcast:
        broadcast_parameters(module, process_group)
    module.__dict__["_grad_sync_cfg"] = (process_group, bucket_bytes)
    return module


def broadcast_parameters(module: torch.nn.Module, process_group=None) -> torch.nn.Module:
    """Make every rank start from rank 0's parameters and buffers (no gradient hook is installed)."""
    ts = list(module.parameters()) + list(module.buffers())
    for t in ts:
        dist.broadcast(t.data, src=0, group=process_group)
    # .data writes are invisible to the version counters the engine's packed-weight caches are keyed on
    torch.autograd.graph.increment_version(ts)
    return module


def make_sync(module) -> Optional[GradSync]:
    cfg = module.__dict__.get("_grad_sync_cfg")
    if cfg is None:
        return None
    return GradSync(cfg[0], cfg[1])


def allreduce_mean_(tensors: Sequence[Optional[torch.Tensor]], process_group=None,
                    bucket_bytes: int = 8 << 20) -> None:
    """In-place mean over ranks of a list of tensors (``None`` entries are skipped on every rank).
    Stand-alone form of the bucket logic, for optimizers/gradients produced outside the engine."""
    sync = GradSync(process_group, bucket_bytes)
    live = [(str(i), t) for i, t in enumerate(tensors) if t is not None]
    sync.push(live)
    red = sync.finish()
    for name, t in live:
        if name in red:
            t.copy_(red[name])
