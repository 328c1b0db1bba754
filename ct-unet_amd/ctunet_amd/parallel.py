"""Patch-level data parallelism: one process per GPU, gradient all-reduce over RCCL/xGMI.

The reference's only multi-GPU mechanism is single-process ``nn.DataParallel``
(/root/reference/ctunet/pytorch/Model.py:481-487), which cannot even split the batch of 1 that
every example ini uses.  This replaces it (it does not mirror it): each rank runs the full net on
its own patch; the only exchange is the mean of the parameter gradients before the optimizer step.

Design for MI355X (8 GPUs, point-to-point xGMI, 7 links x ~153 GB/s): the payload is tiny
(3.3 MB for UNet(), 27 MB for recAE_v2_fixed) so the collective is latency-bound; gradients are
flattened into a few buckets in the order backward produces them (head -> decoder ->
encoder) and each bucket's all-reduce (RCCL through the C ABI, ``ctu_comm_allreduce_f32``) is issued on a
side stream as soon as its last weight gradient kernel has been enqueued, overlapping the remaining
backward convolutions.
Parameters the graph never touches (the dead centre block of the generic UNet, models.py:241) never
enter a bucket on any rank; their ``.grad`` stays ``None`` as in the reference.
BatchNorm statistics stay per-rank (no SyncBN), matching batch-1-per-GPU reference semantics.

backend "nccl" IS RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


DEFAULT_BUCKET_BYTES = 512 << 10


class Communicator:
    """One RCCL communicator per rank behind the C ABI (``ctu_comm_*``, include/ctunet_hip.h): ``ncclCommInitRank`` on
    this process's current GPU, ``ncclAllReduce(avg)`` on a caller-chosen stream.  ``torch.distributed`` is only the side
    channel that hands rank 0's unique id to the other ranks."""

    def __init__(self, process_group, ident: bytes):
        """ident: rank 0's 128-byte RCCL unique id, already agreed on and broadcast by get_communicator (no collective of
        torch.distributed happens here: the only rendezvous is ncclCommInitRank's own)."""
        import ctypes as C
        from . import _lib
        self._lib = _lib
        lib = _lib.load()
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        handle = C.c_void_p()
        _lib.check(lib.ctu_comm_init(C.byref(handle), self.world, self.rank, (C.c_char * 128).from_buffer_copy(ident)), "comm_init")
        self._handle = handle

    def allreduce_(self, t: torch.Tensor, average: bool = True, stream=None) -> None:
        """In place over ranks; enqueued on ``stream`` (default: torch's current stream), no synchronisation."""
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("ctunet_amd.parallel.Communicator: contiguous float32 GPU tensors only")
        st = (stream or torch.cuda.current_stream(t.device)).cuda_stream
        self._lib.check(self._lib.load().ctu_comm_allreduce_f32(self._handle, t.data_ptr(), t.data_ptr(), t.numel(),
                                                                int(average), st), "comm_allreduce_f32")

    def close(self) -> None:
        if self._handle is not None and self._handle.value:
            self._lib.check(self._lib.load().ctu_comm_destroy(self._handle), "comm_destroy")
        self._handle = None


class TorchCommunicator:
    """Same interface through torch.distributed's own RCCL process group ("nccl" backend on ROCm).  Only used when the C
    ABI's communicator cannot be brought up on this node (get_communicator says so loudly): still RCCL over xGMI, still no
    host copy -- the collective is enqueued on the caller's stream by making it torch's current stream."""
    backend = "torch.distributed (RCCL)"

    def __init__(self, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)

    def allreduce_(self, t: torch.Tensor, average: bool = True, stream=None) -> None:
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("ctunet_amd.parallel.TorchCommunicator: contiguous float32 GPU tensors only")
        with torch.cuda.stream(stream or torch.cuda.current_stream(t.device)):
            dist.all_reduce(t, op=dist.ReduceOp.AVG if average else dist.ReduceOp.SUM, group=self.group)

    def close(self) -> None:
        pass


Communicator.backend = "ctu_comm (RCCL behind the C ABI)"
_COMMS: Dict[object, object] = {}


def get_communicator(process_group=None):
    """The (cached) RCCL communicator of this rank for ``process_group``; collective on first use.  The C ABI's communicator
    (``ctu_comm_*``) by default.  Availability is AGREED ON before anyone enters a rendezvous: every rank reports whether the
    library loaded and librccl could be bound (rank 0 also whether it could create the unique id) through one MIN all-reduce;
    only if all say yes is the id broadcast and ``ncclCommInitRank`` entered -- by every rank.  Otherwise all ranks take
    torch.distributed's RCCL group instead and say so (``bench.py --gpus N`` treats that as an error unless told otherwise)."""
    c = _COMMS.get(process_group)
    if c is None:
        import ctypes as C
        rank = dist.get_rank(process_group)
        ok, err, ident = 1, None, None
        try:
            from . import _lib
            lib = _lib.load()
            if not lib.ctu_comm_available():
                raise RuntimeError("librccl could not be bound (ctu_comm_available() == 0)")
            if rank == 0:
                buf = (C.c_char * 128)()
                if lib.ctu_comm_unique_id(buf) != 0:
                    raise RuntimeError("rank 0 could not create an RCCL unique id: " + (lib.ctu_last_error() or b"").decode())
                ident = bytes(buf)
        except Exception as e:                                  # noqa: BLE001 -- reported below, on every rank
            ok, err = 0, e
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend(process_group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
        if int(flag.item()) == 0:
            import warnings
            warnings.warn(f"ctunet_amd.parallel: ctu_comm_* communicator unavailable on some rank ({err!r}); "
                          "gradient all-reduces go through torch.distributed's RCCL process group")
            c = TorchCommunicator(process_group)
        else:
            box = [ident]
            dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                       group=process_group)
            c = Communicator(process_group, box[0])
        if not _COMMS:
            import atexit
            atexit.register(close_communicators)               # (a caller that forgets close_communicators() still frees them)
        _COMMS[process_group] = c
    return c


def close_communicators() -> None:
    for c in _COMMS.values():
        c.close()
    _COMMS.clear()


class GradSync:
    """Bucketed mean-all-reduce of gradients produced block by block, overlapped with the rest of backward.

    The engine pushes each block's weight gradients as soon as their kernels are enqueued (head, decoder top -> bottom,
    centre, encoder bottom -> top); whenever ``bucket_bytes`` are pending the bucket is flattened and its all-reduce is
    launched on a side stream behind an event, so it runs under the backward kernels of the blocks that follow.  With the
    default 512 KiB, UNet()'s 3.3 MB of live gradients leave in four buckets (head + top three decoder blocks / deepest
    decoder block / deepest encoder block / the rest at ``finish``): everything but the last, 0.2 MB bucket overlaps the
    128^3 / 64^3 encoder backward.  GPU tensors go through the C ABI's RCCL communicator (``ctu_comm_allreduce_f32``,
    ncclAvg); CPU tensors (the gloo tests of this logic) through ``torch.distributed``."""

    def __init__(self, process_group=None, bucket_bytes: int = DEFAULT_BUCKET_BYTES):
        self.group = process_group
        self.bucket_bytes = int(bucket_bytes)
        self._pending: List[Tuple[str, torch.Tensor]] = []
        self._pending_bytes = 0
        self._inflight: List[Tuple[torch.Tensor, List[Tuple[str, torch.Tensor, int]], object]] = []
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.launched_before_finish = 0          # buckets whose collective was issued from push() (overlap evidence)

    # -- called by the engine as soon as a block's gradients have been enqueued
    def push(self, named_grads: Sequence[Tuple[str, torch.Tensor]]) -> None:
        if self.world == 1:
            return
        for name, g in named_grads:
            self._pending.append((name, g))
            self._pending_bytes += g.numel() * g.element_size()
        if self._pending_bytes >= self.bucket_bytes:
            self._launch()
            self.launched_before_finish += 1

    def _launch(self) -> None:
        if not self._pending:
            return
        items = self._pending
        self._pending, self._pending_bytes = [], 0
        ref = items[0][1]
        flat = torch.cat([g.reshape(-1) for _, g in items])       # one flattening launch per bucket
        layout, off = [], 0
        for name, g in items:
            layout.append((name, g, off))
            off += g.numel()
        if flat.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            self._comm_stream.wait_event(ev)
            get_communicator(self.group).allreduce_(flat, average=True, stream=self._comm_stream)
            done = torch.cuda.Event()
            done.record(self._comm_stream)
            flat.record_stream(self._comm_stream)
            work = done
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
            if flat.is_cuda:
                torch.cuda.current_stream(flat.device).wait_event(work)      # ncclAvg: already the mean
            else:
                work.wait()
                flat.mul_(inv)
            for name, g, off in layout:
                out[name] = flat[off:off + g.numel()].view_as(g)
        self._inflight = []
        return out


def distribute(module: torch.nn.Module, process_group=None, bucket_bytes: int = DEFAULT_BUCKET_BYTES,
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
    sync = GradSync(cfg[0], cfg[1])
    module.__dict__["_last_grad_sync"] = sync        # tests / bench read its launched_before_finish (overlap evidence)
    return sync


def allreduce_mean_(tensors: Sequence[Optional[torch.Tensor]], process_group=None,
                    bucket_bytes: int = DEFAULT_BUCKET_BYTES) -> None:
    """In-place mean over ranks of a list of tensors (``None`` entries are skipped on every rank).
    Stand-alone form of the bucket logic, for optimizers/gradients produced outside the engine."""
    sync = GradSync(process_group, bucket_bytes)
    live = [(str(i), t) for i, t in enumerate(tensors) if t is not None]
    sync.push(live)
    red = sync.finish()
    for name, t in live:
        if name in red:
            t.copy_(red[name])
