"""Whole-step HIP graph: forward + loss + backward + optimizer of one fixed-shape batch, captured once
and replayed, so the ~270 kernel launches of a step cost one graph launch (MI355X guide: "capture
launch-bound inner loops in hipGraphs").

The step is the reference's ``Model.forward_pass`` train branch (ctunet/pytorch/Model.py:343-374) minus its
host round trips: the per-term ``float(loss)`` syncs become ONE device->host copy after the replay.

With ``distributed`` set (one process per GPU) the step is a CHAIN of graph segments cut at the gradient-bucket
boundaries of backward (head + upper decoder / deep decoder / deep encoder / rest, ``parallel.DEFAULT_BUCKET_BYTES``):
segment k ends by flattening bucket k into a static buffer; its all-reduce (RCCL through the C ABI, never captured) is
launched on a side stream behind an event and runs UNDER segment k+1; the last segment is the fused optimizer step,
which waits for every bucket.  Only the last, smallest bucket's collective is exposed.  The segments are captured by
driving the engine directly (forward, fused loss kernels, backward) on the capturing thread -- no autograd thread takes
part, so a capture can be ended and the next begun at a bucket boundary in the middle of backward.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .losses import fused_ce_dice


class _Segmenter:
    """The ``sync`` object ``UNetEngine.backward`` pushes gradients into: collects them into buckets and calls
    ``boundary(flat)`` whenever one is complete (and once more from ``finish``)."""

    def __init__(self, bucket_bytes: int, boundary):
        self.bucket_bytes, self.boundary = int(bucket_bytes), boundary
        self.pending: List[Tuple[str, torch.Tensor]] = []
        self.nbytes = 0
        self.layout: List[List[Tuple[str, torch.Size, int]]] = []
        self.flats: List[torch.Tensor] = []

    def _close(self) -> None:
        if not self.pending:
            return
        flat = torch.cat([g.reshape(-1) for _, g in self.pending])
        lay, off = [], 0
        for name, g in self.pending:
            lay.append((name, g.shape, off))
            off += g.numel()
        self.layout.append(lay)
        self.flats.append(flat)
        self.pending, self.nbytes = [], 0
        self.boundary(len(self.flats) - 1, flat)

    def push(self, named_grads) -> None:
        for name, g in named_grads:
            self.pending.append((name, g))
            self.nbytes += g.numel() * g.element_size()
        if self.nbytes >= self.bucket_bytes:
            self._close()

    def finish(self) -> Dict[str, torch.Tensor]:
        self._close()
        out = {}
        for flat, lay in zip(self.flats, self.layout):
            for name, shape, off in lay:
                out[name] = flat[off:off + shape.numel()].view(shape)
        return out


class GraphedTrainStep:
    """model: a ctunet_amd model on the GPU; optimizer: ctunet_amd.optim.Adam/AdamW (or torch.optim with capturable=True).
    Build it before any eager backward of the same model, or drop every reference to that iteration's autograd graph (its
    loss tensors) first: a live graph keeps its AccumulateGrad nodes on the default stream, and autograd's stream hand-over
    to them inside the capture ends the capture with a fault in the HIP runtime (measured, torch 2.10 / ROCm 7.2)."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, example_input: torch.Tensor,
                 example_targets: Sequence[torch.Tensor], ce_lambda: float, dice_lambda: float,
                 input_requires_grad: bool = True, warmup: int = 3, distributed: bool = False, process_group=None,
                 bucket_bytes: Optional[int] = None):
        self.model, self.opt = model, optimizer
        if hasattr(optimizer, "guard"):
            optimizer.guard(model)             # fp16: an overflowed step is skipped inside the replayed graph too
        self._params = [p for p in model.parameters()]
        self.distributed, self.group = distributed, process_group
        if distributed and model.__dict__.get("_grad_sync_cfg") is not None:
            raise RuntimeError("GraphedTrainStep(distributed=True) does its own all-reduce: do not also call "
                               "parallel.distribute() on the model (use parallel.broadcast_parameters)")
        self.ce, self.dice = float(ce_lambda), float(dice_lambda)
        self.x = example_input.detach().clone()
        self.targets = [t.detach().clone().contiguous() for t in example_targets]
        self.x_req = input_requires_grad
        self.double = len(self.targets) == 2
        self.values: Optional[torch.Tensor] = None
        self.skip_comm = False                 # measurement only (bench.py's comm_ms_exposed): replay without the collectives
        self.keys = self._keys()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        if not distributed:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with torch.cuda.graph(self.graph):
                self.values = self._step()
            return
        import torch.distributed as dist
        from .parallel import DEFAULT_BUCKET_BYTES, get_communicator
        self.world = dist.get_world_size(process_group)
        self.bucket_bytes = DEFAULT_BUCKET_BYTES if bucket_bytes is None else int(bucket_bytes)
        self.comm = get_communicator(process_group)
        self.comm_stream = torch.cuda.Stream()
        self.segments: List[torch.cuda.CUDAGraph] = []
        self.flats: List[torch.Tensor] = []
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):            # (the optimizer state must exist before its segment is captured)
                self._segmented_step(capture=False)
            torch.cuda.synchronize()
            # with a process group alive other threads of the process may issue HIP calls while this one captures (the
            # nccl backend's watchdog polls its events): thread-local capture mode, see DESIGN 6
            self._segmented_step(capture=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

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

    # ------------------------------------------------------------------ single graph (one GPU): through autograd
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
        self.opt.step()
        for p in self.model.parameters():
            p.grad = None
        return torch.stack([t.detach() for t in terms] + [loss.detach()])

    # ------------------------------------------------------------------ segmented (N > 1): the engine driven directly
    def _segmented_step(self, capture: bool) -> None:
        from .engine import _tensor_dict
        model = self.model
        eng = model._engine()
        P = _tensor_dict(model)
        pool = None

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=pool, capture_error_mode="thread_local")
            self.segments.append(g)
            return g

        def boundary(k: int, flat: torch.Tensor) -> None:
            nonlocal pool
            if capture:
                self.flats.append(flat)
                self.segments[-1].capture_end()
                if pool is None:
                    pool = self.segments[0].pool()
                begin()
            else:
                self._launch_allreduce(flat)

        if capture:
            begin()
        with torch.no_grad():
            out0, out1, ctx = eng.forward(P, self.x, True, True, bool(getattr(model, "chk", False)))
            outs = [out0] if out1 is None else [out0, out1]
            terms, gouts = [], []
            for o, t in zip(outs, self.targets):
                tt, ws = ops.loss_fwd(o, t, self.ce, self.dice, self.double)
                gouts.append(ops.loss_bwd(o, t, self.ce, self.dice, self.double, ws, None, None))
                terms.append(tt)
            # list order of the reference: ce terms first, then dice terms (ProblemHandler.py:252-273)
            tl = ([t[0] for t in terms] if self.ce else []) + ([t[1] for t in terms] if self.dice else [])
            loss = tl[0]
            for t in tl[1:]:
                loss = loss + t
            values = torch.stack(tl + [loss])
            seg = _Segmenter(self.bucket_bytes, boundary)
            grads, _ = eng.backward(P, ctx, gouts[0], gouts[1] if len(gouts) == 2 else None, self.x_req, seg)
            if not capture:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            eng.fold_overflow(grads, self.x.device)    # (capture: part of the last segment, replayed behind the last all-reduce)
            for name, p in model.named_parameters():
                p.grad = grads.get(name)
            self.opt.step()
            for p in self._params:
                p.grad = None
        if capture:
            self.segments[-1].capture_end()
            self.values = values

    def _launch_allreduce(self, flat: torch.Tensor) -> None:
        """Mean over ranks of one bucket on the side stream, behind everything enqueued on the current stream so far."""
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        self.comm_stream.wait_event(ev)
        self.comm.allreduce_(flat, average=True, stream=self.comm_stream)

    def __call__(self, x: Optional[torch.Tensor] = None, targets: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        """Copies the batch into the captured buffers, replays the step, returns the loss terms (device tensor,
        order ``self.keys``); call ``.tolist()`` on it for the reference's logged floats (one sync)."""
        if x is not None:
            self.x.detach().copy_(x)
        if targets is not None:
            for dst, src in zip(self.targets, targets):
                dst.copy_(src)
        if not self.distributed:
            self.graph.replay()
        else:
            cur = torch.cuda.current_stream()
            for k, flat in enumerate(self.flats):
                self.segments[k].replay()
                if not self.skip_comm:
                    self._launch_allreduce(flat)           # runs under segment k + 1
            cur.wait_stream(self.comm_stream)
            self.segments[-1].replay()                    # the fused optimizer step, reading the averaged buckets
        # the replayed optimizer kernel wrote the parameters through raw pointers: bump their version counters, so that
        # an eagerly launched forward after this replay (the reference's train-then-validate epoch loop,
        # Model.py:240-243) re-packs the engine's MFMA-ordered weight copies instead of hitting a stale cache entry
        torch.autograd.graph.increment_version(self._params)
        return self.values
