#!/usr/bin/env python3
"""Headline benchmark: voxels/sec, forward+backward, 128^3 fp32 patch (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" is what ``Model.forward_pass`` does per batch in the reference
(ctunet/pytorch/Model.py:343-374): input.requires_grad_(), forward of ``UNet()`` in train mode,
Dice + cross-entropy loss (lambda 1, 1), backward, gradient mean over ranks (N > 1),
Adam(amsgrad) step, grads -> None.  Per-GPU batch 1 of a synthetic 128^3 fp32 patch (the example inis
use i_batch_size = 1); weak scaling.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ct-unet_amd")]

import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_HBM_GBS = 8000.0             # spec; ~6300 achievable

# algorithmic work per input voxel, UNet() default, live graph, no recompute
FLOP_PER_VOXEL_FWD_BWD = 129.6e3


class Holder:
    """Stands in for ctunet.Model in comp_losses_metrics (Model.py:101,363)."""
    verbose = False

    def __init__(self):
        self.params = dict(ce_lambda=1.0, dice_lambda=1.0, save_dice_plots=False, save_hd_plots=False)
        self.losses_and_metrics = {}
        self.pt_loss = None


def synth_batch(size, rank, device):
    """Synthetic CT-like patch + one-hot target: seeds 1234+rank / 4321+rank."""
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(1, 1, size, size, size, generator=g)
    gt = torch.Generator().manual_seed(4321 + rank)
    m = (torch.rand(1, size, size, size, generator=gt) < 0.2).long()
    t = torch.nn.functional.one_hot(m, 2).movedim(4, 1).float().contiguous()
    return x.to(device), t.to(device)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may really use: min(affinity mask, cgroup cpu quota)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/r01_hbm_traffic.json, produced by scripts/collect_traffic.py with the guide's gfx950 corrections);
    None when no PMC run covers this kernel."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            k = json.load(f)["kernels"].get(kernel.split(" (")[0])
        return None if k is None else round(k["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(size, steps=5):
    """The oracle (same ATen-CPU graph as the reference, no checkpoint recompute) timed on the host
    cores: 1 warm-up + best of `steps` forward+backward steps of the same 128^3 workload."""
    from oracle import unet_oracle as O
    import ctunet_amd
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = ctunet_amd.UNet()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x, t = synth_batch(size, 0, "cpu")
    spec = O.SPECS["UNet"]
    best = float("inf")
    for i in range(steps + 1):
        t0 = time.perf_counter()
        O.grads(spec, sd, x, lambda o: O.loss_single(o, t, 1.0, 1.0)[0], training=True)
        dt = time.perf_counter() - t0
        log(f"cpu baseline step {i}: {dt:.2f} s on {cores} threads")
        if i > 0:
            best = min(best, dt)
    # "Dice vs CPU ref" (BASELINE metric): hard segmentation argmax(out, 1) of the HIP path against the
    # oracle's on identical weights and input (train-mode forward, the measured path), plus the raw-output error
    with torch.no_grad():
        ref = O.forward(spec, {k: v.clone() for k, v in sd.items()}, x, training=True, update_stats=False)
        net.load_state_dict(sd)
        got = net.cuda().train()(x.cuda()).cpu()
    a, b = got.argmax(1) == 1, ref.argmax(1) == 1
    tot = int(a.sum()) + int(b.sum())
    dice = 2.0 * int((a & b).sum()) / tot if tot else 1.0
    rel = float((got - ref).abs().max() / ref.abs().max())
    log(f"dice vs cpu ref {dice:.6f}, max rel output error {rel:.2e}")
    return {"value": size ** 3 / best, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": f"{steps} fwd+bwd steps (best, after 1 warm-up) of the same {size}^3 batch-1 UNet() train step, "
                      f"oracle = torch.nn.functional graph on ATen-CPU fp32, no checkpoint recompute, {best:.2f} s/step",
            "dice_vs_cpu_ref": dice, "max_rel_output_err": rel}


def main():
    # Libraries (RCCL's version banner, ...) write to fd 1; the contract is ONE JSON line on stdout.  Route fd 1 to
    # stderr for the whole run and keep the real stdout for the final line.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not replay the step from a HIP graph (N=1 default: graph)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1 or "RANK" in os.environ           # torchrun with 1 process also takes the N>1 code path
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import ctunet_amd
    from ctunet_amd import ProblemHandler, ops, parallel

    torch.manual_seed(0)
    net = ctunet_amd.UNet().to(dev).train()
    use_graph = not args.eager
    if distributed:
        if use_graph:
            parallel.broadcast_parameters(net)              # graph(fwd+bwd) -> flat RCCL all-reduce -> optimizer
        else:
            parallel.distribute(net)                        # eager: bucketed all-reduce overlapped with backward
    from ctunet_amd import optim as ctu_optim
    opt = ctu_optim.Adam(net.parameters(), lr=1e-4, weight_decay=0, amsgrad=True)         # Model.py:514-520, fused kernel
    x, target = synth_batch(args.size, rank, dev)
    holder = Holder()

    def eager_step():
        xi = x.detach().requires_grad_(True)                # Model.py:351-352
        out = net(xi)
        ProblemHandler.ProblemHandler.comp_losses_metrics(holder, out, target, 0, 1)   # one D2H sync for the logged floats
        holder.pt_loss.backward()
        opt.step()
        for p in net.parameters():                          # Model.py:373-374
            p.grad = None

    step = eager_step
    mode = "eager"
    if use_graph:
        # a failed capture is fatal (non-zero exit): the headline number is never silently measured on eager launches;
        # eager is the explicit --eager flag, decided identically on every rank before any step runs
        from ctunet_amd.graph import GraphedTrainStep
        gstep = GraphedTrainStep(net, opt, x, [target], 1.0, 1.0, input_requires_grad=True, distributed=distributed)

        def step():
            vals = gstep(x, [target])                   # same batch each step (synthetic), copied in like a loader would
            holder.losses_and_metrics.setdefault("epoch_loss", []).append(vals.tolist()[-1])   # one D2H sync
        mode = "hipgraph"

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: warm-up {args.warmup} steps")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log("timed region")
    timer = None
    if rank == 0 and not args.no_kernel_timer and mode == "eager":
        timer = ops.KernelTimer()                           # HIP events around every conv launch of the timed region
        ops.TIMER = timer
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ops.TIMER = None
    timer_steps = args.steps
    if rank == 0 and not args.no_kernel_timer and mode == "hipgraph":
        # Kernels inside a graph replay cannot be bracketed by events: time the SAME kernels (same launch shapes,
        # same buffers' sizes) in a few eagerly launched steps right after the timed region.
        timer_steps = min(5, args.steps)
        timer = ops.KernelTimer()
        ops.TIMER = timer
        for _ in range(timer_steps):
            eager_step()
        torch.cuda.synchronize()
        ops.TIMER = None
    if distributed:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    vox = world * args.size ** 3 * args.steps
    value = vox / dt
    log(f"{dt / args.steps * 1e3:.2f} ms/step, {value / 1e6:.1f} Mvox/s")

    if rank == 0:
        roofline = None
        kernels = {}
        if timer is not None:
            torch.cuda.synchronize()
            summ = timer.summary()
            for tag, d in summ.items():
                kernels[tag] = {"launches_per_step": d["launches"] / timer_steps, "avg_ms": round(d["avg_ms"], 4),
                                "ms_per_step": round(d["total_ms"] / timer_steps, 3),
                                "achieved_tflops": round(d["flops"] / (d["total_ms"] * 1e-3) / 1e12, 2)}
            # dominant kernel for the roofline leg: largest share of the step among the kernels whose algorithmic FLOPs are
            # the FLOPs they execute (the fused up-convolution symbols are credited with the two layers they replace)
            plain = {k: v for k, v in summ.items() if not k.startswith("upconv_fused")}
            dom = max((plain or summ).items(), key=lambda kv: kv[1]["total_ms"])
            ach = dom[1]["flops"] / (dom[1]["total_ms"] * 1e-3) / 1e12
            roofline = {"kernel": dom[0], "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                        "traffic": pmc_traffic(dom[0]),
                        "algorithmic_mb_per_launch": round(dom[1]["bytes"] / dom[1]["launches"] / 1e6, 1),
                        "avg_launch_ms": round(dom[1]["avg_ms"], 4),
                        "measured": ("HIP events around each launch, " +
                                     ("inside the timed region" if mode == "eager" else
                                      f"{timer_steps} eager steps right after the graph-replayed timed region")),
                        "algorithmic_gflop_per_launch": round(dom[1]["flops"] / dom[1]["launches"] / 1e9, 3)}
        line = {
            "metric": "voxels/sec fwd+bwd, 128^3 fp32 patch", "value": value, "unit": "voxels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"UNet() default (1 in, 2 out, i_size 8, 4 blocks), {args.size}^3 patch, batch 1 per "
                                   "GPU, train step = fwd + Dice/CE loss + bwd + grad all-reduce + Adam(amsgrad)",
                       "patch": args.size, "per_gpu_batch": 1, "parallelism": f"dp{world}", "launch": mode,
                       "whole_step_tflops_algorithmic": round(FLOP_PER_VOXEL_FWD_BWD * value / 1e12, 2),
                       "whole_step_frac_of_mfma_peak": round(FLOP_PER_VOXEL_FWD_BWD * value / world / 1e12 /
                                                             PEAK_FP32_MFMA_TFLOPS, 4),
                       "flop_accounting": "algorithmic = the reference's layers (SURVEY 8d); the fused decoder "
                                          "up-convolution kernels (upconv_fused_*) execute 3.9x fewer multiply-adds than the "
                                          "ConvTranspose3d + Conv3d pair they are credited with, so their TFLOP/s can exceed "
                                          "the MFMA peak; the roofline kernel is chosen among the unfused convolutions"},
            "roofline": roofline, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.size)
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if distributed:
        dist.barrier()
        parallel.close_communicators()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
