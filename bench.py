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



class Holder:
    """Stands in for ctunet.Model in comp_losses_metrics (Model.py:101,363)."""
    verbose = False

    def __init__(self):
        self.params = dict(ce_lambda=1.0, dice_lambda=1.0, save_dice_plots=False, save_hd_plots=False)
        self.losses_and_metrics = {}
        self.pt_loss = None


def synth_batch(size, rank, device, batch=1, in_ch=1, n_targets=1):
    """Synthetic CT-like patch + one-hot target(s): seeds 1234+rank / 4321+rank(+i)."""
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(batch, in_ch, size, size, size, generator=g)
    ts = []
    for i in range(n_targets):
        gt = torch.Generator().manual_seed(4321 + rank + 1000 * i)
        m = (torch.rand(batch, size, size, size, generator=gt) < 0.2).long()
        ts.append(torch.nn.functional.one_hot(m, 2).movedim(4, 1).float().contiguous().to(device))
    return x.to(device), ts


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
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/rNN_hbm_traffic.json, produced by scripts/collect_traffic.py with the guide's gfx950 corrections);
    None when no PMC run covers this kernel.  16-bit tags are looked up in the section of their own pass ("kernels_bf16")
    by kernel base name (rocprofv3 leaves symbols with __bf16 / _Float16 template arguments mangled or half-demangled:
    `...lp_conv_fwd_pair_kernelIDF16bEE...`), launch-weighted over the template instances that match."""
    import glob
    try:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")))
        with open(files[-1]) as f:
            doc = json.load(f)
        tag = kernel.split(" (")[0]
        dt = "bf16" if "<bf16" in tag else "f16" if "<f16" in tag else None
        # (the fp16 instantiations move the bytes of the bf16 ones: one 16-bit counter pass serves both)
        ks = doc.get("kernels" if dt is None else "kernels_" + dt, None) or doc.get("kernels_bf16" if dt else "kernels", {})
        if tag in ks:
            return round(ks[tag]["hbm_bytes_per_launch"])
        base = tag.split("<")[0]
        hit = [v for k, v in ks.items() if base + "<" in k or base + "I" in k] if dt else []
        n = sum(v["launches"] for v in hit)
        return round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit) / n) if n else None
    except Exception:
        return None


def cpu_baseline(size, model="UNet", precision="fp32", mode="train", batch=1, steps=3, check_only=False):
    """The oracle (same ATen-CPU graph as the reference) timed on the host cores, bounded: 1 warm-up + best of `steps`
    steps of the same workload.  Train mode reports three points: all granted cores without checkpoint
    recompute (`value`, the algorithmic 3x-forward work), the same with the recompute pass the reference's default
    use_checkpoint=True adds (every block's forward runs again inside backward, models.py:232-255: emulated as one extra
    no-grad forward per step), and the 8-thread point the survey container was measured at."""
    from oracle import unet_oracle as O
    import ctunet_amd
    cores = host_cores()
    torch.manual_seed(0)
    net = getattr(ctunet_amd, model)()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    spec = O.SPECS[model]
    two = spec.head != "plain"
    x, tg = synth_batch(size, 0, "cpu", batch, spec.in_ch, 2 if two else 1)
    fn = (lambda o: O.loss_double(o, tg, 1.0, 1.0)[0]) if two else (lambda o: O.loss_single(o, tg[0], 1.0, 1.0)[0])

    def timed(threads, recompute, n):
        torch.set_num_threads(threads)
        best = float("inf")
        for i in range(n + 1):
            t0 = time.perf_counter()
            if mode == "train":
                O.grads(spec, sd, x, fn, training=True)
                if recompute:
                    with torch.no_grad():
                        O.forward(spec, {k: v.clone() for k, v in sd.items()}, x, training=True, update_stats=False)
            else:
                with torch.no_grad():
                    O.forward(spec, sd, x, training=False)
            dt = time.perf_counter() - t0
            log(f"cpu baseline ({threads} threads{', +recompute' if recompute else ''}) step {i}: {dt:.2f} s")
            if i > 0:
                best = min(best, dt)
        return best
    best = timed(cores, False, 1 if check_only else steps)
    vox = batch * size ** 3
    out = {"value": vox / best, "unit": "voxels/s", "cores": cores, "kind": "port",
           "sample": f"{steps} {'fwd+bwd' if mode == 'train' else 'eval-forward'} steps (best, after 1 warm-up) of the same "
                     f"{size}^3 batch-{batch} {model}() {'train step' if mode == 'train' else 'forward'}, oracle = "
                     f"torch.nn.functional graph on ATen-CPU fp32, no checkpoint recompute, {best:.2f} s/step"}
    if check_only:
        out["sample"] = out["sample"].replace(f"{steps} ", "1 ", 1)
    if mode == "train" and not check_only:
        bc = timed(cores, True, 2)
        out["checkpoint_default"] = {"value": vox / bc, "s_per_step": round(bc, 2),
                                     "note": "reference default use_checkpoint=True: + one recompute forward per step (emulated)"}
        if cores != 8:
            b8 = timed(min(8, cores), False, 2)
            out["threads_8"] = {"value": vox / b8, "s_per_step": round(b8, 2)}
        torch.set_num_threads(cores)
    # "Dice vs CPU ref" (BASELINE metric): hard segmentation argmax(out, 1) of the measured HIP path (same
    # precision, same BatchNorm mode) against the oracle's on identical weights and input, plus the raw-output error
    with torch.no_grad():
        ref = O.forward(spec, {k: v.clone() for k, v in sd.items()}, x, training=mode == "train", update_stats=False)
        net.load_state_dict(sd)
        net = net.cuda().train(mode == "train").set_precision(precision)
        got = net(x.cuda())
    refs = ref if isinstance(ref, tuple) else (ref,)
    gots = got if isinstance(got, tuple) else (got,)
    dices, rels = [], []
    for g_, r_ in zip(gots, refs):
        g_ = g_.cpu()
        a_, b_ = O.argmax1(g_) == 1, O.argmax1(r_) == 1
        tot = int(a_.sum()) + int(b_.sum())
        dices.append(2.0 * int((a_ & b_).sum()) / tot if tot else 1.0)
        rels.append(float((g_ - r_).abs().max() / r_.abs().max()))
    log(f"dice vs cpu ref {min(dices):.6f}, max rel output error {max(rels):.2e}")
    out["dice_vs_cpu_ref"], out["max_rel_output_err"] = min(dices), max(rels)
    return out


# algorithmic forward+backward kFLOP per input voxel (= 3x forward, live graph, no recompute) and the fused-ideal
# forward HBM bytes per voxel in fp32, per model class
ALGO = {"UNet": (129.6e3, 694), "UNetSP": (100.6e3, 617), "UNetDO": (99.5e3, 613), "UNetSPSmall": (35.2e3, 363),
        "recAE_v2_fixed": (570.6e3, 696), "UNet4_2IC": (442.8e3, 614)}     # the classes the example inis name
PEAK_16BIT_MFMA_TFLOPS = 2500.0    # dense bf16 / fp16 (MI355X_MICROARCH.md)


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
    ap.add_argument("--cpu-check-only", action="store_true",
                    help="cpu_baseline leg: ONE oracle step (its time is reported as the bounded sample) plus the Dice / "
                         "output-error check against the CPU reference -- for the large cfg 4 / 5 legs")
    ap.add_argument("--allow-torch-comm", action="store_true",
                    help="N > 1: accept torch.distributed's RCCL group if the C ABI's communicator (ctu_comm_*) cannot be "
                         "brought up on every rank (default: exit non-zero -- the measured path is the C-ABI one)")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not replay the step from a HIP graph (default: graph)")
    # secondary legs (the defaults above ARE the headline: UNet(), 128^3, fp32, batch 1, train step)
    ap.add_argument("--model", default="UNet", choices=sorted(ALGO), help="drop-in class to run")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"], help="activation storage type (cfg 4 / 5)")
    ap.add_argument("--mode", default="train", choices=["train", "infer"], help="infer: eval-mode forward only (cfg 2)")
    ap.add_argument("--batch", type=int, default=1, help="per-GPU batch")
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

    precision = {"f32": "fp32", "bf16": "bf16", "f16": "fp16"}[args.dtype]
    train = args.mode == "train"
    torch.manual_seed(0)
    net = getattr(ctunet_amd, args.model)().to(dev).train(train).set_precision(precision)
    in_ch, two = net._plan.in_ch, net._plan.head_mode != 0   # (bg, flap, full) heads return two 2-channel maps
    use_graph = not args.eager
    if distributed and train:
        if use_graph:
            parallel.broadcast_parameters(net)              # graph(fwd+bwd) -> flat RCCL all-reduce -> optimizer
        else:
            parallel.distribute(net)                        # eager: bucketed all-reduce overlapped with backward
    from ctunet_amd import optim as ctu_optim
    opt = ctu_optim.Adam(net.parameters(), lr=1e-4, weight_decay=0, amsgrad=True).guard(net)     # Model.py:514-520, fused kernel
    x, targets = synth_batch(args.size, rank, dev, args.batch, in_ch, 2 if two else 1)
    holder = Holder()
    handler = ProblemHandler.FlapRecWithShapePriorDoubleOut if two else ProblemHandler.ProblemHandler

    def eager_step():
        if not train:
            with torch.no_grad():
                return net(x)
        xi = x.detach().requires_grad_(True)                # Model.py:351-352
        out = net(xi)
        handler.comp_losses_metrics(holder, out, targets if two else targets[0], 0, 1)   # one D2H sync for the logged floats
        holder.pt_loss.backward()
        opt.step()
        for p in net.parameters():                          # Model.py:373-374
            p.grad = None

    step = eager_step
    mode = "eager"
    if use_graph and train:
        # a failed capture is fatal (non-zero exit): the headline number is never silently measured on eager launches;
        # eager is the explicit --eager flag, decided identically on every rank before any step runs
        from ctunet_amd.graph import GraphedTrainStep
        gstep = GraphedTrainStep(net, opt, x, targets, 1.0, 1.0, input_requires_grad=True, distributed=distributed)
        if distributed and world > 1 and isinstance(gstep.comm, parallel.TorchCommunicator) and not args.allow_torch_comm:
            # the same decision on every rank (get_communicator agreed on it): the C-ABI communicator is the measured path
            raise SystemExit("bench.py: the ctu_comm_* RCCL communicator could not be brought up on every rank and the gradient "
                             "exchange fell back to torch.distributed's process group; pass --allow-torch-comm to measure that")

        def step():
            vals = gstep(x, targets)                    # same batch each step (synthetic), copied in like a loader would
            holder.losses_and_metrics.setdefault("epoch_loss", []).append(vals.tolist()[-1])   # one D2H sync
        mode = "hipgraph"
    elif use_graph:
        with torch.no_grad():
            for _ in range(2):
                net(x)
            torch.cuda.synchronize()
            g_inf = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_inf):
                y_static = net(x)

        def step():
            g_inf.replay()
            return y_static
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
        # two untimed eager steps first: outside the graph's private pool the caching allocator has to hipMalloc its blocks
        # once (synchronous; at 256^3 that stall landed between the event pairs: 1.5 ms "per launch" for a 0.24 ms kernel)
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()
        timer = ops.KernelTimer()
        ops.TIMER = timer
        for _ in range(timer_steps):
            eager_step()
        torch.cuda.synchronize()
        ops.TIMER = None
    comm_ms_exposed = None
    if distributed and mode == "hipgraph" and train:
        # exposed communication = step time with the bucket all-reduces minus the same graph segments replayed without
        # them (after the timed region; those steps skip the averaging and are not part of any reported number)
        gstep.skip_comm = True
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_nc = time.perf_counter() - t1
        gstep.skip_comm = False
        comm_ms_exposed = (dt - dt_nc) / args.steps * 1e3
    if distributed:
        tt = torch.tensor([dt, comm_ms_exposed if comm_ms_exposed is not None else 0.0], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt[0].item()
        if comm_ms_exposed is not None:
            comm_ms_exposed = tt[1].item()
    vox = world * args.batch * args.size ** 3 * args.steps
    value = vox / dt
    log(f"{dt / args.steps * 1e3:.2f} ms/step, {value / 1e6:.1f} Mvox/s")

    if rank == 0:
        lowp = args.dtype != "f32"
        flop_vox, bytes_vox = ALGO[args.model]
        if not train:
            flop_vox /= 3.0
        roofline = None
        kernels = {}
        if timer is not None:
            torch.cuda.synchronize()
            summ = timer.summary()
            for tag, d in summ.items():
                kernels[tag] = {"launches_per_step": d["launches"] / timer_steps, "avg_ms": round(d["avg_ms"], 4),
                                "ms_per_step": round(d["total_ms"] / timer_steps, 3),
                                "achieved_tflops": round(d["flops"] / (d["total_ms"] * 1e-3) / 1e12, 2),
                                "achieved_gbs": round(d["bytes"] / (d["total_ms"] * 1e-3) / 1e9, 1)}
            # dominant kernel for the roofline leg: largest share of the step among the kernels whose algorithmic FLOPs are
            # the FLOPs they execute (the fused up-convolution symbols are credited with the two layers they replace)
            plain = {k: v for k, v in summ.items() if not k.startswith(("upconv_fused", "lp_upconv"))}
            dom = max((plain or summ).items(), key=lambda kv: kv[1]["total_ms"])
            meas = ("HIP events around each launch, " +
                    ("inside the timed region" if mode == "eager" else
                     f"{timer_steps} eager steps (after 2 untimed ones) right after the graph-replayed timed region"))
            if lowp:
                # 16-bit activations: every 128^3 / 64^3 stage is HBM-bound -- bytes, not FLOPs
                ach = dom[1]["bytes"] / (dom[1]["total_ms"] * 1e-3) / 1e9
                roofline = {"kernel": dom[0], "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": pmc_traffic(dom[0]),
                            "algorithmic_mb_per_launch": round(dom[1]["bytes"] / dom[1]["launches"] / 1e6, 1),
                            "avg_launch_ms": round(dom[1]["avg_ms"], 4), "measured": meas,
                            "achieved_tflops": round(dom[1]["flops"] / (dom[1]["total_ms"] * 1e-3) / 1e12, 1)}
            else:
                ach = dom[1]["flops"] / (dom[1]["total_ms"] * 1e-3) / 1e12
                roofline = {"kernel": dom[0], "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                            "traffic": pmc_traffic(dom[0]),
                            "algorithmic_mb_per_launch": round(dom[1]["bytes"] / dom[1]["launches"] / 1e6, 1),
                            "avg_launch_ms": round(dom[1]["avg_ms"], 4), "measured": meas,
                            "algorithmic_gflop_per_launch": round(dom[1]["flops"] / dom[1]["launches"] / 1e9, 3)}
        what = ("train step = fwd + Dice/CE loss + bwd + grad all-reduce + Adam(amsgrad)" if train else
                "eval-mode forward (BatchNorm folded into the consumers' loads), no loss")
        desc = {"UNet": "UNet() default (1 in, 2 out, i_size 8, 4 blocks)"}.get(args.model, args.model + "()")
        headline = train and args.model == "UNet" and args.dtype == "f32" and args.batch == 1
        line = {
            # (a secondary leg always names its model, mode and batch: it can never read as the headline line)
            "metric": "voxels/sec fwd+bwd, 128^3 fp32 patch" if headline else
                      f"voxels/sec {'fwd+bwd' if train else 'fwd'}, {args.size}^3 {precision} patch, {args.model}() batch {args.batch}"
                      f"{'' if train else ' eval'} (secondary leg)",
            "value": value, "unit": "voxels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{desc}, {args.size}^3 patch, batch {args.batch} per GPU, {what}",
                       "patch": args.size, "per_gpu_batch": args.batch, "parallelism": f"dp{world}", "launch": mode,
                       "comm_ms_exposed": None if comm_ms_exposed is None else round(comm_ms_exposed, 4),
                       "grad_buckets": (len(gstep.flats) if distributed and mode == "hipgraph" and train else None),
                       "grad_bucket_bytes": ([int(f.numel()) * 4 for f in gstep.flats] if distributed and mode == "hipgraph" and train else None),
                       "comm_backend": (getattr(gstep.comm, "backend", None) if distributed and mode == "hipgraph" and train else None),
                       "precision_note": (None if args.dtype == "f32" else
                                          "16-bit activations / activation gradients in HBM and LDS, v_mfma_f32_16x16x32 with fp32 accumulation; "
                                          "fp32 BatchNorm statistics, master weights, weight gradients and optimizer state; every kernel of "
                                          "the step reads and writes the 16-bit tensors directly (no fp32 copies)"),
                       # peak device memory of this process over warm-up + timed steps (allocator high-water mark, all pools)
                       "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 1e9, 3),
                       "whole_step_tflops_algorithmic": round(flop_vox * value / 1e12, 2),
                       "whole_step_frac_of_mfma_peak": round(flop_vox * value / world / 1e12 /
                                                             (PEAK_16BIT_MFMA_TFLOPS if lowp else PEAK_FP32_MFMA_TFLOPS), 4),
                       "whole_step_algorithmic_gbs": round((3 if train else 1) * bytes_vox * (0.5 if lowp else 1.0) * value / world / 1e9, 1),
                       "flop_accounting": "algorithmic = the reference's layers (SURVEY 8d); the fused decoder "
                                          "up-convolution kernels (upconv_fused_*) execute 3.9x fewer multiply-adds than the "
                                          "ConvTranspose3d + Conv3d pair they are credited with, so their TFLOP/s can exceed "
                                          "the MFMA peak; the roofline kernel is chosen among the unfused convolutions"},
            "roofline": roofline, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.size, args.model, precision, args.mode, args.batch, check_only=args.cpu_check_only)
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if distributed:
        dist.barrier()
        parallel.close_communicators()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
