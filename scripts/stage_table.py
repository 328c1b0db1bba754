"""Per-conv-stage table of one train (or eval) step: stage (W, padded C_in, padded C_out) -> kernel symbol, launches per
step, microseconds, algorithmic TFLOP/s and GB/s, and -- from rocprofv3 counter passes over THIS script -- the HBM bytes
per launch and the matrix-pipe busy share of every stage.

    python scripts/stage_table.py [--model UNet] [--size 128] [--dtype f32|bf16|f16] [--mode train|infer] [--batch 1]
                                  [--counters FETCH_DIR,WRITE_DIR,BUSY_DIR] [--out profiles/r03_stage_table_f32.md]
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d FETCH_DIR -o r -- python scripts/stage_table.py --profiled ...
    (same with WRITE_SIZE -> WRITE_DIR and SQ_VALU_MFMA_BUSY_CYCLES -> BUSY_DIR; separate passes, kernel trace only)

Counter attribution: the timer brackets every conv-family launch from the very first step, so the i-th timed launch of a
kernel family is the i-th dispatch of that family in a counter pass of the same command line (the launch sequence is
deterministic); families whose dispatch count differs from the timer's record count are left blank rather than guessed.
FETCH_SIZE x2 and KiB -> bytes as MI355X_MICROARCH.md prescribes for gfx950; matrix-pipe busy % = SQ_VALU_MFMA_BUSY_CYCLES
(summed over the 1024 SIMDs) / (1024 x the launch's HIP-event duration x 2.4 GHz nominal).

HIP events around every conv / ConvTranspose launch (ops.KernelTimer, the same instrument bench.py's roofline leg
uses), eagerly launched steps.  The algorithmic figures are the reference's layers: 2 C_in C_out
k^3 FLOP and (C_in + C_out) x 4 (2 for 16-bit) bytes per output voxel."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ct-unet_amd")]
import torch
import ctunet_amd
from ctunet_amd import ProblemHandler, ops

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="UNet")
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"])
ap.add_argument("--mode", default="train", choices=["train", "infer"])
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--traffic", default=None, help="(old) per-symbol averages of scripts/collect_traffic.py")
ap.add_argument("--counters", default=None, help="FETCH_DIR,WRITE_DIR,BUSY_DIR of rocprofv3 passes over this command + --profiled")
ap.add_argument("--profiled", action="store_true", help="run the steps and exit (the run rocprofv3 wraps)")
ap.add_argument("--out", default=None)
a = ap.parse_args()

torch.manual_seed(0)
train = a.mode == "train"
net = getattr(ctunet_amd, a.model)().cuda().train(train).set_precision({"f32": "fp32", "bf16": "bf16", "f16": "fp16"}[a.dtype])
two = net._plan.head_mode != 0
s = a.size
x = torch.randn(a.batch, net._plan.in_ch, s, s, s, device="cuda")
tg = [torch.nn.functional.one_hot((torch.rand(a.batch, s, s, s, device="cuda") < 0.2).long(), 2).movedim(4, 1).float().contiguous()
      for _ in range(2 if two else 1)]


class H:
    verbose = False
    params = dict(ce_lambda=1.0, dice_lambda=1.0, save_dice_plots=False, save_hd_plots=False)
    losses_and_metrics = {}
    pt_loss = None


handler = ProblemHandler.FlapRecWithShapePriorDoubleOut if two else ProblemHandler.ProblemHandler


def step():
    if not train:
        with torch.no_grad():
            return net(x)
    for p in net.parameters():
        p.grad = None
    out = net(x.detach().requires_grad_(True))
    handler.comp_losses_metrics(H, out, tg if two else tg[0], 0, 1)
    H.pt_loss.backward()


WARM = 3
ops.TIMER = ops.KernelTimer()                # from the first launch on: the record order is what the counter passes are aligned by
for _ in range(WARM):
    step()
torch.cuda.synchronize()
tm = ops.TIMER
n_warm = len(tm.records)
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
ops.TIMER = None
if a.profiled:
    sys.exit(0)


def family(tag):
    """Substring every dispatch of the launch site carries in its kernel name (mangled or not)."""
    base = tag.split("<")[0].split(" ")[0]
    return {"upconv_fused_wgrad_kernel": "conv3d_wgrad_k3s_kernel", "lp_upconv_wgrad_kernel": "lp_upwg4_kernel",
            "lp_conv_fwd_kernel": "lp_conv_fwd", "lp_conv_fwd_pair_kernel": "lp_conv_fwd", "lp_conv_fwd_p1_kernel": "lp_conv_fwd",
            "lp_conv_fwd_small_kernel": "lp_conv_fwd", "lp_wgrad8_kernel": "lp_wgrad8_kernel",
            "convt2_fwd": "convt2_kernel", "convt2_bwd_data": "convt2_kernel", "convt2_wgrad": "convt2_wgrad_kernel",
            "convt2_fwd_lp": "lp_convt_fwd_kernel", "convt2_bwd_data_lp": "lp_convt_bwd_data_kernel",
            "convt2_wgrad_lp": "lp_convt_wgrad_kernel"}.get(base, base)


def dispatches(d, counter):
    """[(kernel name, counter value summed over its instances)] in dispatch order."""
    import csv
    import glob
    f = (glob.glob(f"{d}/*counter_collection.csv") + glob.glob(f"{d}/*/*counter_collection.csv"))[0]
    acc, names = {}, {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            i = int(r["Dispatch_Id"])
            acc[i] = acc.get(i, 0.0) + float(r["Counter_Value"])
            names[i] = r["Kernel_Name"]
    return [(names[i], acc[i]) for i in sorted(acc)]


def attribute(d, counter):
    """{record index: counter value}: i-th record of a family <-> i-th non-reduction dispatch of that family."""
    seq = dispatches(d, counter)
    fams = sorted({family(r[0]) for r in tm.records}, key=len, reverse=True)
    per = {f: [] for f in fams}
    for name, v in seq:
        if "reduce" in name or "pack" in name or "project" in name or "face_sums" in name:
            continue
        for f in fams:                       # longest family name first: lp_conv_fwd before conv_fwd-like prefixes
            if f in name:
                per[f].append(v)
                break
    out, seen = {}, {f: 0 for f in fams}
    want = {f: sum(1 for r in tm.records if family(r[0]) == f) for f in fams}
    for i, r in enumerate(tm.records):
        f = family(r[0])
        if len(per[f]) == want[f]:
            out[i] = per[f][seen[f]]
        seen[f] += 1
    for f in fams:
        if len(per[f]) != want[f]:
            print(f"[stage_table] {counter}: family {f}: {len(per[f])} dispatches vs {want[f]} timed launches -- left blank", file=sys.stderr)
    return out


fetch = write = busy = {}
if a.counters:
    fd, wd, bd = a.counters.split(",")
    fetch, write, busy = attribute(fd, "FETCH_SIZE"), attribute(wd, "WRITE_SIZE"), attribute(bd, "SQ_VALU_MFMA_BUSY_CYCLES")
traffic = {}
if a.traffic and os.path.exists(a.traffic):
    traffic = json.load(open(a.traffic))["kernels"]
ovh = tm._overhead_ms()
sites = {}
for i, ((tag, fl, nb, ea, eb), det) in enumerate(zip(tm.records, tm.details)):
    d = sites.setdefault((tag, det), dict(launches=0, total_ms=0.0, flops=0.0, bytes=0.0, rd=[], wr=[], busy=[]))
    if i in fetch:
        d["rd"].append(fetch[i] * 2048.0)
    if i in write:
        d["wr"].append(write[i] * 1024.0)
    if i in busy:
        d["busy"].append(busy[i])
    if i < n_warm:
        continue
    d["launches"] += 1
    d["total_ms"] += max(ea.elapsed_time(eb) - ovh, 1e-4)
    d["flops"] += fl
    d["bytes"] += nb
rows = sorted(sites.items(), key=lambda kv: -kv[1]["total_ms"])
lines = [f"# conv stages of one {a.model}() {a.mode} step, {s}^3, batch {a.batch}, {a.dtype} (HIP events, {a.steps} eager steps)", "",
         "| stage (W, C_in_p, C_out_p) | kernel | launches/step | us/launch | ms/step | algorithmic TFLOP/s | algorithmic GB/s | "
         "algorithmic MB/launch | counter MB/launch (read + written) | matrix pipe busy % |",
         "|---|---|---|---|---|---|---|---|---|---|"]
tot = 0.0
mean = lambda v: sum(v) / len(v)
for (tag, det), d in rows:
    if not d["launches"]:
        continue
    n = d["launches"] / a.steps
    ms = d["total_ms"] / a.steps
    us = d["total_ms"] / d["launches"] * 1e3
    tot += ms
    tf = d["flops"] / (d["total_ms"] * 1e-3) / 1e12
    gbs = d["bytes"] / (d["total_ms"] * 1e-3) / 1e9
    if d["rd"] and d["wr"]:
        tr = f"{(mean(d['rd']) + mean(d['wr'])) / 1e6:.1f} ({mean(d['rd']) / 1e6:.1f} + {mean(d['wr']) / 1e6:.1f})"
    else:
        k = traffic.get(tag.split(" (")[0])
        tr = f"{k['hbm_bytes_per_launch'] / 1e6:.1f} (symbol avg)" if k else "-"
    bz = f"{100.0 * mean(d['busy']) / (1024 * us * 1e-6 * 2.4e9):.0f}" if d["busy"] else "-"
    lines.append(f"| {det} | `{tag}` | {n:g} | {us:.1f} | {ms:.3f} | {tf:.1f} | {gbs:.0f} | {d['bytes'] / d['launches'] / 1e6:.1f} | {tr} | {bz} |")
lines += ["", f"conv / ConvTranspose kernels total: {tot:.3f} ms/step"]
text = "\n".join(lines)
print(text)
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    open(a.out, "w").write(text + "\n")
