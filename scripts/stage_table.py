"""Per-conv-stage table of one train (or eval) step: stage (W, padded C_in, padded C_out) -> kernel symbol, launches per
step, microseconds, algorithmic TFLOP/s and GB/s, and -- where a counter profile of the same kernels is committed -- the
fabric bytes per launch and matrix-pipe busy share of that kernel symbol.

    python scripts/stage_table.py [--model UNet] [--size 128] [--dtype f32|bf16|f16] [--mode train|infer] [--batch 1]
                                  [--traffic profiles/r02_hbm_traffic.json] [--out profiles/r02_stage_table_f32.md]

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
ap.add_argument("--traffic", default=None)
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


for _ in range(3):
    step()
torch.cuda.synchronize()
ops.TIMER = ops.KernelTimer()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
tm = ops.TIMER
ops.TIMER = None
traffic = {}
if a.traffic and os.path.exists(a.traffic):
    traffic = json.load(open(a.traffic))["kernels"]
rows = sorted(tm.by_layer().items(), key=lambda kv: -kv[1]["total_ms"])
lines = [f"# conv stages of one {a.model}() {a.mode} step, {s}^3, batch {a.batch}, {a.dtype} (HIP events, {a.steps} eager steps)", "",
         "| stage (W, C_in_p, C_out_p) | kernel | launches/step | us/launch | ms/step | algorithmic TFLOP/s | algorithmic GB/s | counter MB/launch (symbol avg) |",
         "|---|---|---|---|---|---|---|---|"]
tot = 0.0
by_bytes = {}
for (tag, det), d in rows:
    by_bytes.setdefault((tag, det), 0.0)
for rec, det in zip(tm.records, tm.details):
    by_bytes[(rec[0], det)] = by_bytes.get((rec[0], det), 0.0) + rec[2]
for (tag, det), d in rows:
    n = d["launches"] / a.steps
    ms = d["total_ms"] / a.steps
    tot += ms
    tf = d["flops"] / (d["total_ms"] * 1e-3) / 1e12
    gbs = by_bytes[(tag, det)] / (d["total_ms"] * 1e-3) / 1e9
    k = traffic.get(tag.split(" (")[0])
    tr = f"{k['hbm_bytes_per_launch'] / 1e6:.1f}" if k else "-"
    lines.append(f"| {det} | `{tag}` | {n:g} | {d['total_ms'] / d['launches'] * 1e3:.1f} | {ms:.3f} | {tf:.1f} | {gbs:.0f} | {tr} |")
lines += ["", f"conv / ConvTranspose kernels total: {tot:.3f} ms/step"]
text = "\n".join(lines)
print(text)
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    open(a.out, "w").write(text + "\n")
