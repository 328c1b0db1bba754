"""Per-layer conv timing of one UNet() train step at 128^3 (dev tool): HIP events around every conv launch, eager."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ct-unet_amd")]
import torch
from ctunet_amd import ops, models, losses

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
net = models.UNet().cuda().train()
x = torch.randn(1, 1, size, size, size, device="cuda")
lab = torch.randint(0, 2, (1, size, size, size), device="cuda")
t = torch.stack([1 - lab, lab], 1).float()
def step():
    net.zero_grad(set_to_none=True)
    out = net(x)
    ce, dc = losses.fused_ce_dice(out, t, 1.0, 1.0, True)
    (ce + dc).backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
ops.TIMER = ops.KernelTimer()
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
rows = sorted(ops.TIMER.by_layer().items(), key=lambda kv: -kv[1]["total_ms"])
tot = 0.0
for (tag, det), d in rows:
    ms = d["total_ms"] / n
    tot += ms
    print(f"{tag[:58]:58s} W,cin_p,cout_p={str(det):16s} x{d['launches'] // n:2d} {ms * 1e3:8.1f} us/step  {d['flops'] / d['total_ms'] / 1e9:7.1f} TFLOP/s")
print("total conv ms/step", tot)
