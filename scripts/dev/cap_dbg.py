"""dev: capture a small train step in a HIP graph.  usage: cap_dbg.py <dtype> <loss_scale|none> <n_blocks> <size> [eager_first]"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ct-unet_amd")]
import torch
import ctunet_amd
from ctunet_amd import losses as L, optim
from ctunet_amd.graph import GraphedTrainStep
dt, ls, nb, s = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
eager_first = len(sys.argv) > 5 and sys.argv[5] == "1"
torch.manual_seed(0)
net = ctunet_amd.UNet(n_blocks=nb, use_checkpoint=False).cuda().train()
x = torch.randn(1, 1, s, s, s).cuda()
t = torch.nn.functional.one_hot((torch.rand(1, s, s, s) < 0.3).long(), 2).movedim(4, 1).float().contiguous().cuda()
net.set_precision({"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dt], loss_scale=None if ls == "none" else float(ls))
opt = optim.Adam(net.parameters(), lr=1e-3, amsgrad=True).guard(net)
if eager_first:
    ce, dc = L.fused_ce_dice(net(x), t, 1.0, 1.0, False)
    (ce + dc).backward()
    opt.step()
    torch.cuda.synchronize()
    print("eager ok", flush=True)
    del ce, dc
    for p in net.parameters():
        p.grad = None
g = GraphedTrainStep(net, opt, x, [t], 1.0, 1.0, input_requires_grad=True)
print("captured", flush=True)
g(x, [t]); g(x, [t])
torch.cuda.synchronize()
print("replayed ok", sys.argv[1:], flush=True)
