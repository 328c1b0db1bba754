"""dev: eager train-step time of the other shipped classes at 128^3 (secondary numbers for profiles/README.md)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ct-unet_amd")]
import torch
import ctunet_amd
from ctunet_amd import optim as O2, ProblemHandler as PH

class H:
    verbose = False
    def __init__(s):
        s.params = dict(ce_lambda=1.0, dice_lambda=1.0, save_dice_plots=False, save_hd_plots=False); s.losses_and_metrics = {}; s.pt_loss = None

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for name, cin, two in (("UNet", 1, False), ("UNetSP", 2, True), ("UNetSPSmall", 2, True), ("recAE_v2_fixed", 1, False), ("UNet4_2IC", 2, False)):
    torch.manual_seed(0)
    net = getattr(ctunet_amd, name)().cuda().train()
    opt = O2.Adam(net.parameters(), lr=1e-4)
    x = torch.randn(1, cin, size, size, size, device="cuda")
    lab = (torch.rand(1, size, size, size, device="cuda") < 0.2).long()
    t = torch.nn.functional.one_hot(lab, 2).movedim(-1, 1).float().contiguous()
    h = H()
    def step():
        out = net(x.detach().requires_grad_(True))
        if two:
            PH.FlapRecWithShapePriorDoubleOut.comp_losses_metrics(h, out, [t, t], 0, 1)
        else:
            PH.ProblemHandler.comp_losses_metrics(h, out, t, 0, 1)
        h.pt_loss.backward(); opt.step()
        for p in net.parameters(): p.grad = None
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name:16s} {size}^3 eager train step {dt*1e3:8.2f} ms  {size**3/dt/1e6:8.1f} Mvox/s", flush=True)
    del net, opt
    torch.cuda.empty_cache()
