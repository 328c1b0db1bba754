"""Per-parameter gradient error against an fp64 oracle run, in backward order (dev tool): where does a deviation enter?
usage: diag_grad_layers.py CLASS [seed] [size]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ct-unet_amd"), os.path.join(ROOT, "tests")]
import torch
import ctunet_amd
from ctunet_amd import ProblemHandler as PH
from oracle import unet_oracle as O
from util import CLASS_INPUT, gen, onehot_target

name = sys.argv[1]
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
size = int(sys.argv[3]) if len(sys.argv) > 3 else CLASS_INPUT[name][1]
torch.manual_seed(0)
net = getattr(ctunet_amd, name)()
net.chk = False
sd0 = {k: v.clone() for k, v in net.state_dict().items()}
in_ch = CLASS_INPUT[name][0]
x = torch.randn(1, in_ch, size, size, size, generator=gen(seed))
spec = O.SPECS[name]
two = spec.head != "plain"
handler = two or spec.out_ch == 2
tg = [onehot_target((1, 2, size, size, size), 4321 + i, 0.2) for i in range(2 if two else 1)]


def loss_fn(t):
    if two:
        return lambda o: O.loss_double(o, t, 1.0, 1.0)[0]
    if handler:
        return lambda o: O.loss_single(o, t[0], 1.0, 1.0)[0]
    return lambda o: (o ** 2).mean()


def run(dtype):
    t = [a.to(dtype) for a in tg]
    sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    return O.grads(spec, sd, x.to(dtype), loss_fn(t), training=True)


o32, l32, g32, dx32 = run(torch.float32)
o64, l64, g64, dx64 = run(torch.float64)


class H:
    verbose = False
    params = dict(ce_lambda=1.0, dice_lambda=1.0, save_dice_plots=False, save_hd_plots=False)
    losses_and_metrics = {}
    pt_loss = None


net = net.cuda().train()
xi = x.cuda().requires_grad_(True)
out = net(xi)
if two:
    PH.FlapRecWithShapePriorDoubleOut.comp_losses_metrics(H, out, [t.cuda() for t in tg], 0, 1)
    loss = H.pt_loss
elif handler:
    PH.ProblemHandler.comp_losses_metrics(H, out, tg[0].cuda(), 0, 1)
    loss = H.pt_loss
else:
    loss = (out ** 2).mean()
loss.backward()
outs = out if isinstance(out, tuple) else (out,)
refs = o64 if isinstance(o64, tuple) else (o64,)
for o, r in zip(outs, refs):
    print("out err vs fp64", float((o.detach().cpu().double() - r).abs().max()), " cpu32:", float((( o32 if not isinstance(o32, tuple) else o32[0]).double() - (r if len(refs) == 1 else refs[0])).abs().max()))
print("loss", loss.item(), l64.item(), l32.item())
names = [n for n, _ in net.named_parameters()][::-1] + ["dx"]
for n_ in names:
    got = xi.grad if n_ == "dx" else dict(net.named_parameters())[n_].grad
    r64 = dx64 if n_ == "dx" else g64[n_]
    c32 = dx32 if n_ == "dx" else g32[n_]
    if r64 is None:
        continue
    sc = r64.abs().max().item() + 1e-30
    eh = (got.detach().cpu().double() - r64).abs().max().item() / sc
    ec = (c32.double() - r64).abs().max().item() / sc
    flag = " <<<" if eh > max(5 * ec, 2e-3) else ""
    print(f"{n_:34s} scale {sc:9.2e}  hip {eh:9.2e}  cpu32 {ec:9.2e}{flag}")
