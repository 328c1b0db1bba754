"""Diagnostic: error of the HIP path and of the fp32 CPU oracle against an fp64 oracle run (test infra)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ct-unet_amd"), os.path.join(ROOT, "tests")]
import torch
import ctunet_amd
from oracle import unet_oracle as O
from util import CLASS_INPUT, gen, onehot_target

name = sys.argv[1] if len(sys.argv) > 1 else "UNet"
s = int(sys.argv[2]) if len(sys.argv) > 2 else CLASS_INPUT[name][1]
in_ch = CLASS_INPUT[name][0]
torch.manual_seed(0)
net = getattr(ctunet_amd, name)()
net.chk = False
sd0 = {k: v.clone() for k, v in net.state_dict().items()}
x = torch.randn(1, in_ch, s, s, s, generator=gen(1234))
spec = O.SPECS[name]
two = spec.head != "plain"
tg = [onehot_target((1, 2, s, s, s), 4321 + i, 0.2) for i in range(2 if two else 1)]


def lossfn(dtype):
    t = [a.to(dtype) for a in tg]
    if two:
        return lambda o: O.loss_double(o, t, 1.0, 1.0)[0]
    if spec.out_ch == 2:
        return lambda o: O.loss_single(o, t[0], 1.0, 1.0)[0]
    return lambda o: (o ** 2).mean()


def run_oracle(dtype):
    sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    out, loss, g, dx = O.grads(spec, sd, x.to(dtype), lossfn(dtype), training=True)
    return out, loss, g, dx

o64, l64, g64, dx64 = run_oracle(torch.float64)
o32, l32, g32, dx32 = run_oracle(torch.float32)
net = net.cuda().train()
xi = x.cuda().requires_grad_(True)
out = net(xi)
loss = lossfn(torch.float32)([o for o in out] if two else out) if False else None
tgc = [t.cuda() for t in tg]
if two:
    loss = O.loss_double(out, tgc, 1.0, 1.0)[0]
elif spec.out_ch == 2:
    loss = O.loss_single(out, tgc[0], 1.0, 1.0)[0]
else:
    loss = (out ** 2).mean()
loss.backward()


def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()

print(f"{name} {s}^3 loss64={l64.item():.9f} cpu32={l32.item():.9f} hip={loss.item():.9f}")
outs = out if two else (out,)
o64s = o64 if two else (o64,)
o32s = o32 if two else (o32,)
for i in range(len(outs)):
    print(f"out{i}: cpu32 {err(o32s[i], o64s[i]):.2e}  hip {err(outs[i], o64s[i]):.2e}")
print(f"dx: cpu32 {err(dx32, dx64):.2e}  hip {err(xi.grad, dx64):.2e}")
for n_, p in net.named_parameters():
    if g64[n_] is None:
        continue
    print(f"{n_:32s} scale {g64[n_].abs().max().item():.2e}  cpu32 {err(g32[n_], g64[n_]):.2e}  hip {err(p.grad, g64[n_]):.2e}")
