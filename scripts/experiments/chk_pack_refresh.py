import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/ct-unet_amd"]
import torch, ctunet_amd
from ctunet_amd import optim as O2, losses
x = torch.randn(1, 1, 32, 32, 32, generator=torch.Generator().manual_seed(1)).cuda()
lab = (torch.rand(1, 32, 32, 32, generator=torch.Generator().manual_seed(2)) < 0.3).long()
t = torch.nn.functional.one_hot(lab, 2).movedim(-1, 1).float().cuda()
def run(kind):
    torch.manual_seed(0)
    net = ctunet_amd.UNet(n_blocks=2, use_checkpoint=False).cuda().train()
    opt = O2.Adam(net.parameters(), lr=1e-2) if kind == "fused" else torch.optim.Adam(net.parameters(), lr=1e-2, amsgrad=True)
    out = []
    p0 = next(net.parameters())
    for i in range(4):
        ce, dc = losses.fused_ce_dice(net(x), t, 1.0, 1.0, False)
        (ce + dc).backward()
        v0 = p0._version
        opt.step()
        out.append((round((ce + dc).item(), 6), p0._version - v0))
        for p in net.parameters(): p.grad = None
    return out
print("fused", run("fused"))
print("torch", run("torch"))
