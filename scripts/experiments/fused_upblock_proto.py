"""Prototype (CPU, fp64) of the algebraic fusion planned for the decoder's ConvTranspose3d(k2,s2) -> Conv3d(k3,p1) pair
(models.py:37-38): on the coarse grid the pair is, per output parity p in {0,1}^3, a 2x2x2 convolution

    y[2i+p] = sum_{d in D(p)} x[i+d] . W_eff[p][d]  +  b_eff[border class of 2i+p]
    W_eff[p][d][ci,co] = sum_{(t,a) in S(p,d)} sum_cm WT[ci,cm,a] W3[co,cm,t]        (per axis: p=0: d=-1 <- (t=-1,a=1);
                                                                                       d=0 <- (t=0,a=0),(t=1,a=1);
                                                                                       p=1: d=0 <- (t=-1,a=0),(t=0,a=1);
                                                                                       d=+1 <- (t=1,a=0))
    b_eff[class][co] = sum_{t valid in class} sum_cm bT[cm] W3[co,cm,t]              (class = which of t=-1 / t=+1 fall
                                                                                       outside the fine volume, per axis)

i.e. 8 taps instead of 27 + the transposed conv, and no fine-grid intermediate: 4096 instead of 15872 FLOP per fine voxel
for 32 -> 32 -> 8.  The script checks the forward against torch's two ops and the weight/bias gradient projections
(dWT, dW3, dbT from dW_eff, db_eff) against autograd of the unfused composition.  Nothing here is product code.
"""
import itertools
import torch
import torch.nn.functional as F

torch.manual_seed(0)
DT = torch.float64

# per axis: parity -> list of (coarse offset d, [(t, a), ...]) with t in {-1,0,1} (conv3 tap), a in {0,1} (convT tap)
AXIS = {0: [(-1, [(-1, 1)]), (0, [(0, 0), (1, 1)])],
        1: [(0, [(-1, 0), (0, 1)]), (1, [(1, 0)])]}


def build_weff(WT, W3):
    """W_eff[p (8)][d (8, index = 4*jz+2*jy+jx over the two offsets of each axis)] -> [C, Co]."""
    C, Cm, Co = WT.shape[0], WT.shape[1], W3.shape[0]
    weff = torch.zeros(2, 2, 2, 2, 2, 2, C, Co, dtype=DT)
    for pz, py, px in itertools.product((0, 1), repeat=3):
        for jz, (dz, sz) in enumerate(AXIS[pz]):
            for jy, (dy, sy) in enumerate(AXIS[py]):
                for jx, (dx, sx) in enumerate(AXIS[px]):
                    acc = torch.zeros(C, Co, dtype=DT)
                    for (tz, az), (ty, ay), (tx, ax) in itertools.product(sz, sy, sx):
                        acc += WT[:, :, az, ay, ax] @ W3[:, :, tz + 1, ty + 1, tx + 1].T
                    weff[pz, py, px, jz, jy, jx] = acc
    return weff


def border_classes(n_fine):
    """per axis and fine index f: 0 interior, 1 low border (t=-1 outside), 2 high border (t=+1 outside)."""
    cls = torch.zeros(n_fine, dtype=torch.long)
    cls[0] = 1
    cls[-1] = 2
    return cls


def build_beff(bT, W3):
    """b_eff[cz][cy][cx][co]: taps that fall outside the fine volume contribute nothing (zero padding of the conv)."""
    Co = W3.shape[0]
    valid = {0: (0, 1, 2), 1: (1, 2), 2: (0, 1)}            # tap indices kept per class
    beff = torch.zeros(3, 3, 3, Co, dtype=DT)
    for cz, cy, cx in itertools.product(range(3), repeat=3):
        w = W3[:, :, valid[cz]][:, :, :, valid[cy]][:, :, :, :, valid[cx]]
        beff[cz, cy, cx] = torch.einsum("omzyx,m->o", w, bT)
    return beff


def fused_forward(x, weff, beff):
    N, C, D, H, W = x.shape
    Co = weff.shape[-1]
    xp = F.pad(x, (1, 1, 1, 1, 1, 1))
    y = torch.zeros(N, Co, 2 * D, 2 * H, 2 * W, dtype=DT)
    for pz, py, px in itertools.product((0, 1), repeat=3):
        # offsets of parity p start at d0 = p - 1 on every axis: a 2x2x2 conv on the padded coarse grid
        k = weff[pz, py, px].permute(4, 3, 0, 1, 2)          # [Co, C, 2, 2, 2]
        sub = xp[:, :, pz:pz + D + 1, py:py + H + 1, px:px + W + 1]
        y[:, :, pz::2, py::2, px::2] = F.conv3d(sub, k)
    cz, cy, cx = border_classes(2 * D), border_classes(2 * H), border_classes(2 * W)
    bias = beff[cz][:, cy][:, :, cx]                           # [2D, 2H, 2W, Co]
    return y + bias.permute(3, 0, 1, 2).unsqueeze(0)


def project_grads(dweff, dbeff, WT, bT, W3):
    """dWT, dW3, dbT from the gradients w.r.t. the composite weights / border-class biases."""
    dWT, dW3, dbT = torch.zeros_like(WT), torch.zeros_like(W3), torch.zeros_like(bT)
    for pz, py, px in itertools.product((0, 1), repeat=3):
        for jz, (dz, sz) in enumerate(AXIS[pz]):
            for jy, (dy, sy) in enumerate(AXIS[py]):
                for jx, (dx, sx) in enumerate(AXIS[px]):
                    g = dweff[pz, py, px, jz, jy, jx]          # [C, Co]
                    for (tz, az), (ty, ay), (tx, ax) in itertools.product(sz, sy, sx):
                        dWT[:, :, az, ay, ax] += g @ W3[:, :, tz + 1, ty + 1, tx + 1]
                        dW3[:, :, tz + 1, ty + 1, tx + 1] += g.T @ WT[:, :, az, ay, ax]
    valid = {0: (0, 1, 2), 1: (1, 2), 2: (0, 1)}
    for cz, cy, cx in itertools.product(range(3), repeat=3):
        g = dbeff[cz, cy, cx]                                   # [Co]
        for tz, ty, tx in itertools.product(valid[cz], valid[cy], valid[cx]):
            dW3[:, :, tz, ty, tx] += torch.outer(g, bT)
            dbT += W3[:, :, tz, ty, tx].T @ g
    return dWT, dW3, dbT


def main():
    N, C, Cm, Co, D, H, W = 2, 6, 6, 5, 3, 4, 5
    x = torch.randn(N, C, D, H, W, dtype=DT, requires_grad=True)
    WT = torch.randn(C, Cm, 2, 2, 2, dtype=DT, requires_grad=True)
    bT = torch.randn(Cm, dtype=DT, requires_grad=True)
    W3 = torch.randn(Co, Cm, 3, 3, 3, dtype=DT, requires_grad=True)
    ref = F.conv3d(F.conv_transpose3d(x, WT, bT, stride=2), W3, padding=1)
    weff = build_weff(WT.detach(), W3.detach()).requires_grad_(True)
    beff = build_beff(bT.detach(), W3.detach()).requires_grad_(True)
    xf = x.detach().clone().requires_grad_(True)
    got = fused_forward(xf, weff, beff)
    print("forward max abs err", float((got - ref).abs().max()))
    dy = torch.randn_like(ref)
    ref.backward(dy)
    got.backward(dy)
    dWT, dW3, dbT = project_grads(weff.grad, beff.grad, WT.detach(), bT.detach(), W3.detach())
    for name, a, b in (("dx", xf.grad, x.grad), ("dWT", dWT, WT.grad), ("dW3", dW3, W3.grad), ("dbT", dbT, bT.grad)):
        print(name, "max abs err", float((a - b).abs().max()), "scale", float(b.abs().max()))
    flops_direct = 2 * 27 * 32 * 8 + 2 * 32 * 32
    print("FLOP per fine voxel, 32->32->8: direct", flops_direct, "fused", 2 * 8 * 32 * 8, "ratio", flops_direct / (2 * 8 * 32 * 8))


if __name__ == "__main__":
    main()
