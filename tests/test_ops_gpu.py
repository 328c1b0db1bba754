"""Per-op parity of the HIP kernels (through the C ABI) against the oracle's building blocks
(torch.nn.functional on CPU, fp32).  Tolerances: the MFMA path is an exact fp32 fma chain, so
differences are summation-order only; gates are 1e-4 relative to the output scale, two orders
tighter than the north star's 1e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from ctunet_amd import ops
    return ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def to_cl(x, cs=None, c0=0, cp=None):
    """CPU NCDHW -> GPU channels-last buffer [N,D,H,W,cs] with x at channels [c0, c0+C), rest NaN-free garbage=0."""
    ops = _ops()
    n, c, d, h, w = x.shape
    cp = cp or ops.pad8(c)
    cs = cs or cp
    buf = torch.zeros(n, d, h, w, cs)
    buf[..., c0:c0 + c] = x.permute(0, 2, 3, 4, 1)
    return ops.CL(buf.cuda(), c0, cp)


def from_cl(a, c):
    return a.buf[..., a.c0:a.c0 + c].permute(0, 4, 1, 2, 3).cpu()


def rel_err(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def xf_vectors(c, cp, seed):
    sc = torch.zeros(cp); sh = torch.zeros(cp)
    sc[:c] = torch.rand(c, generator=g(seed)) * 2 - 0.5
    sh[:c] = torch.randn(c, generator=g(seed + 1)) * 0.3
    return sc, sh


CONV_CASES = [
    # N, Ci, Co, D, H, W, k, xf, bias
    (1, 8, 8, 8, 8, 16, 3, False, False),
    (1, 8, 8, 4, 8, 32, 3, True, False),       # W >= 32, C_out 8: "pair" layout kernel
    (2, 32, 7, 6, 5, 40, 3, True, True),       # pair layout, ragged box, 4 chunks, bias, 7 real channels
    (1, 1, 8, 8, 4, 70, 3, False, False),
    (1, 1, 8, 16, 16, 16, 3, False, False),
    (2, 8, 16, 8, 12, 20, 3, True, False),     # ragged: H, W not multiples of the tile
    (1, 32, 8, 8, 8, 16, 3, True, False),      # the dominant decoder shape
    (1, 16, 32, 8, 8, 8, 3, True, False),      # 8-wide tile
    (1, 64, 64, 4, 4, 4, 3, True, False),      # 4-wide tile, NT=4
    (1, 64, 128, 4, 4, 4, 3, False, False),    # two y-blocks
    (1, 7, 14, 8, 8, 8, 3, True, False),       # UNetSP widths
    (1, 3, 6, 2, 2, 2, 3, False, False),       # volume smaller than every tile
    (1, 8, 8, 8, 8, 16, 5, False, True),       # legacy k5 + bias
    (1, 2, 7, 6, 10, 18, 5, True, True),
    (1, 16, 64, 4, 4, 8, 5, True, True),
    (1, 8, 8, 8, 8, 32, 5, True, True),        # k5 pair layout (conv3d_fwd_k5_persist<1, true>): W >= 32, C_out 8
    (2, 32, 7, 6, 5, 40, 5, True, True),       # k5 pair layout, ragged boxes, 4 chunks, 7 real channels
    (1, 16, 16, 32, 32, 64, 5, True, False),   # k5 persistent kernel, one N tile (>= 256 boxes), 2 chunks
    (1, 8, 32, 32, 32, 64, 5, False, True),    # k5 persistent kernel, two N tiles per block
    (1, 16, 24, 30, 34, 70, 5, True, False),   # k5 persistent kernel, ragged boxes, half-empty second N tile
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_fwd_and_stats(case):
    ops = _ops()
    n, ci, co, d, h, w, k, xf, bias = case
    x = torch.randn(n, ci, d, h, w, generator=g(1))
    wt = torch.randn(co, ci, k, k, k, generator=g(2)) * 0.2
    b = torch.randn(co, generator=g(3)) if bias else None
    cip, cop = ops.pad8(ci), ops.pad8(co)
    xa = x
    xc = to_cl(x)
    if xf:
        sc, sh = xf_vectors(ci, cip, 5)
        xa = F.relu(x * sc[:ci].view(1, -1, 1, 1, 1) + sh[:ci].view(1, -1, 1, 1, 1))
        xc = xc.with_xf(sc.cuda(), sh.cuda(), True)
    ref = F.conv3d(xa, wt, b, 1, (k - 1) // 2)
    lay = ops.conv_layout(k, cop, w)
    wp = ops.pack_conv_w(wt.cuda(), None, cip, cop, 0, lay)
    bp = b.cuda() if bias else None          # logical, unpadded bias
    # write into a channel slice of a wider buffer to exercise strides
    obuf = torch.full((n, d, h, w, cop + 8), 7.0, device="cuda")
    out = ops.CL(obuf, 8, cop)
    nb = ops.conv_num_blocks((n, d, h, w), cop, lay, k)
    stats = torch.full((nb, 2, cop), float("nan"), device="cuda")      # every row must be written
    ops.conv3d_fwd(xc, wp, bp, out, k, stats, None, lay)
    torch.cuda.synchronize()
    got = from_cl(out, co)
    assert rel_err(got, ref) < 1e-4
    assert torch.all(obuf[..., :8] == 7.0)                      # neighbouring slice untouched
    if cop > co:
        assert torch.all(out.buf[..., out.c0 + co:out.c0 + cop] == 0)   # padded channels are zero
    s = stats.sum(0).cpu().double()
    ref_s1 = ref.double().sum((0, 2, 3, 4)); ref_s2 = (ref.double() ** 2).sum((0, 2, 3, 4))
    assert torch.allclose(s[0, :co], ref_s1, rtol=1e-4, atol=1e-3 * ref_s2.max().sqrt().item())
    assert torch.allclose(s[1, :co], ref_s2, rtol=1e-4)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_backward(case):
    """dgrad through the mode-1 packing of the same forward kernel, wgrad kernel, bias grad."""
    ops = _ops()
    n, ci, co, d, h, w, k, xf, bias = case
    x = torch.randn(n, ci, d, h, w, generator=g(11))
    wt = (torch.randn(co, ci, k, k, k, generator=g(12)) * 0.2).requires_grad_(True)
    gy = torch.randn(n, co, d, h, w, generator=g(13))
    cip, cop = ops.pad8(ci), ops.pad8(co)
    xc = to_cl(x)
    xa = x.clone()
    if xf:
        sc, sh = xf_vectors(ci, cip, 15)
        xa = F.relu(x * sc[:ci].view(1, -1, 1, 1, 1) + sh[:ci].view(1, -1, 1, 1, 1))
        xc = xc.with_xf(sc.cuda(), sh.cuda(), True)
    xa = xa.detach().requires_grad_(True)
    b = torch.zeros(co, requires_grad=True)
    y = F.conv3d(xa, wt, b, 1, (k - 1) // 2)
    y.backward(gy)
    gc = to_cl(gy)
    # data gradient
    lay = ops.conv_layout(k, cip, w)
    wpd = ops.pack_conv_w(wt.detach().cuda(), None, cop, cip, 1, lay)
    gin = ops.CL(torch.empty(n, d, h, w, cip, device="cuda"), 0, cip)
    ops.conv3d_fwd(gc, wpd, None, gin, k, None, None, lay)
    # weight gradient
    ws = torch.empty(ops.conv3d_wgrad_ws((n, d, h, w), k, cip, cop), device="cuda")
    dw, db = ops.conv3d_wgrad(xc, gc, co, ci, k, None, ws, True)
    torch.cuda.synchronize()
    assert rel_err(from_cl(gin, ci), xa.grad) < 1e-4
    assert rel_err(dw.cpu(), wt.grad) < 1e-4
    assert rel_err(db.cpu(), b.grad) < 1e-4


@pytest.mark.parametrize("case", [(1, 1, 8, 8, 8, 32), (2, 2, 7, 6, 5, 40), (1, 1, 3, 4, 4, 16), (1, 2, 4, 9, 7, 70)])
def test_first_layer_direct_kernels(case):
    """C_in <= 2 first conv: direct forward (+BN partials), input gradient and weight gradient, NCDHW in place."""
    ops = _ops()
    n, ci, co, d, h, w = case
    assert ops.conv_first_supported(3, ci, 8, w)
    x = torch.randn(n, ci, d, h, w, generator=g(1)).requires_grad_(True)
    wt = (torch.randn(co, ci, 3, 3, 3, generator=g(2)) * 0.3).requires_grad_(True)
    ref = F.conv3d(x, wt, None, 1, 1)
    gy = torch.randn(ref.shape, generator=g(3))
    ref.backward(gy)
    obuf = torch.full((n, d, h, w, 16), 7.0, device="cuda")
    out = ops.CL(obuf, 8, 8)
    nb = ops.conv_first_num_blocks((n, d, h, w))
    stats = torch.full((nb, 2, 8), float("nan"), device="cuda")
    ops.conv_first_fwd(x.detach().cuda(), wt.detach().cuda(), None, out, stats)
    assert rel_err(from_cl(out, co), ref.detach()) < 1e-5
    assert torch.all(obuf[..., :8] == 7.0) and torch.all(obuf[..., 8 + co:] == 0)
    s = stats.sum(0).cpu().double()
    assert torch.allclose(s[0, :co], ref.detach().double().sum((0, 2, 3, 4)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(s[1, :co], (ref.detach().double() ** 2).sum((0, 2, 3, 4)), rtol=1e-4)
    gc = to_cl(gy)
    dx = ops.conv_first_bwd_data(gc, wt.detach().cuda(), ci)
    assert rel_err(dx.cpu(), x.grad) < 1e-5
    ws = torch.empty(ops.conv_first_wgrad_ws((n, d, h, w), ci), device="cuda")
    dw = ops.conv_first_wgrad(x.detach().cuda(), gc, co, ws)
    assert rel_err(dw.cpu(), wt.grad) < 1e-4


def test_conv3d_imap_concat():
    """Input channels scattered in a padded concat buffer (UNetSP: 7+pad | 7+pad)."""
    ops = _ops()
    n, d, h, w, k = 1, 8, 8, 8, 3
    xa, xb = torch.randn(n, 7, d, h, w, generator=g(1)), torch.randn(n, 7, d, h, w, generator=g(2))
    wt = torch.randn(5, 14, k, k, k, generator=g(3)) * 0.2
    ref = F.conv3d(torch.cat((xa, xb), 1), wt, None, 1, 1)
    buf = torch.zeros(n, d, h, w, 16)
    buf[..., 0:7] = xa.permute(0, 2, 3, 4, 1); buf[..., 8:15] = xb.permute(0, 2, 3, 4, 1)
    xc = ops.CL(buf.cuda(), 0, 16)
    pos = list(range(7)) + list(range(8, 15))                    # logical channel -> padded position
    inv = [-1] * 16
    for logical, p_ in enumerate(pos):
        inv[p_] = logical
    cinv = torch.tensor(inv, dtype=torch.int32).cuda()           # padded position -> logical channel
    wp = ops.pack_conv_w(wt.cuda(), cinv, 16, 8, 0)
    out = ops.CL(torch.empty(n, d, h, w, 8, device="cuda"), 0, 8)
    ops.conv3d_fwd(xc, wp, None, out, k, None)
    assert rel_err(from_cl(out, 5), ref) < 1e-4
    gy = torch.randn(n, 5, d, h, w, generator=g(4))
    ws = torch.empty(ops.conv3d_wgrad_ws((n, d, h, w), k, 16, 8), device="cuda")
    dw, _ = ops.conv3d_wgrad(xc, to_cl(gy), 5, 14, k, cinv, ws, False)
    xcat = torch.cat((xa, xb), 1).requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    F.conv3d(xcat, wr, None, 1, 1).backward(gy)
    assert rel_err(dw.cpu(), wr.grad) < 1e-4
    wpd = ops.pack_conv_w(wt.cuda(), cinv, 8, 16, 1)
    gin = ops.CL(torch.empty(n, d, h, w, 16, device="cuda"), 0, 16)
    ops.conv3d_fwd(to_cl(gy), wpd, None, gin, k, None)
    got = gin.buf.cpu()
    assert rel_err(got[..., 0:7].permute(0, 4, 1, 2, 3), xcat.grad[:, :7]) < 1e-4
    assert rel_err(got[..., 8:15].permute(0, 4, 1, 2, 3), xcat.grad[:, 7:]) < 1e-4
    assert torch.all(got[..., 7] == 0) and torch.all(got[..., 15] == 0)


@pytest.mark.parametrize("shape", [(2, 8, 8, 8, 8), (1, 7, 4, 6, 10), (1, 64, 4, 4, 4), (1, 112, 2, 2, 2)])
def test_batchnorm_train_fwd_bwd(shape):
    """conv stats -> finalize -> lazy transform; then BN+ReLU backward, vs F.batch_norm/relu autograd."""
    ops = _ops()
    n, c, d, h, w = shape
    cp = ops.pad8(c)
    y = (torch.randn(shape, generator=g(1)) * 1.7 + 0.4)
    gamma = torch.rand(c, generator=g(2)) * 1.5 - 0.25
    beta = torch.randn(c, generator=g(3)) * 0.2
    rm, rv = torch.randn(c, generator=g(4)) * 0.1, torch.rand(c, generator=g(5)) + 0.5
    # reference
    yr = y.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    a = F.relu(F.batch_norm(yr, rm_r, rv_r, gr, br, True, 0.1, 1e-5))
    ga = torch.randn(shape, generator=g(6))
    a.backward(ga)
    # ours: identity 1x1-free path -- build stats with the channel-sum style partial layout via a conv of k=3 delta kernel
    yc = to_cl(y)
    wt = torch.zeros(c, c, 3, 3, 3); wt[range(c), range(c), 1, 1, 1] = 1.0
    wp = ops.pack_conv_w(wt.cuda(), None, cp, cp, 0)
    out = ops.CL(torch.empty(n, d, h, w, cp, device="cuda"), 0, cp)
    nb = ops.conv_num_blocks((n, d, h, w), cp, 0, 3)
    stats = torch.zeros(nb, 2, cp, device="cuda")
    ops.conv3d_fwd(yc, wp, None, out, 3, stats)
    rm_g, rv_g = rm.cuda(), rv.cuda()
    vec = ops.bn_finalize(stats, nb, c, cp, n * d * h * w, gamma.cuda(), beta.cuda(), rm_g, rv_g, 0.1, 1e-5, 1)
    torch.cuda.synchronize()
    assert torch.allclose(rm_g.cpu(), rm_r, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv_g.cpu(), rv_r, rtol=1e-5, atol=1e-6)
    act = from_cl(out, c) * vec[0, :c].cpu().view(1, -1, 1, 1, 1) + vec[1, :c].cpu().view(1, -1, 1, 1, 1)
    assert rel_err(F.relu(act), a.detach()) < 1e-4
    assert torch.all(vec[:, c:] == 0)
    gac = to_cl(ga)
    part = torch.empty(ops.bn_bwd_partials_floats(n * d * h * w, cp), device="cuda")
    dgam, dbet = ops.bn_relu_bwd(out, gac, vec, gamma.cuda(), c, part)
    torch.cuda.synchronize()
    assert rel_err(dgam.cpu(), gr.grad) < 1e-4
    assert rel_err(dbet.cpu(), br.grad) < 1e-4
    assert rel_err(from_cl(gac, c), yr.grad) < 2e-4


def test_batchnorm_double_update_and_eval():
    ops = _ops()
    c, cp = 5, 8
    stats = torch.zeros(3, 2, cp); stats[:, 0, :c] = torch.rand(3, c, generator=g(1)) * 10
    stats[:, 1, :c] = torch.rand(3, c, generator=g(2)) * 50 + 40
    count = 100.0
    mean = stats[:, 0, :c].sum(0) / count; var = stats[:, 1, :c].sum(0) / count - mean ** 2
    gamma, beta = torch.rand(c) + 0.5, torch.randn(c)
    rm, rv = torch.zeros(c), torch.ones(c)
    exp_rm, exp_rv = rm.clone(), rv.clone()
    for _ in range(2):
        exp_rm = 0.9 * exp_rm + 0.1 * mean
        exp_rv = 0.9 * exp_rv + 0.1 * var * count / (count - 1)
    rm_g, rv_g = rm.cuda(), rv.cuda()
    ops.bn_finalize(stats.cuda(), 3, c, cp, count, gamma.cuda(), beta.cuda(), rm_g, rv_g, 0.1, 1e-5, 2)
    assert torch.allclose(rm_g.cpu(), exp_rm, rtol=1e-5) and torch.allclose(rv_g.cpu(), exp_rv, rtol=1e-5)
    vec = ops.bn_eval_affine(gamma.cuda(), beta.cuda(), rm_g, rv_g, 1e-5, c, cp).cpu()
    sc = gamma / torch.sqrt(exp_rv + 1e-5)
    assert torch.allclose(vec[0, :c], sc, rtol=1e-5) and torch.allclose(vec[1, :c], beta - exp_rm * sc, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape,xf", [((1, 8, 8, 8, 8), False), ((2, 7, 4, 6, 10), True), ((1, 64, 2, 2, 2), True)])
def test_maxpool_fwd_bwd(shape, xf):
    ops = _ops()
    n, c, d, h, w = shape
    cp = ops.pad8(c)
    x = torch.randn(shape, generator=g(1))
    xc = to_cl(x)
    xa = x
    if xf:
        sc, sh = xf_vectors(c, cp, 3)
        xa = F.relu(x * sc[:c].view(1, -1, 1, 1, 1) + sh[:c].view(1, -1, 1, 1, 1))
        xc = xc.with_xf(sc.cuda(), sh.cuda(), True)
    xa = xa.detach().requires_grad_(True)
    ref = F.max_pool3d(xa, 2, 2)
    out = ops.CL(torch.empty(n, d // 2, h // 2, w // 2, cp, device="cuda"), 0, cp)
    ops.maxpool_fwd(xc, out)
    # identity transform: bit exact; with the transform the kernel uses one fused fma (1 ulp vs mul+add)
    assert torch.equal(from_cl(out, c), ref.detach()) if not xf else torch.allclose(from_cl(out, c), ref.detach(), rtol=1e-6, atol=1e-7)
    gy = torch.randn(ref.shape, generator=g(2))
    ref.backward(gy)
    base = torch.randn(shape, generator=g(4))
    gin = to_cl(base)
    ops.maxpool_bwd(xc, to_cl(gy), gin, True)
    got = from_cl(gin, c)
    if xf:
        # ties at 0 after ReLU route to the first element; ReLU's own backward zeroes them, so compare where a > 0
        mask = (xa.detach() > 0)
        assert torch.allclose((got - base)[mask], xa.grad[mask])
    else:
        assert torch.allclose(got - base, xa.grad)


@pytest.mark.parametrize("shape", [(1, 8, 8, 8, 8), (2, 7, 4, 6, 10), (1, 64, 4, 4, 4), (1, 16, 16, 16, 32)])
def test_maxpool_bwd_with_bn_reduction(shape):
    """maxpool_bwd(bn=...) + bn_relu_bwd(pre_reduced=...) vs autograd of skip + max_pool3d(relu(batch_norm(y))):
    the pool backward adds its share to the skip's gradient and emits the BatchNorm-backward reduction rows itself."""
    ops = _ops()
    n, c, d, h, w = shape
    cp = ops.pad8(c)
    y = (torch.randn(shape, generator=g(1)) * 1.3 + 0.2)
    gamma = torch.rand(c, generator=g(2)) + 0.25
    beta = torch.randn(c, generator=g(3)) * 0.2
    yr = y.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    a = F.relu(F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5))
    gskip = torch.randn(shape, generator=g(4))                  # the concat consumer's gradient w.r.t. a
    gpool = torch.randn(n, c, d // 2, h // 2, w // 2, generator=g(5))
    ((a * gskip).sum() + (F.max_pool3d(a, 2, 2) * gpool).sum()).backward()
    # ours: batch statistics through the channel-sum partials of an identity conv, as in test_batchnorm_train_fwd_bwd
    wt = torch.zeros(c, c, 3, 3, 3); wt[range(c), range(c), 1, 1, 1] = 1.0
    wp = ops.pack_conv_w(wt.cuda(), None, cp, cp, 0)
    out = ops.CL(torch.empty(n, d, h, w, cp, device="cuda"), 0, cp)
    nb = ops.conv_num_blocks((n, d, h, w), cp, 0, 3)
    stats = torch.zeros(nb, 2, cp, device="cuda")
    ops.conv3d_fwd(to_cl(y), wp, None, out, 3, stats)
    vec = ops.bn_finalize(stats, nb, c, cp, n * d * h * w, gamma.cuda(), beta.cuda(), torch.zeros(c).cuda(), torch.ones(c).cuda(),
                          0.1, 1e-5, 1)
    act = out.with_xf(vec[0], vec[1], True)
    res = {}
    for fused in (False, True):
        ga = to_cl(gskip)
        part = torch.empty(max(ops.bn_bwd_partials_floats(n * d * h * w, cp), ops.maxpool_bwd_bn_blocks((n, d, h, w), cp) * 2 * cp),
                           device="cuda")
        rows = ops.maxpool_bwd(act, to_cl(gpool), ga, True, (vec, part) if fused else None)
        assert (rows is not None) == fused
        dgam, dbet = ops.bn_relu_bwd(out, ga, vec, gamma.cuda(), c, part, None, rows)
        torch.cuda.synchronize()
        res[fused] = (dgam.cpu(), dbet.cpu(), from_cl(ga, c))
        assert rel_err(dgam.cpu(), gr.grad) < 1e-4
        assert rel_err(dbet.cpu(), br.grad) < 1e-4
        assert rel_err(from_cl(ga, c), yr.grad) < 2e-4
    assert rel_err(res[True][2], res[False][2]) < 1e-5


@pytest.mark.parametrize("case", [(1, 8, 8, 4, 4, 4, False), (2, 32, 32, 4, 6, 10, True), (1, 14, 14, 4, 4, 4, True),
                                  (1, 128, 128, 2, 2, 2, True), (1, 6, 6, 1, 1, 1, False)])
def test_convtranspose(case):
    ops = _ops()
    n, ci, co, d, h, w, xf = case
    cip, cop = ops.pad8(ci), ops.pad8(co)
    x = torch.randn(n, ci, d, h, w, generator=g(1))
    wt = (torch.randn(ci, co, 2, 2, 2, generator=g(2)) * 0.3).requires_grad_(True)
    b = torch.randn(co, generator=g(3)).requires_grad_(True)
    xc = to_cl(x)
    xa = x
    if xf:
        sc, sh = xf_vectors(ci, cip, 4)
        xa = F.relu(x * sc[:ci].view(1, -1, 1, 1, 1) + sh[:ci].view(1, -1, 1, 1, 1))
        xc = xc.with_xf(sc.cuda(), sh.cuda(), True)
    xa = xa.detach().requires_grad_(True)
    ref = F.conv_transpose3d(xa, wt, b, stride=2)
    gy = torch.randn(ref.shape, generator=g(5))
    ref.backward(gy)
    wp = ops.pack_convt_w(wt.detach().cuda(), None, cip, cop, 0)
    bp = b.detach().cuda()
    out = ops.CL(torch.empty(n, 2 * d, 2 * h, 2 * w, cop, device="cuda"), 0, cop)
    ops.convt_fwd(xc, wp, bp, out)
    assert rel_err(from_cl(out, co), ref.detach()) < 1e-4
    gc = to_cl(gy)
    wpd = ops.pack_convt_w(wt.detach().cuda(), None, cop, cip, 1)
    gin = ops.CL(torch.empty(n, d, h, w, cip, device="cuda"), 0, cip)
    ops.convt_bwd_data(gc, wpd, gin)
    assert rel_err(from_cl(gin, ci), xa.grad) < 1e-4
    ws = torch.empty(ops.convt_wgrad_ws((n, d, h, w), cip, cop), device="cuda")
    dw, db = ops.convt_wgrad(xc, gc, ci, co, None, ws)
    assert rel_err(dw.cpu(), wt.grad) < 1e-4
    assert rel_err(db.cpu(), b.grad) < 1e-4


@pytest.mark.parametrize("co,act,mode", [(2, 2, 0), (2, 1, 0), (3, 2, 1), (3, 2, 2), (3, 3, 0), (1, 0, 0)])
def test_head_fwd_bwd(co, act, mode):
    ops = _ops()
    n, d, h, w = 2, 4, 6, 10
    xa_, xb_ = torch.randn(n, 7, d, h, w, generator=g(1)), torch.randn(n, 7, d, h, w, generator=g(2))
    buf = torch.zeros(n, d, h, w, 16)
    buf[..., 0:7] = xa_.permute(0, 2, 3, 4, 1); buf[..., 8:15] = xb_.permute(0, 2, 3, 4, 1)
    sc, sh = torch.rand(16, generator=g(3)) + 0.2, torch.randn(16, generator=g(4)) * 0.3
    xc = ops.CL(buf.cuda(), 0, 16, sc.cuda(), sh.cuda(), True)
    imap_l = list(range(7)) + list(range(8, 15))
    imap = torch.tensor(imap_l, dtype=torch.int32).cuda()
    xcat = torch.cat((xa_, xb_), 1)
    a = F.relu(xcat * sc[imap_l].view(1, -1, 1, 1, 1) + sh[imap_l].view(1, -1, 1, 1, 1)).requires_grad_(True)
    wt = (torch.randn(co, 14, generator=g(5)) * 0.4).requires_grad_(True)
    b = torch.randn(co, generator=g(6)).requires_grad_(True)
    lc = F.conv3d(a, wt.view(co, 14, 1, 1, 1), b)
    y = F.softmax(lc, 1) if act & 1 else lc
    y = torch.sigmoid(y) if act & 2 else y
    if mode == 0:
        refs = (y,)
    else:
        sk = torch.cat((y[:, 0:1], y[:, 1:2] + y[:, 2:3]), 1); fl = torch.cat((1 - y[:, 1:2], y[:, 1:2]), 1)
        refs = (F.softmax(sk, 1), F.softmax(fl, 1)) if mode == 2 else (sk, fl)
    gs = [torch.randn(r.shape, generator=g(7 + i)) for i, r in enumerate(refs)]
    torch.autograd.backward(refs, gs)
    o0, o1 = ops.head_fwd(xc, wt.detach().cuda(), b.detach().cuda(), imap, act, mode)
    assert rel_err(o0.cpu(), refs[0].detach()) < 1e-5
    if mode:
        assert rel_err(o1.cpu(), refs[1].detach()) < 1e-5
    gin = ops.CL(torch.empty(n, d, h, w, 16, device="cuda"), 0, 16)
    dw, db = ops.head_bwd(xc, wt.detach().cuda(), b.detach().cuda(), imap, act, mode, gs[0].cuda(),
                          gs[1].cuda() if mode else None, gin)
    got = gin.buf.cpu()[..., imap_l].permute(0, 4, 1, 2, 3)
    assert rel_err(got, a.grad) < 1e-4
    assert rel_err(dw.cpu(), wt.grad) < 1e-4
    assert rel_err(db.cpu(), b.grad) < 1e-4


@pytest.mark.parametrize("cp,bn_cp,co,act,mode", [(16, 8, 2, 1, 0), (8, 8, 2, 0, 0), (32, 16, 3, 2, 1)])
def test_head_bwd_with_bn_reduction(cp, bn_cp, co, act, mode):
    """head_bwd(bn=...) emits the BatchNorm-backward reduction rows of the layer behind its first bn_cp input channels:
    the rows summed must equal those of bn_relu_bwd_reduce on the gradient the head wrote, and the full BatchNorm backward
    (dgamma, dbeta, raw-output gradient) must agree between the two routes."""
    ops = _ops()
    n, d, h, w = 2, 4, 6, 10
    nvox = n * d * h * w
    buf = (torch.randn(n, d, h, w, cp, generator=g(1)) * 1.2 + 0.1).cuda()
    vec = torch.zeros(4, cp)
    vec[0] = torch.rand(cp, generator=g(2)) + 0.3                   # scale = gamma * invstd
    vec[1] = torch.randn(cp, generator=g(3)) * 0.3                  # shift
    vec[2] = torch.randn(cp, generator=g(4)) * 0.2                  # mean
    vec[3] = torch.rand(cp, generator=g(5)) + 0.5                   # invstd
    vec = vec.cuda()
    gamma = (vec[0] / vec[3])[:bn_cp].contiguous()
    xc = ops.CL(buf, 0, cp, vec[0], vec[1], True)
    wt = (torch.randn(co, cp, generator=g(6)) * 0.4).cuda()
    b = torch.randn(co, generator=g(7)).cuda()
    o0, o1 = ops.head_fwd(xc, wt, b, None, act, mode)
    g0 = torch.randn(o0.shape, generator=g(8)).cuda()
    g1 = torch.randn(o1.shape, generator=g(9)).cuda() if mode else None
    bnvec = vec[:, :bn_cp]
    res = {}
    for fused in (False, True):
        gin = ops.CL(torch.empty(n, d, h, w, cp, device="cuda"), 0, cp)
        part = torch.empty(max(ops.bn_bwd_partials_floats(nvox, bn_cp), ops.head_bwd_blocks((n, d, h, w)) * 2 * bn_cp), device="cuda")
        out = ops.head_bwd(xc, wt, b, None, act, mode, g0, g1, gin, (bnvec, part) if fused else None)
        rows = out[2] if fused else None
        assert len(out) == (3 if fused else 2)
        ga = ops.CL(gin.buf, 0, bn_cp)
        dgam, dbet = ops.bn_relu_bwd(ops.CL(buf, 0, bn_cp), ga, bnvec, gamma, bn_cp, part, None, rows)
        torch.cuda.synchronize()
        res[fused] = (out[0].cpu(), out[1].cpu(), dgam.cpu(), dbet.cpu(), gin.buf.cpu())
    for a_, b_ in zip(res[True], res[False]):
        assert rel_err(a_, b_) < 1e-5


@pytest.mark.parametrize("ce,dice,sm", [(1.0, 1.0, False), (1.0, 1.0, True), (0.0, 1.0, True), (1.0, 0.0, False),
                                        (0.5, 2.0, True)])
def test_loss_fwd_bwd(ce, dice, sm):
    import sys
    from oracle import unet_oracle as O
    ops = _ops()
    p = torch.rand(2, 2, 6, 8, 10, generator=g(1)).requires_grad_(True)
    m = (torch.rand(2, 6, 8, 10, generator=g(2)) < 0.3).long()
    t = F.one_hot(m, 2).movedim(4, 1).float().contiguous()
    terms = []
    if ce:
        terms.append(ce * O.cross_entropy(p, t))
    if dice:
        terms.append(dice * O.dice_loss(F.softmax(p, 1) if sm else p, t))
    total = sum(terms)
    total.backward()
    tg, ws = ops.loss_fwd(p.detach().cuda(), t.cuda(), ce, dice, sm)
    assert abs(tg.sum().item() - total.item()) < 1e-5 * max(1.0, abs(total.item()))
    gp = ops.loss_bwd(p.detach().cuda(), t.cuda(), ce, dice, sm, ws, None, None)
    assert rel_err(gp.cpu(), p.grad) < 1e-4
    gp2 = ops.loss_bwd(p.detach().cuda(), t.cuda(), ce, dice, sm, ws, torch.tensor(2.0).cuda(), torch.tensor(2.0).cuda(), gp.clone(), True)
    assert rel_err(gp2.cpu(), 3 * p.grad) < 1e-4


def test_layout_roundtrip_and_errors():
    ops = _ops()
    from ctunet_amd import _lib
    x = torch.randn(2, 3, 4, 6, 10, generator=g(1))
    a = ops.ncdhw_to_cl(x.cuda())
    assert a.cp == 8 and torch.all(a.buf[..., 3:] == 0)
    assert torch.equal(ops.cl_to_ncdhw(a, 3).cpu(), x)
    with pytest.raises(RuntimeError):
        ops.ncdhw_to_cl(x)                       # CPU tensor: no fallback
    with pytest.raises(_lib.CtuError):
        ops.pack_conv_w(torch.zeros(4, 4, 7, 7, 7).cuda(), None, 8, 8, 0)     # k=7 unsupported -> loud error


@pytest.mark.parametrize("shape", [(1, 8, 4, 4, 4), (2, 12, 2, 6, 10)])
def test_skip_add(shape):
    """ctu_skip_add: out = act(a) + act(b) on channel SLICES of one wide buffer (UNet(cat=False), models.py:250-251)
    and its b = None form (strided slice copy).  fp32 reference: relu(a*sa+sha) + relu(b*sb+shb), 1-ulp fma slack."""
    ops = _ops()
    n, c, d, h, w = shape
    cp = ops.pad8(c)
    a, b = torch.randn(shape, generator=g(5)), torch.randn(shape, generator=g(6))
    sa, sha = xf_vectors(c, cp, 7)
    sb, shb = xf_vectors(c, cp, 8)
    wide = torch.zeros(n, d, h, w, 2 * cp, device="cuda")
    wide[..., :c] = a.permute(0, 2, 3, 4, 1).cuda()
    wide[..., cp:cp + c] = b.permute(0, 2, 3, 4, 1).cuda()
    ca = ops.CL(wide, 0, cp, sa.cuda(), sha.cuda(), True)
    cb = ops.CL(wide, cp, cp, sb.cuda(), shb.cuda(), True)
    out = ops.CL(torch.empty(n, d, h, w, cp, device="cuda"), 0, cp)
    ops.skip_add(ca, cb, out)
    v = lambda t: t[:c].view(1, -1, 1, 1, 1)
    ref = F.relu(a * v(sa) + v(sha)) + F.relu(b * v(sb) + v(shb))
    assert torch.allclose(from_cl(out, c), ref, rtol=1e-6, atol=1e-6)
    ops.skip_add(ops.CL(wide, 0, cp), None, ops.CL(wide, cp, cp))          # slice copy, bit exact
    assert torch.equal(wide[..., cp:], wide[..., :cp])


# ---------------------------------------------------------------------------- inference tail / sample schema (8 f4, f1)
@pytest.mark.parametrize("shape", [(2, 2, 4, 6, 10), (1, 3, 8, 8, 8), (3, 4, 5, 3, 7)])
def test_hard_segm_one_hot_dice_counts_bit_exact(shape):
    """Index / counting work: bit exact against torch.argmax / one_hot / integer counts, including exact ties (the
    first maximum wins) -- utilities.py:103-124, datasets.py:107-110; hard Dice = the oracle's restatement of
    monai's compute_meandice (parity unpinned)."""
    from oracle import unet_oracle as O
    ops = _ops()
    n, c, d, h, w = shape
    p = torch.rand(shape, generator=g(11))
    p[:, :, 0, 0, :] = 0.25                                   # ties across all classes
    p[:, 1:, 1, 0, :] = p[:, :1, 1, 0, :]                     # tie with class 0
    seg = ops.hard_segm(p.cuda())
    assert seg.dtype == torch.float32 and torch.equal(seg.cpu(), p.argmax(1).float())
    assert torch.equal(ops.hard_segm(p[0].contiguous().cuda()).cpu(), p[0].argmax(0).float())
    lab = torch.randint(0, c, (n, d, h, w), generator=g(12)).float()
    oh = ops.one_hot(lab.cuda(), c)
    assert torch.equal(oh.cpu(), F.one_hot(lab.long(), c).movedim(-1, 1).float())
    cnt = ops.hard_dice_counts(p.cuda(), oh).cpu()
    hard = F.one_hot(p.argmax(1), c).movedim(-1, 1).double()
    ref = torch.stack([(hard * oh.cpu().double()).flatten(2).sum(2), hard.flatten(2).sum(2), oh.cpu().double().flatten(2).sum(2)], -1)
    assert torch.equal(cnt, ref)
    from ctunet_amd.utilities import dice_coeff, hard_segm_from_tensor
    assert hard_segm_from_tensor(p.cuda(), keep_dims=True).shape == (n, 1, d, h, w)
    assert abs(dice_coeff(p.cuda(), oh).item() - O.hard_dice(p, oh.cpu())) < 1e-6
    assert abs(dice_coeff(p.cuda(), hard.float().cuda()).item() - 1.0) < 1e-7


def test_hard_dice_counts_full_size():
    """128^3: counts are exact integers and add up (size-independent properties: sum_c |hard_c| = V = sum_c |target_c|)."""
    ops = _ops()
    p = torch.rand(1, 2, 128, 128, 128, generator=g(13)).cuda()
    t = ops.one_hot((torch.rand(1, 128, 128, 128, generator=g(14)) < 0.3).float().cuda(), 2)
    cnt = ops.hard_dice_counts(p, t)
    assert torch.equal(cnt, cnt.round()) and cnt[0, :, 1].sum().item() == 128 ** 3 and cnt[0, :, 2].sum().item() == 128 ** 3
    assert cnt[0, 1, 0].item() == float(((p[0, 1] > p[0, 0]) & (t[0, 1] > 0.5)).sum().item())


# ---------------------------------------------------------------------------- fused up-convolution
@pytest.mark.parametrize("c,co,dims,segs", [
    (8, 8, (1, 4, 4, 16), None),                      # single chunk, one box, every voxel on a face
    (16, 8, (2, 8, 4, 32), None),                     # batch, two boxes along w
    (24, 16, (1, 4, 8, 16), ((12, 0), (12, 16))),     # concat input layout (two slices of a 32-wide buffer)
    (32, 8, (1, 6, 5, 20), None),                     # ragged: boxes hang over the volume on every axis
    (16, 32, (1, 4, 4, 16), None),                    # 2 channel tiles, 2 parity groups
    (8, 64, (1, 4, 4, 16), None),                     # 4 channel tiles, 4 parity groups
])
def test_upconv_fused_fwd(c, co, dims, segs):
    """ConvTranspose3d(C,C,2,2,bias) -> Conv3d(C,Co,3,p=1) (models.py:37-38) as one coarse-grid kernel vs the two torch
    ops in fp32 (reference = F.conv3d(F.conv_transpose3d(relu(bn(x))))), incl. the border-class bias at every face
    and the BatchNorm partial sums.  Tolerance 1e-4 of the output scale: the composite weights re-associate the sums."""
    ops = _ops()
    n, d, h, w = dims
    cop = ops.pad8(co)
    x = torch.randn(n, c, d, h, w, generator=g(21))
    wt = torch.randn(c, c, 2, 2, 2, generator=g(22)) * 0.2
    bt = torch.randn(c, generator=g(23))
    w3 = torch.randn(co, c, 3, 3, 3, generator=g(24)) * 0.1
    if segs is None:
        cp = ops.pad8(c)
        xc = to_cl(x)
        cinv = None
        sc, sh = xf_vectors(c, cp, 25)
        scl, shl = sc[:c], sh[:c]
    else:                                             # logical channels live at padded positions (free-concat layout)
        cp = 32
        buf = torch.zeros(n, d, h, w, cp)
        cinv_l = [-1] * cp
        lo = 0
        for cnt, start in segs:
            buf[..., start:start + cnt] = x[:, lo:lo + cnt].permute(0, 2, 3, 4, 1)
            for q in range(cnt):
                cinv_l[start + q] = lo + q
            lo += cnt
        xc = ops.CL(buf.cuda(), 0, cp)
        cinv = torch.tensor(cinv_l, dtype=torch.int32, device="cuda")
        sc, sh = xf_vectors(cp, cp, 25)
        pos = [i for i, v in enumerate(cinv_l) if v >= 0]
        scl, shl = sc[pos], sh[pos]
    xa = F.relu(x * scl.view(1, -1, 1, 1, 1) + shl.view(1, -1, 1, 1, 1))
    ref = F.conv3d(F.conv_transpose3d(xa, wt, bt, stride=2), w3, padding=1)
    wp, beff, _ = ops.upconv_fused_pack(wt.cuda(), bt.cuda(), w3.cuda(), cinv, cp, cop)
    out = ops.CL(torch.full((n, 2 * d, 2 * h, 2 * w, cop), float("nan"), device="cuda"), 0, cop)
    nb = ops.upconv_fused_num_blocks(dims, cop)
    stats = torch.zeros(nb, 2, cop, device="cuda")
    ops.upconv_fused_fwd(xc.with_xf(sc.cuda(), sh.cuda(), True), wp, beff, out, stats)
    got = from_cl(out, co)
    assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    if cop > co:
        assert torch.equal(out.buf[..., co:].cpu(), torch.zeros(n, 2 * d, 2 * h, 2 * w, cop - co))
    s = stats.sum(0).cpu()
    assert torch.allclose(s[0, :co], ref.sum((0, 2, 3, 4)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(s[1, :co], (ref * ref).sum((0, 2, 3, 4)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("c,co,dims,segs", [
    (8, 8, (1, 4, 4, 16), None),
    (16, 8, (2, 8, 4, 32), None),
    (24, 16, (1, 4, 8, 16), ((12, 0), (12, 16))),
    (32, 8, (1, 6, 5, 20), None),
    (16, 32, (1, 4, 4, 16), None),
])
def test_upconv_fused_wgrad(c, co, dims, segs):
    """dWT, dbT, dW3 of ConvTranspose3d -> Conv3d through the composite-weight gradient and its projection, against
    torch autograd of the two unfused ops in fp64 (tolerance 2e-4 of each tensor's scale; fp32 sums over up to 2e4
    voxels).  Border planes matter for dbT / dW3 (zero padding of the transposed-conv output)."""
    ops = _ops()
    n, d, h, w = dims
    cop = ops.pad8(co)
    x = torch.randn(n, c, d, h, w, generator=g(31))
    wt = (torch.randn(c, c, 2, 2, 2, generator=g(32)) * 0.2)
    bt = torch.randn(c, generator=g(33))
    w3 = (torch.randn(co, c, 3, 3, 3, generator=g(34)) * 0.1)
    dy = torch.randn(n, co, 2 * d, 2 * h, 2 * w, generator=g(35))
    if segs is None:
        cp = ops.pad8(c)
        xc = to_cl(x)
        cinv = imap = None
        sc, sh = xf_vectors(c, cp, 36)
        scl, shl = sc[:c], sh[:c]
    else:
        cp = 32
        buf = torch.zeros(n, d, h, w, cp)
        cinv_l, imap_l, lo = [-1] * cp, [], 0
        for cnt, start in segs:
            buf[..., start:start + cnt] = x[:, lo:lo + cnt].permute(0, 2, 3, 4, 1)
            for q in range(cnt):
                cinv_l[start + q] = lo + q
                imap_l.append(start + q)
            lo += cnt
        xc = ops.CL(buf.cuda(), 0, cp)
        cinv = torch.tensor(cinv_l, dtype=torch.int32, device="cuda")
        imap = torch.tensor(imap_l, dtype=torch.int32, device="cuda")
        sc, sh = xf_vectors(cp, cp, 36)
        scl, shl = sc[imap_l], sh[imap_l]
    xa = F.relu(x * scl.view(1, -1, 1, 1, 1) + shl.view(1, -1, 1, 1, 1)).double()
    wt64, bt64, w364 = (t.double().requires_grad_(True) for t in (wt, bt, w3))
    ref = F.conv3d(F.conv_transpose3d(xa, wt64, bt64, stride=2), w364, padding=1)
    ref.backward(dy.double())
    _, _, pws = ops.upconv_fused_pack(wt.cuda(), bt.cuda(), w3.cuda(), cinv, cp, cop)
    gcl = to_cl(dy)
    dwt, dbt, dw3 = ops.upconv_fused_wgrad(xc.with_xf(sc.cuda(), sh.cuda(), True), gcl, c, co, bt.cuda(), pws, imap)
    for name, got, want in (("dWT", dwt, wt64.grad), ("dbT", dbt, bt64.grad), ("dW3", dw3, w364.grad)):
        err = (got.cpu().double() - want).abs().max().item()
        assert err <= 2e-4 * want.abs().max().item(), (name, err, want.abs().max().item())


@pytest.mark.parametrize("c,co,dims,segs", [
    (8, 8, (1, 4, 4, 16), None),
    (16, 8, (2, 8, 4, 32), None),
    (24, 16, (1, 4, 8, 16), ((12, 0), (12, 16))),
    (32, 8, (1, 6, 5, 20), None),
    (64, 16, (1, 4, 4, 16), None),
    (16, 32, (1, 4, 4, 16), None),
])
def test_upconv_fused_bwd_data(c, co, dims, segs):
    """Gradient of ConvTranspose3d -> Conv3d w.r.t. the transposed conv's input as one coarse-grid kernel (the adjoint
    parity convolutions over a strided view of the fine-grid gradient) vs torch autograd of the two ops in fp64;
    2e-4 of the gradient's scale."""
    ops = _ops()
    n, d, h, w = dims
    cop = ops.pad8(co)
    wt = (torch.randn(c, c, 2, 2, 2, generator=g(42)) * 0.2)
    bt = torch.randn(c, generator=g(43))
    w3 = (torch.randn(co, c, 3, 3, 3, generator=g(44)) * 0.1)
    dy = torch.randn(n, co, 2 * d, 2 * h, 2 * w, generator=g(45))
    xa = torch.randn(n, c, d, h, w, generator=g(41)).double().requires_grad_(True)
    F.conv3d(F.conv_transpose3d(xa, wt.double(), bt.double(), stride=2), w3.double(), padding=1).backward(dy.double())
    if segs is None:
        cp, cinv, pos = ops.pad8(c), None, list(range(c))
    else:
        cp, cinv_l, pos, lo = 32, [-1] * 32, [], 0
        for cnt, start in segs:
            for q in range(cnt):
                cinv_l[start + q] = lo + q
                pos.append(start + q)
            lo += cnt
        cinv = torch.tensor(cinv_l, dtype=torch.int32, device="cuda")
    wp, _, _ = ops.upconv_fused_pack(wt.cuda(), bt.cuda(), w3.cuda(), cinv, cp, cop)
    wpd = ops.upconv_fused_pack_bwd(wp, cp, cop)
    gin = ops.CL(torch.full((n, d, h, w, cp), float("nan"), device="cuda"), 0, cp)
    ops.upconv_fused_bwd_data(to_cl(dy), wpd, gin)
    got = gin.buf.cpu()[..., pos].permute(0, 4, 1, 2, 3).double()
    err = (got - xa.grad).abs().max().item()
    assert err <= 2e-4 * xa.grad.abs().max().item(), (err, xa.grad.abs().max().item())
    rest = [q for q in range(cp) if q not in pos]
    if rest:
        assert torch.equal(gin.buf.cpu()[..., rest], torch.zeros(n, d, h, w, len(rest)))      # padding positions get zeros


def _bn_setup(ops, y, c, cp, seed):
    """BatchNorm vectors [4, cp] (scale, shift, mean, invstd) of a raw tensor y [N,C,D,H,W] with random gamma / beta."""
    gamma = torch.rand(c, generator=g(seed)) * 1.5 - 0.25
    beta = torch.randn(c, generator=g(seed + 1)) * 0.2
    mean = y.double().mean(dim=(0, 2, 3, 4))
    var = y.double().var(dim=(0, 2, 3, 4), unbiased=False)
    invstd = (1.0 / torch.sqrt(var + 1e-5)).float()
    vec = torch.zeros(4, cp)
    vec[0, :c] = gamma * invstd
    vec[1, :c] = beta - mean.float() * gamma * invstd
    vec[2, :c] = mean.float()
    vec[3, :c] = invstd
    return gamma, beta, vec.cuda()


@pytest.mark.parametrize("case", [
    # N, Ci, Co, D, H, W, cs of the gradient / raw-output buffers, channel offset
    (1, 8, 8, 8, 8, 32, 8, 0),          # 8 -> 8: (shift, channel) tiles on both sides, 4x4x16 boxes
    (2, 8, 8, 4, 8, 16, 16, 8),         # ... in the second half of a concat-level buffer
    (1, 16, 8, 8, 4, 16, 8, 0),         # C -> 8
    (1, 8, 16, 4, 8, 24, 16, 0),        # 8 -> C
    (1, 16, 16, 8, 8, 16, 32, 16),      # full 16 x 16 tile, slice of a wider buffer
    (1, 32, 16, 4, 4, 16, 16, 0),       # two input-channel tiles share a gradient tile (identical gy stores)
])
def test_conv3d_wgrad_with_lazy_batchnorm_backward(case):
    """ops.conv3d_wgrad_bn (BatchNorm + ReLU backward folded into the weight-gradient kernel's staging) against the
    three-pass path it replaces (reduce, finalize, in-place apply, plain weight gradient) and against fp64 autograd of
    relu(batch_norm(y)) for the raw-output gradient it writes out."""
    ops = _ops()
    n, ci, co, d, h, w, cs, c0 = case
    cip, cop = ops.pad8(ci), ops.pad8(co)
    assert ops.conv3d_wgrad_bn_supported((n, d, h, w), 3, cip, cop)
    x = torch.randn(n, ci, d, h, w, generator=g(41))
    y = torch.randn(n, co, d, h, w, generator=g(42)) * 1.3 + 0.3
    ga = torch.randn(n, co, d, h, w, generator=g(43))
    gamma, beta, vec = _bn_setup(ops, y, co, cop, 44)
    sc, sh = xf_vectors(ci, cip, 46)
    xc = to_cl(x).with_xf(sc.cuda(), sh.cuda(), True)
    yc, gac = to_cl(y, cs, c0, cop), to_cl(ga, cs, c0, cop)
    part = torch.empty(ops.bn_bwd_partials_floats(n * d * h * w, cop), device="cuda")
    ws = torch.empty(ops.conv3d_wgrad_ws((n, d, h, w), 3, cip, cop), device="cuda")
    # lazy path
    dg1, db1, coef = ops.bn_relu_bwd(yc, gac, vec, gamma.cuda(), co, part, lazy=True)
    gy = ops.CL(torch.full_like(gac.buf, float("nan")), c0, cop)
    ga_before = gac.buf.clone()
    dw1 = ops.conv3d_wgrad_bn(xc, gac, yc, vec, coef, gy, co, ci, 3, None, ws)
    torch.cuda.synchronize()
    assert torch.equal(gac.buf, ga_before)                      # the activated-output gradient is left alone
    # three-pass path
    gac2 = to_cl(ga, cs, c0, cop)
    dg2, db2 = ops.bn_relu_bwd(yc, gac2, vec, gamma.cuda(), co, part)
    dw2, _ = ops.conv3d_wgrad(xc, gac2, co, ci, 3, None, ws, False)
    torch.cuda.synchronize()
    assert torch.equal(dg1, dg2) and torch.equal(db1, db2)
    gy_l, gy_3 = from_cl(gy, co), from_cl(gac2, co)
    assert not torch.isnan(gy_l).any()
    assert rel_err(gy_l, gy_3) < 2e-6
    assert rel_err(dw1.cpu(), dw2.cpu()) < 1e-5
    if cs > cop:                                                # channels outside the slice are never written
        other = torch.ones(cs, dtype=torch.bool); other[c0:c0 + cop] = False
        assert torch.isnan(gy.buf[..., other.cuda()]).all()
    # fp64 autograd of relu(batch_norm(y)) for the raw-output gradient
    y64 = y.double().requires_grad_(True)
    a = F.relu(F.batch_norm(y64, None, None, gamma.double(), beta.double(), True, 0.1, 1e-5))
    a.backward(ga.double())
    assert rel_err(gy_l.double(), y64.grad) < 2e-4


@pytest.mark.parametrize("c,co,dims,cs", [
    (16, 8, (1, 4, 4, 16), 8),          # 8 padded output channels, gradient stride 8: the (w-parity, c_out) tile
    (16, 8, (1, 4, 8, 8), 16),          # 8 output channels inside a 16-wide buffer: the plain tile, two quads masked
    (32, 16, (2, 4, 4, 8), 16),
    (16, 32, (1, 4, 4, 8), 32),         # two gradient channel tiles
])
def test_upconv_fused_wgrad_with_lazy_batchnorm_backward(c, co, dims, cs):
    """ops.upconv_fused_wgrad(lazy=...) against the same call on the gradient a separate BatchNorm-backward pass produced."""
    ops = _ops()
    n, d, h, w = dims
    cp, cop = ops.pad8(c), ops.pad8(co)
    assert ops.upconv_fused_wgrad_bn_supported(dims, cp, cop)
    x = torch.randn(n, c, d, h, w, generator=g(51))
    wt = torch.randn(c, c, 2, 2, 2, generator=g(52)) * 0.2
    bt = torch.randn(c, generator=g(53))
    w3 = torch.randn(co, c, 3, 3, 3, generator=g(54)) * 0.1
    y = torch.randn(n, co, 2 * d, 2 * h, 2 * w, generator=g(55)) * 1.2 - 0.2
    ga = torch.randn(n, co, 2 * d, 2 * h, 2 * w, generator=g(56))
    gamma, beta, vec = _bn_setup(ops, y, co, cop, 57)
    sc, sh = xf_vectors(c, cp, 59)
    xc = to_cl(x).with_xf(sc.cuda(), sh.cuda(), True)
    _, _, pws = ops.upconv_fused_pack(wt.cuda(), bt.cuda(), w3.cuda(), None, cp, cop)
    yc, gac = to_cl(y, cs, 0, cop), to_cl(ga, cs, 0, cop)
    part = torch.empty(ops.bn_bwd_partials_floats(yc.nvox, cop), device="cuda")
    _, _, coef = ops.bn_relu_bwd(yc, gac, vec, gamma.cuda(), co, part, lazy=True)
    gy = ops.CL(torch.full_like(gac.buf, float("nan")), 0, cop)
    got = ops.upconv_fused_wgrad(xc, gac, c, co, bt.cuda(), pws, None, (yc, vec, coef, gy))
    gac2 = to_cl(ga, cs, 0, cop)
    ops.bn_relu_bwd(yc, gac2, vec, gamma.cuda(), co, part)
    want = ops.upconv_fused_wgrad(xc, gac2, c, co, bt.cuda(), pws, None)
    torch.cuda.synchronize()
    assert rel_err(from_cl(gy, co), from_cl(gac2, co)) < 2e-6
    for name, a, b in zip(("dWT", "dbT", "dW3"), got, want):
        assert rel_err(a.cpu(), b.cpu()) < 1e-5, name


@pytest.mark.parametrize("case", [(1, 1, 8, 8, 8, 32, 8, 0), (2, 2, 7, 6, 5, 40, 16, 8), (1, 1, 3, 4, 4, 16, 8, 0)])
def test_first_layer_wgrad_with_lazy_batchnorm_backward(case):
    """ops.conv_first_wgrad_bn against the three-pass path (reduce, finalize, in-place apply, plain weight gradient);
    ragged boxes included (the first-layer kernel checks bounds per item)."""
    ops = _ops()
    n, ci, co, d, h, w, cs, c0 = case
    x = torch.randn(n, ci, d, h, w, generator=g(61)).cuda()
    y = torch.randn(n, co, d, h, w, generator=g(62)) * 0.9 - 0.1
    ga = torch.randn(n, co, d, h, w, generator=g(63))
    gamma, beta, vec = _bn_setup(ops, y, co, 8, 64)
    yc, gac = to_cl(y, cs, c0, 8), to_cl(ga, cs, c0, 8)
    part = torch.empty(ops.bn_bwd_partials_floats(n * d * h * w, 8), device="cuda")
    ws = torch.empty(ops.conv_first_wgrad_ws((n, d, h, w), ci), device="cuda")
    _, _, coef = ops.bn_relu_bwd(yc, gac, vec, gamma.cuda(), co, part, lazy=True)
    gy = ops.CL(torch.full_like(gac.buf, float("nan")), c0, 8)
    dw1 = ops.conv_first_wgrad_bn(x, gac, yc, vec, coef, gy, co, ws)
    gac2 = to_cl(ga, cs, c0, 8)
    ops.bn_relu_bwd(yc, gac2, vec, gamma.cuda(), co, part)
    dw2 = ops.conv_first_wgrad(x, gac2, co, ws)
    torch.cuda.synchronize()
    assert rel_err(from_cl(gy, co), from_cl(gac2, co)) < 2e-6
    assert torch.all(from_cl(gy, 8)[:, co:] == 0)
    assert rel_err(dw1.cpu(), dw2.cpu()) < 1e-5
