"""Single-layer microbenchmark through the C ABI (dev tool): conv fwd / wgrad / convT at a given shape.
usage: bench_layer.py <op> <cin> <cout> <size> [k] [iters]   op in {fwd, wgrad, wgrad_bn, convt, convt_bwd, convt_wgrad}
CTU_DT=bf16|f16 in the environment: the 16-bit (ctu_lp_*) kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ct-unet_amd")]
import torch
from ctunet_amd import ops, _lib
if os.environ.get("CTU_LIB"):          # dev: A/B a differently built library
    _lib.LIB_PATH = os.environ["CTU_LIB"]

op, ci, co, s = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
cip, cop = ops.pad8(ci), ops.pad8(co)
dev = "cuda"
DT = {"bf16": torch.bfloat16, "f16": torch.float16}.get(os.environ.get("CTU_DT", ""), torch.float32)
LP = DT != torch.float32
torch.manual_seed(0)
x = ops.CL(torch.randn(1, s, s, s, cip, device=dev).to(DT), 0, cip)
sc, sh = torch.rand(cip, device=dev) + 0.5, torch.randn(cip, device=dev) * 0.1
xx = x.with_xf(sc, sh, True)
flops = 2.0 * ci * co * k ** 3 * s ** 3
if op == "fwd":
    w = torch.randn(co, ci, k, k, k, device=dev) * 0.1
    lay = ops.conv_layout(k, cop, s, DT, cip)
    if os.environ.get("CTU_LAY") is not None:
        lay = int(os.environ["CTU_LAY"])
    wp = ops.pack_conv_w_lp(w, None, cip, cop, 0, DT, None, lay) if LP else ops.pack_conv_w(w, None, cip, cop, 0, lay)
    out = ops.CL(torch.empty(1, s, s, s, cop, device=dev, dtype=DT), 0, cop)
    nb = ops.conv_num_blocks((1, s, s, s), cop, lay, k, DT, cip)
    stats = torch.empty(nb, 2, cop, device=dev)
    fn = lambda: ops.conv3d_fwd(xx, wp, None, out, k, stats, None, lay)
elif op == "wgrad":
    g = ops.CL(torch.randn(1, s, s, s, cop, device=dev).to(DT), 0, cop)
    ws = torch.empty(ops.conv3d_wgrad_ws((1, s, s, s), k, cip, cop, DT), device=dev)
    fn = lambda: ops.conv3d_wgrad(xx, g, co, ci, k, None, ws, False)
elif op == "wgrad_bn":      # weight gradient with the BatchNorm + ReLU backward folded in (fp32)
    g = ops.CL(torch.randn(1, s, s, s, cop, device=dev), 0, cop)
    y = ops.CL(torch.randn(1, s, s, s, cop, device=dev), 0, cop)
    gy = ops.CL(torch.empty(1, s, s, s, cop, device=dev), 0, cop)
    vec = torch.rand(4, cop, device=dev) + 0.5
    coef = torch.randn(5, cop, device=dev) * 0.1
    ws = torch.empty(ops.conv3d_wgrad_ws((1, s, s, s), k, cip, cop, DT), device=dev)
    fn = lambda: ops.conv3d_wgrad_bn(xx, g, y, vec, coef, gy, co, ci, k, None, ws)
elif op == "convt":
    w = torch.randn(ci, co, 2, 2, 2, device=dev) * 0.1
    wp = ops.pack_convt_w_lp(w, None, cip, cop, 0, DT) if LP else ops.pack_convt_w(w, None, cip, cop, 0)
    out = ops.CL(torch.empty(1, 2 * s, 2 * s, 2 * s, cop, device=dev, dtype=DT), 0, cop)
    b = torch.zeros(co, device=dev)
    fn = lambda: ops.convt_fwd(xx, wp, b, out)
    flops = 2.0 * ci * co * 8 * s ** 3
elif op == "convt_bwd":
    w = torch.randn(ci, co, 2, 2, 2, device=dev) * 0.1
    wp = ops.pack_convt_w_lp(w, None, cop, cip, 1, DT) if LP else ops.pack_convt_w(w, None, cop, cip, 1)
    g = ops.CL(torch.randn(1, 2 * s, 2 * s, 2 * s, cop, device=dev).to(DT), 0, cop)
    gin = ops.CL(torch.empty(1, s, s, s, cip, device=dev, dtype=DT), 0, cip)
    fn = lambda: ops.convt_bwd_data(g, wp, gin)
    flops = 2.0 * ci * co * 8 * s ** 3
elif op == "convt_wgrad":
    g = ops.CL(torch.randn(1, 2 * s, 2 * s, 2 * s, cop, device=dev).to(DT), 0, cop)
    ws = torch.empty(ops.convt_wgrad_ws((1, s, s, s), cip, cop, DT), device=dev)
    fn = lambda: ops.convt_wgrad(xx, g, ci, co, None, ws)
    flops = 2.0 * ci * co * 8 * s ** 3
else:
    raise SystemExit("bad op")
for _ in range(3):
    fn()
torch.cuda.synchronize()
a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    fn()
b_.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b_) / iters
esz = 2 if LP else 4
nbytes = esz * s ** 3 * (cip + (8 if op.startswith("convt") else 1) * cop)
print(f"{op} {ci}->{co} k{k} @{s}^3 {os.environ.get('CTU_DT', 'f32')}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s  {nbytes/ms/1e6:.0f} GB/s algorithmic")
