"""dev: compare the fused up-convolution weight gradient (dW_eff) of two library builds on the same operands."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ct-unet_amd")]
import torch
from ctunet_amd import _lib
lib = _lib.load()
alt = C.CDLL(os.path.join(ROOT, "scripts/build/lib_up4off.so"))
n, d, h, w, cin = 1, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mode = sys.argv[5] if len(sys.argv) > 5 else "rand"
torch.manual_seed(0)
x = torch.randn(n, d, h, w, cin, device="cuda")
g = torch.randn(n, 2 * d, 2 * h, 2 * w, 8, device="cuda")
if mode == "ones":
    x = torch.ones_like(x); g = torch.ones_like(g)
x = x.bfloat16(); g = g.bfloat16()
P = C.c_void_p
res = []
for L in (lib, alt):
    L.ctu_lp_upconv_fused_wgrad_ws_floats.restype = C.c_size_t
    L.ctu_lp_upconv_fused_wgrad_ws_floats.argtypes = [C.c_int] * 5
    nws = L.ctu_lp_upconv_fused_wgrad_ws_floats(n, d, h, w, cin)
    ws = torch.zeros(nws, device="cuda")
    dweff = torch.zeros(8, 8, cin, 8, device="cuda")
    L.ctu_lp_upconv_fused_wgrad.argtypes = [C.c_int, P, C.c_int, C.c_int, P, P, C.c_int, P, C.c_int, P, P] + [C.c_int] * 4 + [P]
    rc = L.ctu_lp_upconv_fused_wgrad(1, x.data_ptr(), cin, cin, None, None, 0, g.data_ptr(), 8, dweff.data_ptr(), ws.data_ptr(), n, d, h, w,
                                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0, rc
    res.append(dweff.cpu())
a, b = res
print("max |new - old|", float((a - b).abs().max()), "scale", float(b.abs().max()))
bad = ((a - b).abs() > 1e-3 * b.abs().max()).nonzero()
print("bad entries", len(bad), "of", a.numel())
import collections
print("by parity", collections.Counter(int(i[0]) for i in bad))
print("by tap", collections.Counter(int(i[1]) for i in bad))
print("by ci", collections.Counter(int(i[2]) for i in bad))
print("by co", collections.Counter(int(i[3]) for i in bad))
for i in bad[:10]:
    print(tuple(int(v) for v in i), float(a[tuple(i)]), float(b[tuple(i)]))
