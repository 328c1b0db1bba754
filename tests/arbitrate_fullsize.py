"""Arbitration of the full-size fp32-vs-fp32 gradient disagreement: one train step of <class> at <size>^3 through
the HIP path, the ATen-CPU fp32 oracle AND an fp64 run of the oracle; every gradient tensor of both fp32 sides is judged
against fp64 (largest error relative to the tensor's largest entry, and cosine).  Not collected by pytest (minutes of host
time); run once on the GPU box and keep the table under profiles/:

    python tests/arbitrate_fullsize.py UNetSP 256 > profiles/r03_arbitrate_UNetSP_256.txt
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "ct-unet_amd"), HERE]
from test_models_gpu import oracle_train_check  # noqa: E402

name, size = sys.argv[1], int(sys.argv[2])
rows = []
verdict = "fp64 rule holds (no worse than max(5x the ATen-CPU fp32 error, 2e-3 of the tensor's scale), cosine >= 0.995)"
try:
    oracle_train_check(name, size, want_fp64=True, report=rows)
except AssertionError as e:
    verdict = "fp64 rule VIOLATED: " + str(e)[:2000]
print(f"# {name} {size}^3 train step: gradient tensors against an fp64 run of the oracle")
print(f"# verdict: {verdict}")
print(f"{'tensor':42s} {'scale':>10s} {'HIP err/scale':>14s} {'ATen err/scale':>15s} {'HIP cos':>10s} {'ATen cos':>10s}")
for n_, sc, eh, ea, ch, ca in rows:
    print(f"{n_:42s} {sc:10.3e} {eh / (sc + 1e-300):14.3e} {ea / (sc + 1e-300):15.3e} {ch:10.6f} {ca:10.6f}")
