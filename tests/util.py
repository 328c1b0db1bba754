"""Shared helpers for the test-suite (test infrastructure, not product code)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def sd_from(rec, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in rec.items() if k.startswith(prefix)}


def gen(seed):
    return torch.Generator().manual_seed(seed)


def onehot_target(shape, seed, p=0.3):
    """Same generator as tests/golden/make_golden.py::onehot_target."""
    n, _, d, h, w = shape
    m = (torch.rand(n, d, h, w, generator=gen(seed)) < p).long()
    return torch.nn.functional.one_hot(m, 2).movedim(4, 1).float().contiguous()


def summarize(t):
    """Same statistics as make_golden.summarize."""
    f = t.detach().flatten().double().cpu()
    idx = torch.linspace(0, f.numel() - 1, 16).long()
    return {"mean": f.mean().item(), "std": f.std().item(), "abs_sum": f.abs().sum().item(), "sample": f[idx].tolist()}


def close_summary(got, exp, rtol, atol):
    ok = abs(got["mean"] - exp["mean"]) <= atol + rtol * abs(exp["mean"])
    ok &= abs(got["std"] - exp["std"]) <= atol + rtol * abs(exp["std"])
    ok &= abs(got["abs_sum"] - exp["abs_sum"]) <= rtol * abs(exp["abs_sum"]) + atol
    ok &= bool(np.allclose(got["sample"], exp["sample"], rtol=rtol, atol=atol))
    return ok


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


CLASS_INPUT = {  # name -> (in_ch, size) as in make_golden.class_checksums
    "UNet": (1, 32), "UNet4b2i3o": (2, 32), "UNet5b2i3o": (2, 64), "UNet4b1i3o": (1, 32), "UNetSP": (2, 32),
    "UNetSPSmall": (2, 64), "UNetDO": (1, 32), "recAE_v2_fixed": (1, 32), "UNet4_2IC": (2, 32),
}
