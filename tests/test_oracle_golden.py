"""Pins the CPU oracle (oracle/unet_oracle.py) to outputs of the reference itself.

The fixtures under tests/golden/ were produced by importing the reference's own modules
(tests/golden/make_golden.py).  Tolerances: same torch build, same ops -> differences are
summation-order only; 1e-5 relative (observed ~1e-7)."""
import numpy as np
import pytest
import torch

from oracle import unet_oracle as O
from util import CLASS_INPUT, close_summary, gen, load_json, load_npz, onehot_target, rel_err, sd_from, summarize

TINY = {
    "tiny_unet.npz": O.NetSpec(in_ch=1, out_ch=2, n_blocks=2, i_size=3),
    "tiny_unet_add.npz": O.NetSpec(in_ch=1, out_ch=2, n_blocks=2, i_size=3, cat=False, apply_softmax=True),
    "tiny_unet_noskip.npz": O.NetSpec(in_ch=1, out_ch=2, n_blocks=2, i_size=3, skip=False),
    "tiny_unet_sp.npz": O.NetSpec(in_ch=2, out_ch=3, n_blocks=2, i_size=3, head="sp"),
    "tiny_legacy.npz": O.NetSpec(family="legacy", in_ch=1, out_ch=2, i_size=1, k=5, pad=2),
    # mask-stable fixtures (make_golden.py stabilize_bn: every pre-activation 6 sigma from zero): train step only
    "tiny_unet_stable.npz": O.NetSpec(in_ch=1, out_ch=2, n_blocks=2, i_size=3),
    "tiny_unet_sp_stable.npz": O.NetSpec(in_ch=2, out_ch=3, n_blocks=2, i_size=3, head="sp"),
    "tiny_legacy_stable.npz": O.NetSpec(family="legacy", in_ch=1, out_ch=2, i_size=1, k=5, pad=2),
}


def _loss_fn(name, rec):
    if "_sp" in name:
        t = (torch.from_numpy(rec["target_sk"]), torch.from_numpy(rec["target_fl"]))
        return lambda out: O.loss_double(out, t, 1.0, 1.0)[0]
    t = torch.from_numpy(rec["target"])
    return lambda out: O.loss_single(out, t, 1.0, 1.0)[0]


@pytest.mark.parametrize("name", list(TINY))
def test_tiny_net_eval_train_grads_buffers(name):
    rec, spec = load_npz(name), TINY[name]
    x = torch.from_numpy(rec["x"])
    if "eval_out0" in rec:
        sd = sd_from(rec)
        out = O.forward(spec, sd, x, training=False)
        outs = out if isinstance(out, tuple) else (out,)
        for i, o in enumerate(outs):
            assert rel_err(o, torch.from_numpy(rec[f"eval_out{i}"])) < 1e-5
    sd = sd_from(rec)
    out, loss, grads, dx = O.grads(spec, sd, x, _loss_fn(name, rec), training=True)
    outs = out if isinstance(out, tuple) else (out,)
    for i, o in enumerate(outs):
        assert rel_err(o, torch.from_numpy(rec[f"train_out{i}"])) < 1e-5
    assert abs(loss.item() - float(rec["train_loss"])) < 1e-6
    assert rel_err(dx, torch.from_numpy(rec["train_dx"])) < 1e-4
    for k, v in rec.items():
        if k.startswith("grad."):
            nm = k[5:]
            if v.size == 0:                       # the reference left .grad = None (dead centre block)
                assert grads[nm] is None
            else:
                assert rel_err(grads[nm], torch.from_numpy(v)) < 1e-4, nm
        if k.startswith("post."):
            assert np.allclose(sd[k[5:]].numpy(), v, rtol=1e-5, atol=1e-6), k


def test_checkpoint_double_bn_update():
    """use_checkpoint=True (the shipped default): every BN of a re-computed block applies its momentum update
    twice per step with the same batch statistics; the dead centre block (never re-computed) only once."""
    rec, spec = load_npz("tiny_unet.npz"), TINY["tiny_unet.npz"]
    sd = sd_from(rec)
    before = {k: v.clone() for k, v in sd.items()}
    O.forward(spec, sd, torch.from_numpy(rec["x"]), training=True)          # first update, from the forward
    n_per_level = {}                                                         # prefix -> (mean, biased var, n)
    stats = {}
    x = torch.from_numpy(rec["x"])
    for k in list(sd):
        if not k.endswith(".running_mean"):
            continue
        p = k[:-len(".running_mean")]
        mean = (sd[k] - (1 - O.BN_MOMENTUM) * before[k]) / O.BN_MOMENTUM            # batch mean, from the update
        unb = (sd[p + ".running_var"] - (1 - O.BN_MOMENTUM) * before[p + ".running_var"]) / O.BN_MOMENTUM
        level = int(p.split(".")[1]) if p.startswith("d_blocks") else (1 - int(p.split(".")[1]) if p.startswith("u_blocks") else 2)
        n = x.shape[0] * x[0, 0].numel() // (8 ** level)
        stats[p] = (mean, unb * (n - 1) / n, n)
    O.bn_checkpoint_replay(spec, sd, stats)                                  # second update, from the recompute
    for k, v in rec.items():
        if k.startswith("chk_post."):
            assert np.allclose(sd[k[9:]].numpy(), v, rtol=1e-4, atol=1e-5), k
    assert rec["chk_dead_grad_is_none"].all()
    assert int(rec["chk_post.cblock.block.1.num_batches_tracked"]) == 1
    assert int(rec["chk_post.d_blocks.0.block.1.num_batches_tracked"]) == 2


def test_losses_against_reference():
    rec = load_npz("losses.npz")
    p, t = torch.from_numpy(rec["p"]), torch.from_numpy(rec["t"])
    p2, t2 = torch.from_numpy(rec["p2"]), torch.from_numpy(rec["t2"])
    assert abs(O.dice_loss(p, t).item() - float(rec["dice"])) < 1e-6
    half = torch.zeros(1, 1, 4, 4, 4); half.view(-1)[:32] = 1
    assert abs(O.dice_loss(torch.full_like(half, 0.5), half).item() - 1 / 3) < 1e-6
    assert abs(float(rec["dice_kat"]) - 1 / 3) < 1e-6
    for ce, dc in [(1.0, 1.0), (0.0, 1.0), (1.0, 0.0), (0.5, 2.0)]:
        pi = p.clone().requires_grad_(True)
        tot, parts = O.loss_single(pi, t, ce, dc)
        tot.backward()
        tag = f"single_{ce}_{dc}"
        assert abs(tot.item() - float(rec[tag + "_loss"])) < 1e-6
        assert rel_err(pi.grad, torch.from_numpy(rec[tag + "_grad"])) < 1e-5
        assert sorted(list(parts) + ["epoch_loss"]) == list(rec[tag + "_keys"])
        pa, pb = p.clone().requires_grad_(True), p2.clone().requires_grad_(True)
        tot, parts = O.loss_double((pa, pb), (t, t2), ce, dc)
        tot.backward()
        tag = f"double_{ce}_{dc}"
        assert abs(tot.item() - float(rec[tag + "_loss"])) < 1e-6
        assert rel_err(pa.grad, torch.from_numpy(rec[tag + "_grad0"])) < 1e-5
        assert rel_err(pb.grad, torch.from_numpy(rec[tag + "_grad1"])) < 1e-5
        for k_, v_ in parts.items():
            assert abs(v_.item() - float(rec[tag + "_lm_" + k_])) < 1e-6
    assert np.array_equal(O.hard_segmentation(p).numpy(), rec["hard_segm"])


@pytest.mark.parametrize("name", list(CLASS_INPUT))
def test_shipped_class_checksums(name):
    """Full-size shipped classes: weights from seed 0 via the drop-in classes' own init, forward/backward
    through the ORACLE, compared with checksums recorded from the reference."""
    import ctunet_amd
    exp = load_json("class_checksums.json")[name]
    torch.manual_seed(0)
    net = getattr(ctunet_amd, name)()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    assert {k: list(v.shape) for k, v in sd.items()} == exp["keys_shapes"]
    assert abs(sum(p.double().sum().item() for p in net.parameters()) - exp["param_sum"]) < 1e-6
    in_ch, s = CLASS_INPUT[name]
    x = torch.randn(1, in_ch, s, s, s, generator=gen(1234))
    spec = O.SPECS[name]
    out = O.forward(spec, sd, x, training=False)
    outs = out if isinstance(out, tuple) else (out,)
    for o, e in zip(outs, exp["eval"]):
        assert close_summary(summarize(o), e, 1e-5, 1e-7)
    tg = [onehot_target((1, 2, s, s, s), 4321 + i, 0.2) for i in range(len(outs))]
    if len(outs) == 2:
        fn = lambda o: O.loss_double(o, tg, 1.0, 1.0)[0]
    elif outs[0].shape[1] == 2:
        fn = lambda o: O.loss_single(o, tg[0], 1.0, 1.0)[0]
    else:
        fn = lambda o: (o ** 2).mean()
    out, loss, grads, dx = O.grads(spec, sd, x, fn, training=True)
    outs = out if isinstance(out, tuple) else (out,)
    for o, e in zip(outs, exp["train"]):
        assert close_summary(summarize(o), e, 1e-4, 1e-7)
    assert abs(loss.item() - exp["loss"]) < 1e-5
    assert close_summary(summarize(dx), exp["dx"], 1e-3, 1e-9)
    for nm, e in exp["grads"].items():
        if e is None:
            assert grads[nm] is None, nm
        else:
            assert close_summary(summarize(grads[nm]), e, 2e-3, 1e-8), nm
    for nm, e in exp["post_buffers"].items():
        got = sd[nm]
        if isinstance(e, dict):
            assert close_summary(summarize(got.float()), e, 1e-4, 1e-7), nm
        else:
            assert float(got) == e, nm
