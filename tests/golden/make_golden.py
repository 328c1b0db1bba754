#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE itself.

Runs only in the build container (needs /root/reference; the reference never
travels to the GPU box).  It imports the reference's own modules --
``ctunet/pytorch/models.py`` by file path (it needs only torch) and
``ctunet/utilities.py`` / ``ctunet/pytorch/ProblemHandler.py`` with the absent
third-party packages (SimpleITK, monai, ...) replaced by MagicMock entries --
and records inputs, outputs, gradients and post-step buffers as small data
files.  Nothing from the reference's source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

torch 2.10.0+rocm7.0 CPU, 8 threads.  Fixtures: tiny_unet.npz, tiny_unet_add.npz,
tiny_unet_noskip.npz, tiny_unet_sp.npz, tiny_legacy.npz, tiny_{unet,unet_sp,legacy}_stable.npz, losses.npz,
class_checksums.json, ini_params.json.
"""
import glob
import importlib.util
import json
import os
import sys
import unittest.mock

sys.dont_write_bytecode = True
import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_models", f"{REF}/ctunet/pytorch/models.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for n in ["SimpleITK", "raster_geometry", "monai", "monai.metrics", "torchio", "torchvision",
              "torchvision.transforms", "tensorboard", "torch.utils.tensorboard"]:
        sys.modules[n] = unittest.mock.MagicMock(name=n)
    sys.path.insert(0, REF)
    from ctunet import utilities as U
    from ctunet.pytorch import ProblemHandler as PH
    return ref, U, PH


def gen(seed):
    return torch.Generator().manual_seed(seed)


def onehot_target(shape, seed, p=0.3):
    """[N,2,D,H,W] float one-hot of a Bernoulli mask, contiguous (datasets.py:89-112 schema)."""
    n, _, d, h, w = shape
    m = (torch.rand(n, d, h, w, generator=gen(seed)) < p).long()
    return torch.nn.functional.one_hot(m, 2).movedim(4, 1).float().contiguous()


def to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


class Holder:
    """Stands in for ctunet.Model in comp_losses_metrics (Model.py:101,363)."""

    def __init__(self, ce, dice):
        self.params = dict(ce_lambda=ce, dice_lambda=dice, save_dice_plots=False, save_hd_plots=False)
        self.losses_and_metrics = {}
        self.pt_loss = None


def randomize_bn(net, seed):
    """Non-trivial gamma/beta/running stats so eval-mode parity is sensitive."""
    g = gen(seed)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 1.5 - 0.25)  # some negative
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def step_record(net, x, loss_fn, out_key="out"):
    """One train-mode forward+backward; returns outputs, grads, post-step buffers."""
    net.train()
    for p in net.parameters():
        p.grad = None
    xi = x.clone().requires_grad_(True)
    out = net(xi)
    loss = loss_fn(out)
    loss.backward()
    rec = {}
    outs = out if isinstance(out, (tuple, list)) else (out,)
    for i, o in enumerate(outs):
        rec[f"train_{out_key}{i}"] = o.detach().numpy()
    rec["train_loss"] = np.float64(loss.item())
    rec["train_dx"] = xi.grad.numpy()
    for n_, p in net.named_parameters():
        rec["grad." + n_] = p.grad.numpy() if p.grad is not None else np.zeros(0, np.float32)
    for n_, b in net.named_buffers():
        rec["post." + n_] = b.detach().numpy().copy()
    return rec


def tiny_generic(ref, U, PH):
    torch.manual_seed(7)
    net = ref.UNet(input_channels=1, out_channels=2, n_blocks=2, i_size=3, use_checkpoint=False)
    randomize_bn(net, 11)
    x = torch.randn(2, 1, 16, 16, 16, generator=gen(21))
    tgt = onehot_target(x.shape, 31)
    rec = {"x": x.numpy(), "target": tgt.numpy()}
    rec.update({"sd." + k: v for k, v in to_np(net.state_dict()).items()})
    net.eval()
    with torch.no_grad():
        rec["eval_out0"] = net(x).numpy()

    def loss_fn(out):
        h = Holder(1.0, 1.0)
        PH.ProblemHandler.comp_losses_metrics(h, out, tgt, 0, 1)
        return h.pt_loss
    rec.update(step_record(net, x, loss_fn))
    # the shipped default use_checkpoint=True: buffers after ONE step
    torch.manual_seed(7)
    net2 = ref.UNet(input_channels=1, out_channels=2, n_blocks=2, i_size=3, use_checkpoint=True)
    net2.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("sd.")})
    r2 = step_record(net2, x, loss_fn)
    for k, v in r2.items():
        if k.startswith("post."):
            rec["chk_" + k] = v
    rec["chk_dead_grad_is_none"] = np.array([int(p.grad is None) for n_, p in net2.named_parameters() if n_.startswith("cblock.")])
    np.savez_compressed(f"{HERE}/tiny_unet.npz", **rec)
    print("tiny_unet", rec["eval_out0"].mean(), rec["train_loss"])


def tiny_skip_modes(ref, U, PH):
    """UNet(cat=False) (additive skips, models.py:250-251) and UNet(use_skip_connections=False) (models.py:252-253):
    options no shipped class sets but that are live in UNet.__init__/forward.  apply_softmax is switched on for the
    first so the softmax-then-sigmoid head (models.py:258-259) is pinned as well."""
    for tag, kw in (("add", dict(cat=False, apply_softmax=True)), ("noskip", dict(use_skip_connections=False))):
        torch.manual_seed(17)
        net = ref.UNet(input_channels=1, out_channels=2, n_blocks=2, i_size=3, use_checkpoint=False, **kw)
        randomize_bn(net, 13)
        x = torch.randn(2, 1, 16, 16, 16, generator=gen(23))
        tgt = onehot_target(x.shape, 33)
        rec = {"x": x.numpy(), "target": tgt.numpy()}
        rec.update({"sd." + k: v for k, v in to_np(net.state_dict()).items()})
        net.eval()
        with torch.no_grad():
            rec["eval_out0"] = net(x).numpy()

        def loss_fn(out):
            h = Holder(1.0, 1.0)
            PH.ProblemHandler.comp_losses_metrics(h, out, tgt, 0, 1)
            return h.pt_loss
        rec.update(step_record(net, x, loss_fn))
        np.savez_compressed(f"{HERE}/tiny_unet_{tag}.npz", **rec)
        print("tiny_unet_" + tag, rec["eval_out0"].mean(), rec["train_loss"])


def tiny_sp(ref, U, PH):
    """UNetSP.forward (models.py:317-330) on a small net: i_size 3 (widths 3/6, not multiples
    of 8), 2 in / 3 out, SP re-encoding, double-out loss."""
    class TinySP(ref.UNetSP):
        def __init__(self):          # same class, smaller hyper-parameters than models.py:272-278
            ref.UNet.__init__(self, input_channels=2, out_channels=3, n_blocks=2, i_size=3,
                              use_checkpoint=False)

    torch.manual_seed(8)
    net = TinySP()
    randomize_bn(net, 12)
    x = torch.randn(1, 2, 16, 16, 16, generator=gen(22))
    t_sk, t_fl = onehot_target((1, 2, 16, 16, 16), 32), onehot_target((1, 2, 16, 16, 16), 33, 0.1)
    rec = {"x": x.numpy(), "target_sk": t_sk.numpy(), "target_fl": t_fl.numpy()}
    rec.update({"sd." + k: v for k, v in to_np(net.state_dict()).items()})
    net.eval()
    with torch.no_grad():
        sk, fl = net(x)
        rec["eval_out0"], rec["eval_out1"] = sk.numpy(), fl.numpy()

    def loss_fn(out):
        h = Holder(1.0, 1.0)
        PH.FlapRecWithShapePriorDoubleOut.comp_losses_metrics(h, out, (t_sk, t_fl), 0, 1)
        return h.pt_loss
    rec.update(step_record(net, x, loss_fn))
    np.savez_compressed(f"{HERE}/tiny_unet_sp.npz", **rec)
    print("tiny_unet_sp", rec["eval_out0"].mean(), rec["train_loss"])


def tiny_legacy(ref, U, PH):
    torch.manual_seed(9)
    net = ref.recAE_v2_fixed(input_channels=1, i_size=1, use_checkpoint=False)
    randomize_bn(net, 13)
    x = torch.randn(1, 1, 32, 32, 32, generator=gen(23))
    tgt = onehot_target(x.shape, 34)
    rec = {"x": x.numpy(), "target": tgt.numpy()}
    rec.update({"sd." + k: v for k, v in to_np(net.state_dict()).items()})
    net.eval()
    with torch.no_grad():
        rec["eval_out0"] = net(x).numpy()

    def loss_fn(out):
        h = Holder(1.0, 1.0)
        PH.ProblemHandler.comp_losses_metrics(h, out, tgt, 0, 1)
        return h.pt_loss
    rec.update(step_record(net, x, loss_fn))
    np.savez_compressed(f"{HERE}/tiny_legacy.npz", **rec)
    print("tiny_legacy", rec["eval_out0"].mean(), rec["train_loss"])


def stabilize_bn(net, seed):
    """BatchNorm parameters that pin every ReLU mask: |beta| = 6 |gamma|, so a channel's pre-activations gamma*yhat + beta stay
    6 sigma away from zero -- two channels of three always on (beta > 0), one always off (beta < 0), at least one on per layer.
    A whole-net gradient comparison then has no mask flips to hide behind: forward and backward are smooth in the weights."""
    g = gen(seed)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            with torch.no_grad():
                c = m.weight.numel()
                gam = (torch.rand(c, generator=g) + 0.5) * (torch.randint(0, 2, (c,), generator=g).float() * 2 - 1)
                on = torch.tensor([1.0 if (i % 3 != 2 or c == 1) else -1.0 for i in range(c)])
                m.weight.copy_(gam)
                m.bias.copy_(6.0 * gam.abs() * on)
                m.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(c, generator=g) + 0.5)


def mask_stable(ref, U, PH):
    """tiny_*_stable.npz: the tiny generic / shape-prior / legacy nets with stabilize_bn's parameters -- weights, input, train
    outputs, loss and ALL gradients from the reference (the tight whole-net gradient gate of tests/test_models_gpu.py)."""
    class TinySP(ref.UNetSP):
        def __init__(self):
            ref.UNet.__init__(self, input_channels=2, out_channels=3, n_blocks=2, i_size=3, use_checkpoint=False)

    def single(out, tgt):
        h = Holder(1.0, 1.0)
        PH.ProblemHandler.comp_losses_metrics(h, out, tgt, 0, 1)
        return h.pt_loss

    def double(out, tgts):
        h = Holder(1.0, 1.0)
        PH.FlapRecWithShapePriorDoubleOut.comp_losses_metrics(h, out, tgts, 0, 1)
        return h.pt_loss
    cases = [("tiny_unet_stable", lambda: ref.UNet(input_channels=1, out_channels=2, n_blocks=2, i_size=3, use_checkpoint=False), 1, 16, 2, 51),
             ("tiny_unet_sp_stable", TinySP, 2, 16, 1, 52),
             ("tiny_legacy_stable", lambda: ref.recAE_v2_fixed(input_channels=1, i_size=1, use_checkpoint=False), 1, 32, 1, 53)]
    for tag, make, in_ch, size, batch, seed in cases:
        torch.manual_seed(seed)
        net = make()
        stabilize_bn(net, seed + 100)
        x = torch.randn(batch, in_ch, size, size, size, generator=gen(seed + 200))
        shape = (batch, 2, size, size, size)
        rec = {"x": x.numpy()}
        if "sp" in tag:
            t_sk, t_fl = onehot_target(shape, seed + 300), onehot_target(shape, seed + 301, 0.1)
            rec["target_sk"], rec["target_fl"] = t_sk.numpy(), t_fl.numpy()
            loss_fn = lambda out: double(out, (t_sk, t_fl))
        else:
            tgt = onehot_target(shape, seed + 300)
            rec["target"] = tgt.numpy()
            loss_fn = lambda out: single(out, tgt)
        rec.update({"sd." + k: v for k, v in to_np(net.state_dict()).items()})
        rec.update(step_record(net, x, loss_fn))
        # how far the fixture's masks are from flipping: smallest |pre-activation| / its channel's std, over all BatchNorms
        np.savez_compressed(f"{HERE}/{tag}.npz", **rec)
        print(tag, rec["train_loss"], max(np.abs(v).max() for k, v in rec.items() if k.startswith("grad.") and v.size))


def losses(ref, U, PH):
    g = gen(41)
    rec = {}
    p = torch.rand(2, 2, 8, 8, 8, generator=g)
    t = onehot_target((2, 2, 8, 8, 8), 42)
    rec["p"], rec["t"] = p.numpy(), t.numpy()
    rec["dice"] = np.float64(U.dice_loss()(p, t).item())
    # closed-form KAT: p == 0.5, half-ones mask -> 1/3
    half = torch.zeros(1, 1, 4, 4, 4); half.view(-1)[:32] = 1
    rec["dice_kat"] = np.float64(U.dice_loss()(torch.full_like(half, 0.5), half).item())
    for ce, dc in [(1.0, 1.0), (0.0, 1.0), (1.0, 0.0), (0.5, 2.0)]:
        pi = p.clone().requires_grad_(True)
        h = Holder(ce, dc)
        PH.ProblemHandler.comp_losses_metrics(h, pi, t, 0, 1)
        h.pt_loss.backward()
        tag = f"single_{ce}_{dc}"
        rec[tag + "_loss"] = np.float64(h.pt_loss.item())
        rec[tag + "_grad"] = pi.grad.numpy()
        rec[tag + "_keys"] = np.array(sorted(h.losses_and_metrics.keys()))
        p2 = torch.rand(2, 2, 8, 8, 8, generator=gen(43))
        t2 = onehot_target((2, 2, 8, 8, 8), 44, 0.1)
        pa, pb = p.clone().requires_grad_(True), p2.clone().requires_grad_(True)
        h = Holder(ce, dc)
        PH.FlapRecWithShapePriorDoubleOut.comp_losses_metrics(h, (pa, pb), (t, t2), 0, 1)
        h.pt_loss.backward()
        tag = f"double_{ce}_{dc}"
        rec[tag + "_loss"] = np.float64(h.pt_loss.item())
        rec[tag + "_grad0"], rec[tag + "_grad1"] = pa.grad.numpy(), pb.grad.numpy()
        rec[tag + "_keys"] = np.array(sorted(h.losses_and_metrics.keys()))
        for k_, v_ in h.losses_and_metrics.items():
            rec[tag + "_lm_" + k_] = np.float64(v_[0])
    rec["p2"], rec["t2"] = p2.numpy(), t2.numpy()
    rec["hard_segm"] = U.hard_segm_from_tensor(p).numpy()
    np.savez_compressed(f"{HERE}/losses.npz", **rec)
    print("losses", rec["dice"], rec["dice_kat"])


def summarize(t):
    f = t.detach().flatten().double()
    idx = torch.linspace(0, f.numel() - 1, 16).long()
    return {"mean": f.mean().item(), "std": f.std().item(), "abs_sum": f.abs().sum().item(),
            "sample": f[idx].tolist()}


def class_checksums(ref, U, PH):
    """Full-size shipped classes: weights from torch.manual_seed(0) + default init, checksums only."""
    out = {}
    for name in ["UNet", "UNet4b2i3o", "UNet5b2i3o", "UNet4b1i3o", "UNetSP", "UNetSPSmall", "UNetDO",
                 "recAE_v2_fixed", "UNet4_2IC"]:
        torch.manual_seed(0)
        net = getattr(ref, name)()
        net.chk = False
        sd = net.state_dict()
        in_ch = 2 if name in ("UNet4b2i3o", "UNet5b2i3o", "UNetSP", "UNetSPSmall", "UNet4_2IC") else 1
        s = 64 if name in ("UNet5b2i3o", "UNetSPSmall") else 32
        x = torch.randn(1, in_ch, s, s, s, generator=gen(1234))
        e = {"n_keys": len(sd), "keys_shapes": {k: list(v.shape) for k, v in sd.items()},
             "param_sum": float(sum(p.double().sum().item() for p in net.parameters())),
             "param_sums": {n_: float(p.double().sum().item()) for n_, p in net.named_parameters()},
             "in_shape": list(x.shape)}
        net.eval()
        with torch.no_grad():
            o = net(x)
        outs = o if isinstance(o, tuple) else (o,)
        e["eval"] = [summarize(t) for t in outs]
        net.train()
        xi = x.clone().requires_grad_(True)
        o = net(xi)
        outs = o if isinstance(o, tuple) else (o,)
        e["train"] = [summarize(t) for t in outs]
        tgt = [onehot_target((1, 2, s, s, s), 4321 + i, 0.2) for i in range(len(outs))]
        if len(outs) == 2:
            h = Holder(1.0, 1.0)
            PH.FlapRecWithShapePriorDoubleOut.comp_losses_metrics(h, outs, tgt, 0, 1)
        elif outs[0].shape[1] == 2:
            h = Holder(1.0, 1.0)
            PH.ProblemHandler.comp_losses_metrics(h, outs[0], tgt[0], 0, 1)
        else:                      # 3-channel raw classes: plain sum-of-squares probe
            h = Holder(0, 0)
            h.pt_loss = (outs[0] ** 2).mean()
        h.pt_loss.backward()
        e["loss"] = h.pt_loss.item()
        e["dx"] = summarize(xi.grad)
        e["grads"] = {n_: (summarize(p.grad) if p.grad is not None else None) for n_, p in net.named_parameters()}
        e["post_buffers"] = {n_: (summarize(b.float()) if b.numel() > 1 else float(b)) for n_, b in net.named_buffers()}
        out[name] = e
        print(name, e["n_keys"], e["param_sum"], e["loss"])
    with open(f"{HERE}/class_checksums.json", "w") as f:
        json.dump(out, f, indent=0)


def ini_params(ref, U, PH):
    out = {}
    for p in sorted(glob.glob(f"{REF}/examples/**/*.ini", recursive=True)):
        out[os.path.relpath(p, REF)] = U.set_cfg_params(p, {})
    with open(f"{HERE}/ini_params.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("ini", len(out))


if __name__ == "__main__":
    ref, U, PH = load_ref()
    if len(sys.argv) > 1 and sys.argv[1] == "skip_modes":      # only the fixtures added in round 2
        tiny_skip_modes(ref, U, PH)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stable":          # only the fixtures added in round 3
        mask_stable(ref, U, PH)
        sys.exit(0)
    tiny_generic(ref, U, PH)
    tiny_skip_modes(ref, U, PH)
    tiny_sp(ref, U, PH)
    tiny_legacy(ref, U, PH)
    mask_stable(ref, U, PH)
    losses(ref, U, PH)
    class_checksums(ref, U, PH)
    ini_params(ref, U, PH)
