"""CPU-side checks: the C-ABI library loads and exports every symbol include/ctunet_hip.h declares
(no compute calls -- there is no GPU here), the ctypes table mirrors the header, and the host logic
of the drop-in classes (names, state_dict, init, error behaviour, ini parsing) matches the reference."""
import ctypes
import json
import os
import re

import pytest
import torch

from util import load_npz, GOLDEN, load_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ctunet_hip.h")


def _header_decls():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"^\s*(?:const\s+char\*|int|size_t)\s+(ctu_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S | re.M):
        args = m.group(2).strip()
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_library_exports_every_declared_symbol():
    from ctunet_amd import _lib
    decls = _header_decls()
    assert len(decls) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in ctunet_hip.h but not exported"
    # the ctypes prototype table covers the header one to one, with matching arity
    assert set(_lib.SIGNATURES) == set(decls)
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name][1]) == n, name
    loaded = _lib.load()
    assert loaded.ctu_arch() == b"gfx950"
    assert loaded.ctu_abi_version() == _lib.ABI_VERSION
    # pure geometry helpers may be called without a GPU
    assert loaded.ctu_conv3d_packed_floats(3, 8, 8, 0) == 27 * 128
    assert loaded.ctu_conv3d_packed_floats(3, 8, 8, 1) == 36 * 128
    assert loaded.ctu_conv3d_packed_floats(7, 8, 8, 0) == 0
    assert loaded.ctu_conv3d_layout(3, 8, 128) == 1 and loaded.ctu_conv3d_layout(3, 16, 128) == 0
    assert loaded.ctu_conv3d_num_blocks(1, 128, 128, 128, 3, 16, 0) == 745     # persistent kernel: one row per block
    assert loaded.ctu_conv3d_num_blocks(1, 128, 128, 128, 3, 8, 1) == 512
    assert loaded.ctu_conv3d_num_blocks(1, 8, 8, 8, 5, 16, 0) == 8              # generic kernel: one row per box


def test_stale_library_is_refused(monkeypatch):
    """A libctunet_hip.so whose ABI version differs from the one _lib.py binds must fail at load, not misbehave later
    (signatures changed during the round: ctu_loss_bwd, ctu_bn_finalize, ctu_bn_bwd_finalize)."""
    from ctunet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1000)
    with pytest.raises(ImportError, match="stale build"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load().ctu_abi_version() == _lib.ABI_VERSION


def test_loss_total_adds_terms_without_an_int_zero():
    from ctunet_amd import losses
    a, b, c = torch.tensor(1.5), torch.tensor(2.0), torch.tensor(-0.25)
    assert losses._total([a]) is a
    assert float(losses._total([a, b, c])) == 3.25
    z = losses._total([])                # both lambdas 0: the reference's sum([]) == 0 (ProblemHandler.py:91)
    assert float(z) == 0.0 and z.dim() == 0


def test_scheduler_is_built_whenever_the_key_exists(monkeypatch):
    """Model.py:544-546: ``if 'scheduler' in self.params`` -- b_scheduler = False still gets a ReduceLROnPlateau."""
    from ctunet_amd import trainer
    import ctunet_amd

    class _Run(trainer.StepRunner):
        def __init__(self, params):
            self.params = dict(params)
            self.models = {"main": ctunet_amd.UNet(n_blocks=2, i_size=2)}
            self.initialize_optimizer()
    base = dict(optimizer="sgd", learning_rate=0.1, momentum=0.9, weight_decay=0.0)
    r = _Run(dict(base, scheduler=False))
    assert isinstance(r.params["scheduler"], torch.optim.lr_scheduler.ReduceLROnPlateau)
    assert "scheduler" not in _Run(base).params


def test_no_gpu_no_fallback():
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    import ctunet_amd
    from ctunet_amd import losses, ops
    net = ctunet_amd.UNet(n_blocks=2, i_size=2)
    with pytest.raises(RuntimeError, match="MI355X"):
        net(torch.zeros(1, 1, 8, 8, 8))
    with pytest.raises(RuntimeError):
        losses.dice_loss()(torch.zeros(1, 2, 4, 4, 4), torch.zeros(1, 2, 4, 4, 4))
    with pytest.raises(RuntimeError):
        ops.ncdhw_to_cl(torch.zeros(1, 1, 4, 4, 4))


def test_product_code_never_imports_oracle():
    """The oracle is test infrastructure: nothing under ct-unet_amd/ imports, links or executes it."""
    pkg = os.path.join(ROOT, "ct-unet_amd")
    pat = re.compile(r"(^|\n)\s*(from|import)\s+oracle|unet_oracle|oracle/|oracle\.")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                assert not pat.search(open(os.path.join(dp, f)).read()), (dp, f)


@pytest.mark.parametrize("name", ["UNet", "UNet4b2i3o", "UNet5b2i3o", "UNet4b1i3o", "UNetSP", "UNetSPSmall", "UNetDO",
                                  "recAE_v2_fixed", "UNet4_2IC"])
def test_state_dict_and_seeded_init_match_reference(name):
    import ctunet_amd
    exp = load_json("class_checksums.json")[name]
    torch.manual_seed(0)
    net = getattr(ctunet_amd, name)()
    sd = net.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == exp["keys_shapes"]
    assert len(sd) == exp["n_keys"]
    for n_, p in net.named_parameters():
        assert abs(p.double().sum().item() - exp["param_sums"][n_]) < 1e-9, n_
    # round trip incl. the DataParallel "module." prefix the reference's checkpoints may carry (Model.py:282,486)
    from ctunet_amd.checkpoint import load_state
    net2 = getattr(ctunet_amd, name)()
    load_state(net2, {"module." + k: v for k, v in sd.items()})
    for k, v in net2.state_dict().items():
        assert torch.equal(v, sd[k])


def test_unsupported_options_raise():
    import ctunet_amd
    for kw in (dict(residual=True), dict(fc_layer=[8, 4]), dict(dropout_p=0.5), dict(kern_sz_conv=7, padding=3),
               dict(out_channels=5)):
        with pytest.raises(NotImplementedError):
            ctunet_amd.UNet(**kw)


def test_skip_mode_state_dicts_match_reference_fixtures():
    """UNet(cat=False) / UNet(use_skip_connections=False): the channel plan (models.py:207-224) halves the decoder
    inputs and the head input; keys and shapes must equal the reference's state dict recorded in the fixtures."""
    import ctunet_amd
    for name, kw in (("tiny_unet_add.npz", dict(cat=False, apply_softmax=True)),
                     ("tiny_unet_noskip.npz", dict(use_skip_connections=False))):
        rec = load_npz(name)
        net = ctunet_amd.UNet(input_channels=1, out_channels=2, n_blocks=2, i_size=3, use_checkpoint=False, **kw)
        want = {k[3:]: v.shape for k, v in rec.items() if k.startswith("sd.")}
        got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert got == {k: tuple(s) for k, s in want.items()}


def test_ini_parser_matches_reference():
    from ctunet_amd.utilities import load_params, set_cfg_params
    exp = load_json("ini_params.json")
    ref_root = "/root/reference"
    if not os.path.isdir(ref_root):
        pytest.skip("reference tree not present (GPU box)")
    for rel, params in exp.items():
        got = set_cfg_params(os.path.join(ref_root, rel), {})
        assert json.loads(json.dumps(got)) == params, rel
        import ctunet_amd
        from ctunet_amd import ProblemHandler
        assert hasattr(ctunet_amd, got["model_class"])                 # s_model_class resolves to a drop-in
        assert hasattr(ProblemHandler, got["problem_handler"])         # s_problem_handler resolves
    with pytest.raises(FileNotFoundError):
        set_cfg_params("/nonexistent.ini", {})
    assert set_cfg_params(None) is None


def test_ini_parser_type_prefixes(tmp_path):
    from ctunet_amd.utilities import set_cfg_params
    p = tmp_path / "x.ini"
    p.write_text("[A]\ni_n = 3\nf_x = 0.5\nb_flag = True\ns_name = abc\nplain = 7\n[B]\ns_resume_model =\n")
    got = set_cfg_params(str(p), {"keep": 1})
    assert got == {"keep": 1, "n": 3, "x": 0.5, "flag": True, "name": "abc", "plain": "7", "resume_model": ""}


def test_inference_tail_has_no_cpu_fallback():
    """hard_segm_from_tensor / dice_coeff run as HIP kernels; CPU tensors are refused, not silently computed."""
    from ctunet_amd.utilities import dice_coeff, hard_segm_from_tensor
    p = torch.rand(2, 2, 4, 4, 4, generator=torch.Generator().manual_seed(0))
    with pytest.raises(RuntimeError):
        hard_segm_from_tensor(p)
    with pytest.raises(RuntimeError):
        dice_coeff(p, p)
