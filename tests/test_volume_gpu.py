"""Whole-volume helpers either side of the patch path: patch tiling + stitching, Hausdorff distance.

The tiler is integer / index work: bit-exact round trips.  The Hausdorff metric has no pin in the reference tree (monai is
un-vendored and absent: PARITY UNPINNED); it is compared with the same published definition restated on scipy.ndimage
(surface = mask & ~binary_erosion(mask), Euclidean distance transform of the other surface's complement, symmetric max)."""
import numpy as np
import pytest
import torch

from util import gen

pytestmark = pytest.mark.gpu


def test_tile_starts_cover_the_axis():
    from ctunet_amd.tiling import tile_starts
    assert tile_starts(128, 192, 32) == [0]
    assert tile_starts(192, 192, 32) == [0]
    assert tile_starts(304, 192, 32) == [0, 112]
    assert tile_starts(512, 192, 32) == [0, 160, 320]
    for size, patch, ov in [(224, 192, 32), (512, 192, 0), (97, 32, 5), (33, 32, 31)]:
        st = tile_starts(size, patch, ov)
        cov = np.zeros(size, bool)
        for a in st:
            assert 0 <= a and a + patch <= size
            cov[a:a + patch] = True
        assert cov.all() and st == sorted(set(st))
    with pytest.raises(ValueError):
        tile_starts(10, 4, 4)


@pytest.mark.parametrize("shape,patch,overlap", [((40, 56, 48), 32, 8), ((64, 64, 64), 32, 0), ((20, 33, 70), (16, 32, 32), (4, 8, 2)),
                                                 ((24, 24, 24), 32, 8)])
def test_extract_stitch_round_trip_is_bit_exact(shape, patch, overlap):
    from ctunet_amd.tiling import VolumeTiler
    t = VolumeTiler(patch, overlap)
    vol = (torch.rand((2,) + shape, generator=gen(1)) < 0.3).float().cuda()          # binary skull masks like the reference's
    patches, coords = t.extract(vol)
    pz = t.patch
    # every patch equals the window it was cut from (zero outside the volume)
    cpu = vol.cpu()
    for i, (z, y, x) in enumerate(coords.cpu().tolist()):
        win = torch.zeros((2,) + pz)
        sub = cpu[:, z:z + pz[0], y:y + pz[1], x:x + pz[2]]
        win[:, :sub.shape[1], :sub.shape[2], :sub.shape[3]] = sub
        assert torch.equal(patches[i].cpu(), win)
    back = t.stitch(patches, coords, shape)
    assert torch.equal(back, vol)                     # mean of identical 0/1 copies is exact
    # real-valued data: voxels covered once or by 2^k patches are exact, the others to 1 ulp of the mean
    volr = torch.randn((1,) + shape, generator=gen(2)).cuda()
    pr, _ = t.extract(volr, coords)
    backr = t.stitch(pr, coords, shape)
    assert torch.allclose(backr, volr, rtol=2e-7, atol=0)
    assert torch.equal(t.stitch(pr, coords, shape), backr)       # deterministic (fixed patch order, no atomics)


def test_split_sample_keeps_the_dataset_schema_and_stitches_predictions():
    """A 96x64x96 synthetic skull sample -> 32^3 patch samples with the reference datasets' schema; running the patches
    through a network and stitching gives a full-volume prediction; with overlap 0 the stitched argmax equals the
    per-patch argmax everywhere."""
    import ctunet_amd
    from ctunet_amd.tiling import VolumeTiler
    from ctunet_amd import ops
    g = gen(5)
    img = (torch.rand(2, 96, 64, 96, generator=g) < 0.2).float().cuda()
    lab = (torch.rand(1, 96, 64, 96, generator=g) < 0.3).float().cuda()
    oh = ops.one_hot(lab, 2)[0]
    t = VolumeTiler(32, 0)
    parts = t.split_sample({"image": img, "target": (oh, oh.clone()), "filepath": "synthetic://vol"})
    assert len(parts) == 3 * 2 * 3
    for p in parts:
        assert set(p) == {"image", "target", "filepath", "coords", "volume_shape"}
        assert p["image"].shape == (2, 32, 32, 32) and isinstance(p["target"], tuple) and p["target"][0].shape == (2, 32, 32, 32)
        assert torch.equal(p["target"][0].sum(0), torch.ones(32, 32, 32, device="cuda"))
    torch.manual_seed(0)
    net = ctunet_amd.UNetSP().cuda().eval()
    with torch.no_grad():
        preds = [net(p["image"].unsqueeze(0))[1][0] for p in parts]
    coords = torch.stack([p["coords"] for p in parts])
    full = t.stitch(torch.stack(preds), coords, img.shape[1:])
    assert full.shape == (2, 96, 64, 96)
    z, y, x = parts[7]["coords"].tolist()
    assert torch.equal(full[:, z:z + 32, y:y + 32, x:x + 32], preds[7])


def _hd_scipy(pred, target):
    from scipy import ndimage as ndi
    n, c = pred.shape[:2]
    hard = pred.argmax(1)
    out = np.full((n, c - 1), np.nan, np.float64)
    for i in range(n):
        for k in range(1, c):
            a, b = (hard[i] == k).numpy(), target[i, k].numpy() != 0
            ea, eb = a & ~ndi.binary_erosion(a), b & ~ndi.binary_erosion(b)
            if not ea.any() or not eb.any():
                continue
            da, db = ndi.distance_transform_edt(~ea), ndi.distance_transform_edt(~eb)
            out[i, k - 1] = max(db[ea].max(), da[eb].max())
    return out


@pytest.mark.parametrize("shape", [(2, 2, 24, 20, 28), (1, 3, 33, 17, 40)])
def test_hausdorff_matches_the_published_definition(shape):
    from ctunet_amd import ops
    from ctunet_amd.utilities import hausdorff
    g = gen(11)
    n, c, d, h, w = shape
    zz, yy, xx = torch.meshgrid(torch.arange(d), torch.arange(h), torch.arange(w), indexing="ij")
    pred = 0.05 * torch.rand(shape, generator=g)
    target = torch.zeros(shape)
    for i in range(n):
        for k in range(1, c):            # blobs: a ball in the prediction, a shifted box + a stray voxel in the target
            cz, cy, cx = d // 2 + k, h // 2 - k, w // 2 + 2 * i
            ball = ((zz - cz) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2) <= (4 + k) ** 2
            pred[i, k][ball] += 1.0
            target[i, k, cz - 3:cz + 5, 2:cy + 2, cx - 6:cx + 1] = 1.0
            target[i, k, d - 1, h - 1, w - 1] = 1.0       # a border voxel is its own surface (background outside)
    got = ops.hausdorff(pred.cuda(), target.cuda()).cpu().double().numpy()
    ref = _hd_scipy(pred, target)
    assert np.allclose(got, ref, rtol=1e-6, atol=0, equal_nan=True)
    assert abs(float(hausdorff(pred.cuda(), target.cuda())) - ref.mean()) < 1e-5
    # empty surfaces -> NaN from the kernel, max(shape) from the mirror of utilities.hausdorff (utilities.py:63,69)
    empty = torch.zeros(shape)
    empty[:, 0] = 1.0
    assert torch.isnan(ops.hausdorff(empty.cuda(), target.cuda())).all()
    assert float(hausdorff(empty.cuda(), target.cuda())) == float(max(shape))
    hard = torch.nn.functional.one_hot(pred.argmax(1), c).movedim(-1, 1).float().cuda()
    assert float(hausdorff(hard, hard)) == 0.0


def test_hausdorff_at_patch_size_is_deterministic_and_bounded():
    from ctunet_amd import ops
    from ctunet_amd.datasets import SyntheticFlapDataset
    s = SyntheticFlapDataset(2, size=128, seed=9)[0]
    full, flap = s["target"]
    a = ops.hausdorff(full.unsqueeze(0), flap.unsqueeze(0))
    b = ops.hausdorff(full.unsqueeze(0), flap.unsqueeze(0))
    assert torch.equal(a, b) and 0 < float(a) <= 128 * 3 ** 0.5
    assert float(ops.hausdorff(flap.unsqueeze(0), flap.unsqueeze(0))) == 0.0
