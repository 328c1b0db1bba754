"""Patch tiling of whole skull volumes (BASELINE config 4: "skull volumes tiled to 192^3 patches").

The reference has no tiler: its datasets resize whole volumes to the network size on the CPU
(/root/reference/ctunet/pytorch/datasets.py:89-112,195-235).  What IS the contract is the sample dict those datasets
emit -- ``{"image": float32 [C,D,H,W], "target": one-hot [2,D,H,W] | (one-hot, one-hot), "filepath": str}`` -- and
``Model.forward_pass`` consuming batches of it (Model.py:342-349).  ``VolumeTiler`` cuts a sample of any size into
overlapping patches that carry the same schema (so the step runs on them unchanged) and stitches per-patch predictions
back into a volume (mean over overlaps).  Both directions are single gather kernels on the GPU
(``ctu_extract_patches`` / ``ctu_stitch_patches``); the start grid is plain host arithmetic.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple, Union

import torch

from . import ops


def tile_starts(size: int, patch: int, overlap: int) -> List[int]:
    """Start offsets along one axis: stride ``patch - overlap``, the last tile shifted back so that it ends at the
    volume border; a volume smaller than the patch gets one tile at 0 (zero padded by the extraction)."""
    if patch <= 0 or overlap < 0 or overlap >= patch:
        raise ValueError("tile_starts: need patch > overlap >= 0")
    if size <= patch:
        return [0]
    step = patch - overlap
    starts = list(range(0, size - patch, step)) + [size - patch]
    return starts


class VolumeTiler:
    def __init__(self, patch: Union[int, Sequence[int]] = 192, overlap: Union[int, Sequence[int]] = 32):
        self.patch = (patch,) * 3 if isinstance(patch, int) else tuple(int(p) for p in patch)
        self.overlap = (overlap,) * 3 if isinstance(overlap, int) else tuple(int(o) for o in overlap)
        if len(self.patch) != 3 or len(self.overlap) != 3:
            raise ValueError("VolumeTiler: patch / overlap are ints or 3-tuples")

    def coords(self, shape: Sequence[int], device="cuda") -> torch.Tensor:
        """int32 [P,3] (z0,y0,x0), z-major order."""
        zs, ys, xs = (tile_starts(int(s), p, o) for s, p, o in zip(shape, self.patch, self.overlap))
        return torch.tensor([(z, y, x) for z in zs for y in ys for x in xs], dtype=torch.int32, device=device)

    def extract(self, vol: torch.Tensor, coords: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """vol [C,D,H,W] -> (patches [P,C,pd,ph,pw], coords)."""
        if coords is None:
            coords = self.coords(vol.shape[1:], vol.device)
        return ops.extract_patches(vol, coords, self.patch), coords

    def stitch(self, patches: torch.Tensor, coords: torch.Tensor, shape: Sequence[int]) -> torch.Tensor:
        """patches [P,C,pd,ph,pw] -> [C,D,H,W] (mean over the patches that cover a voxel)."""
        return ops.stitch_patches(patches, coords, tuple(int(s) for s in shape))

    def split_sample(self, sample: Dict) -> List[Dict]:
        """One dataset sample -> the list of patch samples (same schema; ``coords`` and ``volume_shape`` added so that
        predictions can be stitched)."""
        img = sample["image"]
        coords = self.coords(img.shape[1:], img.device)
        ip, _ = self.extract(img, coords)
        tgt = sample.get("target")
        if tgt is None:
            tp = None
        elif isinstance(tgt, (tuple, list)):
            tp = [self.extract(t, coords)[0] for t in tgt]
        else:
            tp = self.extract(tgt, coords)[0]
        out = []
        for i in range(coords.shape[0]):
            s = {"image": ip[i], "filepath": f"{sample.get('filepath', '')}#patch{i}", "coords": coords[i],
                 "volume_shape": tuple(img.shape[1:])}
            if tp is not None:
                s["target"] = tuple(t[i] for t in tp) if isinstance(tp, list) else tp[i]
            out.append(s)
        return out
