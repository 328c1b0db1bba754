"""Synthetic patch source with the reference datasets' sample schema.

The reference's datasets read NIfTI volumes with SimpleITK and cut random flaps with raster_geometry on the CPU
(/root/reference/ctunet/pytorch/datasets.py:89-112,195-235; transforms.py) -- I/O and augmentation are out of scope.
What the training step consumes is only their OUTPUT contract, reproduced here for synthetic binary skulls generated
on the GPU:

    sample = {"image":  float32 [C, D, H, W]   skull with the flap removed (+ atlas channel when append_atlas),
              "target": float32 one-hot [2, D, H, W]                      (single-output handlers), or
                        (full_skull one-hot, flap one-hot)                 (FlapRec...DoubleOut handlers),
              "filepath": str}

with ``one_hot(label.long(), 2).movedim(-1, 0).float()`` targets (datasets.py:107-110, 212-217; done by
``ctu_one_hot``) and ``full_skull = image + flap`` (datasets.py:228).  ``torch.utils.data.DataLoader``'s default
collate turns the target tuple into the list ``Model.forward_pass`` expects (Model.py:344-349).
Inputs are binary masks cast to float like the reference's (datasets.py:92-94): a hollow ellipsoid shell ("skull"),
a spherical bite out of it ("flap"); geometry is drawn from ``torch.Generator(seed + idx)`` so every rank / epoch can
address its own deterministic items (rank r takes idx = r, r + world, ...).
"""
from __future__ import annotations

import torch
from torch.utils.data import Dataset

from . import ops


class SyntheticFlapDataset(Dataset):
    def __init__(self, n_items: int, size: int = 128, seed: int = 1234, double_out: bool = True,
                 append_atlas: bool = True, device="cuda"):
        if size % 16:
            raise ValueError("ctunet_amd: patch size must be divisible by 16 (4 pooling levels)")
        self.n, self.size, self.seed = int(n_items), int(size), int(seed)
        self.double_out, self.append_atlas = bool(double_out), bool(append_atlas)
        self.device = torch.device(device)
        ax = torch.linspace(-1.0, 1.0, self.size, device=self.device)
        self._zz, self._yy, self._xx = torch.meshgrid(ax, ax, ax, indexing="ij")
        self._atlas = self._shell(torch.tensor([0.0, 0.0, 0.0]), torch.tensor([0.72, 0.8, 0.66]), 0.07)

    def __len__(self):
        return self.n

    def _shell(self, centre, radii, thick):
        c, r = centre.tolist(), radii.tolist()
        q = ((self._zz - c[0]) / r[0]) ** 2 + ((self._yy - c[1]) / r[1]) ** 2 + ((self._xx - c[2]) / r[2]) ** 2
        return ((q <= 1.0) & (q >= (1.0 - thick / min(r)) ** 2)).float()

    def __getitem__(self, idx: int):
        if not 0 <= idx < self.n:
            raise IndexError(idx)
        g = torch.Generator().manual_seed(self.seed + idx)
        u = torch.rand(10, generator=g)
        centre = (u[0:3] - 0.5) * 0.16
        radii = torch.tensor([0.72, 0.8, 0.66]) * (0.9 + 0.2 * u[3:6])
        skull = self._shell(centre, radii, 0.06 + 0.04 * float(u[6]))
        # flap: the part of the shell inside a sphere centred on the upper half of the shell surface
        th, ph = float(u[7]) * 6.2832, 0.25 + 0.9 * float(u[8])
        fc = centre + radii * torch.tensor([torch.cos(torch.tensor(ph)), torch.sin(torch.tensor(ph)) * torch.sin(torch.tensor(th)),
                                            torch.sin(torch.tensor(ph)) * torch.cos(torch.tensor(th))])
        fr = 0.22 + 0.2 * float(u[9])
        fcl = fc.tolist()
        ball = ((self._zz - fcl[0]) ** 2 + (self._yy - fcl[1]) ** 2 + (self._xx - fcl[2]) ** 2) <= fr * fr
        flap = skull * ball.float()
        image = skull - flap                                        # binary {0, 1}, float
        full = ops.one_hot((image + flap).unsqueeze(0).contiguous(), 2)[0]          # datasets.py:228-230
        flap_oh = ops.one_hot(flap.unsqueeze(0).contiguous(), 2)[0]
        img = image.unsqueeze(0)
        if self.append_atlas:                                       # load_atlas_and_append_at_axis(image, 0)
            img = torch.cat((img, self._atlas.unsqueeze(0)), 0)
        target = (full, flap_oh) if self.double_out else flap_oh
        return {"image": img.contiguous(), "target": target, "filepath": f"synthetic://{self.seed}/{idx}"}
