"""Seeded synthetic calibration inputs and random "pretrained" networks (neither repo ships data
or checkpoints).  Generated on the CPU with torch.Generator so that every machine
and every rank sees the same volumes."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def brats_volume(seed: int, size=128, nmod=4) -> torch.Tensor:
    """BraTS-shaped volume (nmod x size^3): box-smoothed Gaussian noise inside a centred ellipsoid,
    exactly 0 outside (body mask = data[:,0] != 0, ptqer.py:337-338), standardised inside."""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(nmod, size, size, size, generator=g)
    v = F.avg_pool3d(v[None], 3, 1, 1, count_include_pad=False)[0]
    ax = (torch.arange(size, dtype=torch.float32) - (size - 1) / 2) / size
    r = (ax[:, None, None] / 0.40) ** 2 + (ax[None, :, None] / 0.45) ** 2 + (ax[None, None, :] / 0.38) ** 2
    m = (r < 1.0)
    for c in range(nmod):
        inside = v[c][m]
        v[c] = torch.where(m, (v[c] - inside.mean()) / inside.std(), torch.zeros(()))
    # keep exact zeros only outside the mask (inside voxels that standardise to exactly 0 are measure-zero)
    return v


def lits_volume(seed: int, size=160) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    sz = (size,) * 3 if isinstance(size, int) else tuple(size)
    v = torch.randn(1, *sz, generator=g)
    v = F.avg_pool3d(v[None], 3, 1, 1, count_include_pad=False)[0]
    return (v - v.mean()) / v.std()


def calib_batch(task: str, ids, size) -> torch.Tensor:
    if task == 'brats':
        return torch.stack([brats_volume(1000 + i, size) for i in ids])
    return torch.stack([lits_volume(2000 + i, size) for i in ids])


def randomise_network(model: nn.Module, seed: int = 0):
    """Kaiming-normal convs + non-trivial BN statistics, so BN folding gives every conv a bias."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.Conv3d):
                fan = m.weight[0].numel()
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan) ** 0.5)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
            if isinstance(m, nn.BatchNorm3d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return model
