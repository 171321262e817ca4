"""Seeded synthetic inputs shared by the parity tests (CPU tensors)."""
import math

import torch


def small_scene(n=400, width=64, height=48, seed=0, spread=1.5, depth=6.0, log_scale=-1.5):
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(n, 3, generator=g) * spread
    means[:, 2] += depth
    quats = torch.randn(n, 4, generator=g)
    scales = torch.exp(torch.randn(n, 3, generator=g) * 0.4 + log_scale)
    opac = torch.sigmoid(torch.randn(n, 1, generator=g) * 1.5)
    shs = torch.randn(n, 16, 3, generator=g) * 0.3
    shs[:, 0] += 0.5
    ang = 0.2
    R = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    viewmat = torch.eye(4)
    viewmat[:3, :3] = R
    viewmat[:3, 3] = torch.tensor([0.1, -0.05, 0.3])
    f = 0.9 * width
    K = torch.tensor([[f, 0, width / 2.0], [0, f, height / 2.0], [0, 0, 1.0]])
    gt = (torch.rand(3, height, width, generator=g) * 255).to(torch.uint8)
    return dict(means=means, quats=quats, scales=scales, opac=opac, shs=shs, viewmat=viewmat, K=K,
                width=width, height=height, gt=gt)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def psnr(a, b):
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return 10 * math.log10(1.0 / max(mse, 1e-30))
