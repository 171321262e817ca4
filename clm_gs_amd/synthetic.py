"""Seeded synthetic scenes and cameras: there are no datasets on the GPU box.

Aerial slab of Gaussians with unit mean nearest-neighbour spacing, nadir pinhole cameras on a
lawn-mower path, ground-truth images rendered from a perturbed copy of the scene.
"""
import math

import torch

from .cameras import Camera


def synth_gaussians(n, seed=0, device="cuda", thickness=0.1):
    """xyz ~ U([-L,L]^2 x [0, thickness*L]) with L s.t. areal density = 1 / unit^2; log-scales
    ~ N(log 0.7, 0.4^2); quats ~ N(0,I) un-normalised; opacity logit ~ N(0,1.5^2);
    SH dc ~ N(0,1), rest ~ N(0,0.1^2)."""
    g = torch.Generator(device=device).manual_seed(seed)
    L = 0.5 * math.sqrt(n)
    xyz = torch.rand((n, 3), generator=g, device=device)
    xyz[:, 0] = (xyz[:, 0] * 2 - 1) * L
    xyz[:, 1] = (xyz[:, 1] * 2 - 1) * L
    xyz[:, 2] = xyz[:, 2] * thickness * L
    scaling = torch.randn((n, 3), generator=g, device=device) * 0.4 + math.log(0.7)
    rotation = torch.randn((n, 4), generator=g, device=device)
    opacity = torch.randn((n, 1), generator=g, device=device) * 1.5
    shs = torch.randn((n, 16, 3), generator=g, device=device) * 0.1
    shs[:, 0, :] = torch.randn((n, 3), generator=g, device=device)
    return dict(xyz=xyz, scaling=scaling, rotation=rotation, opacity=opacity,
                shs48=shs.reshape(n, 48), extent=L)


def nadir_cameras(n_cams, n_gaussians, width, height, visible_fraction, seed=0, device="cuda",
                  thickness=0.1):
    """Cameras looking straight down (-z in world = +z in camera) from a height chosen so the
    image footprint covers `visible_fraction` of the slab's area; centres on a lawn-mower path."""
    L = 0.5 * math.sqrt(n_gaussians)
    f = 0.8 * width
    fovx = 2 * math.atan(width / (2 * f))
    fovy = 2 * math.atan(height / (2 * f))
    area = visible_fraction * (2 * L) ** 2
    # footprint = (W/f * h) x (H/f * h)
    h = math.sqrt(area * f * f / (width * height))
    half_x, half_y = 0.5 * width / f * h, 0.5 * height / f * h
    g = torch.Generator().manual_seed(seed + 1)
    cams = []
    cols = max(1, int(math.ceil(math.sqrt(n_cams))))
    for i in range(n_cams):
        r, c = divmod(i, cols)
        if r % 2:
            c = cols - 1 - c
        u = (c + 0.5) / cols
        v = (r + 0.5) / max(1, math.ceil(n_cams / cols))
        cx = -L + half_x + u * max(0.0, 2 * (L - half_x)) + float(torch.randn((), generator=g)) * 0.01 * L
        cy = -L + half_y + v * max(0.0, 2 * (L - half_y)) + float(torch.randn((), generator=g)) * 0.01 * L
        cz = thickness * L + h
        # camera axes in world: x_c = +x, y_c = -y, z_c = -z  (looking down)
        R = torch.tensor([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])
        C = torch.tensor([cx, cy, cz])
        w2c = torch.eye(4)
        w2c[:3, :3] = R
        w2c[:3, 3] = -R @ C
        cams.append(Camera(i, w2c, fovx, fovy, width, height, device=device))
    return cams
