"""Seeded synthetic scenes and cameras: there are no datasets on the GPU box.

Aerial slab of Gaussians with unit mean nearest-neighbour spacing, nadir pinhole cameras on a
lawn-mower path, ground-truth images rendered from a perturbed copy of the scene.
"""
import math

import torch

from .cameras import Camera


# spatial_lr_scale of the synthetic scenes.  The reference scales the position learning rate by the
# camera extent of the dataset (scene/__init__.py -> create_from_pcd(spatial_lr_scale=cameras_extent));
# for a slab of n Gaussians at unit spacing the geometric extent is 0.5*sqrt(n) (2646 at 28 M), which
# with position_lr_init 1.6e-4 and sqrt(bsz) scaling is a 0.85-unit Adam step on a scene whose nearest
# neighbour spacing is 1.0 -- the step size of iteration 0 of a 100 k-point model, not of a 28 M one.
# A 28 M-Gaussian model exists late in the reference's schedule (after densification, lr decayed);
# LR_EXTENT = 5 gives 1.6e-3 units per step, the regime in which the synthetic optimisation converges.
LR_EXTENT = 5.0


def perturbed_copy(scene, seed=99, xyz_sigma=0.05, opacity_shift=0.8, log_scale_shift=0.15, dc_shift=0.3):
    """The scene the ground-truth images are rendered from: positions jittered, and a SYSTEMATIC error in
    opacity / size / base colour shared by all Gaussians, so that every training step sees a consistent
    gradient and the loss of a converging optimisation falls within tens of batches."""
    g = torch.Generator(device=scene["xyz"].device).manual_seed(seed)
    out = dict(scene)
    out["xyz"] = scene["xyz"] + torch.randn(scene["xyz"].shape, generator=g, device=scene["xyz"].device) * xyz_sigma
    out["opacity"] = scene["opacity"] + opacity_shift
    out["scaling"] = scene["scaling"] + log_scale_shift
    shs = scene["shs48"].clone()
    shs[:, :3] += dc_shift
    out["shs48"] = shs
    return out


# Scale distributions of the synthetic scenes: (median scale in units of the nearest-neighbour spacing, sigma of the
# log-scale).  "slab" is SURVEY 8d's generator (I / V = 3.8 tile intersections per visible Gaussian at 4K); "heavy" is
# a heavy-tailed one chosen so that the MEASURED I / V is ~10, the figure SURVEY 8d's own Rubble-4K illustration uses
# (I = 30 M at V = 3 M): long per-tile lists, multi-round staging and deep early termination in the tile kernels.
SCENE_KINDS = {"slab": (0.7, 0.4), "heavy": (1.2, 0.75)}


def synth_gaussians(n, seed=0, device="cuda", thickness=0.1, kind="slab"):
    """xyz ~ U([-L,L]^2 x [0, thickness*L]) with L s.t. areal density = 1 / unit^2; log-scales
    ~ N(log s, sigma^2) with (s, sigma) = SCENE_KINDS[kind] ((0.7, 0.4) for the default slab); quats ~ N(0,I)
    un-normalised; opacity logit ~ N(0,1.5^2); SH dc ~ N(0,1), rest ~ N(0,0.1^2)."""
    s_med, s_sig = SCENE_KINDS[kind]
    g = torch.Generator(device=device).manual_seed(seed)
    L = 0.5 * math.sqrt(n)
    xyz = torch.rand((n, 3), generator=g, device=device)
    xyz[:, 0] = (xyz[:, 0] * 2 - 1) * L
    xyz[:, 1] = (xyz[:, 1] * 2 - 1) * L
    xyz[:, 2] = xyz[:, 2] * thickness * L
    scaling = torch.randn((n, 3), generator=g, device=device) * s_sig + math.log(s_med)
    rotation = torch.randn((n, 4), generator=g, device=device)
    opacity = torch.randn((n, 1), generator=g, device=device) * 1.5
    shs = torch.randn((n, 16, 3), generator=g, device=device) * 0.1
    shs[:, 0, :] = torch.randn((n, 3), generator=g, device=device)
    return dict(xyz=xyz, scaling=scaling, rotation=rotation, opacity=opacity,
                shs48=shs.reshape(n, 48), extent=L, lr_extent=LR_EXTENT)


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
