"""no_offload engine (reference: strategies/no_offload/engine.py:15-177): per-camera
render over all N Gaussians, gradient accumulation over the batch, one backward through the
activations."""
import math

import torch

from ... import utils
from ...densification import update_densification_stats_baseline_accum_grads
from ...gsplat import (fully_fused_projection, isect_offset_encode, isect_tiles,
                       rasterize_to_pixels, spherical_harmonics)
from ..base_engine import torch_compiled_loss


def baseline_accumGrads_micro_step(means3D, opacities, scales, rotations, shs, sh_degree, camera,
                                   background, mode="train", tile_size=16):
    """One camera: K from FoV, projection(N) -> SH(masked) -> +0.5 clamp -> tile binning ->
    rasterize -> [3,H,W].  Returns (image, means2D (grad retained), radii, None)."""
    args = utils.get_args()
    image_width, image_height = int(utils.get_img_width()), int(utils.get_img_height())
    fx = image_width / (2 * math.tan(camera.FoVx * 0.5))
    fy = image_height / (2 * math.tan(camera.FoVy * 0.5))
    K = torch.tensor([[fx, 0, image_width / 2.0], [0, fy, image_height / 2.0], [0, 0, 1]],
                     device=means3D.device)
    viewmat = camera.world_view_transform.transpose(0, 1)
    radiis, means2D, depths, conics, _ = fully_fused_projection(
        means=means3D, covars=None, quats=rotations, scales=scales, viewmats=viewmat.unsqueeze(0),
        Ks=K.unsqueeze(0), width=image_width, height=image_height, radius_clip=args.radius_clip,
        packed=False)
    if mode == "train":
        means2D.retain_grad()
    camtoworld = torch.inverse(viewmat.unsqueeze(0))
    dirs = means3D[None, :, :] - camtoworld[:, None, :3, 3]
    colors = spherical_harmonics(degrees_to_use=sh_degree, dirs=dirs, coeffs=shs.unsqueeze(0),
                                 masks=(radiis > 0))
    colors = torch.clamp_min(colors + 0.5, 0.0)
    tile_width = math.ceil(image_width / float(tile_size))
    tile_height = math.ceil(image_height / float(tile_size))
    _, isect_ids, flatten_ids = isect_tiles(means2d=means2D, radii=radiis, depths=depths,
                                            tile_size=tile_size, tile_width=tile_width,
                                            tile_height=tile_height, packed=False)
    isect_offsets = isect_offset_encode(isect_ids, 1, tile_width, tile_height)
    rendered_image, _ = rasterize_to_pixels(
        means2d=means2D, conics=conics, colors=colors, opacities=opacities.squeeze(1).unsqueeze(0),
        image_width=image_width, image_height=image_height, tile_size=tile_size,
        isect_offsets=isect_offsets, flatten_ids=flatten_ids, backgrounds=background)
    rendered_image = rendered_image.squeeze(0).permute(2, 0, 1)  # [3,H,W] view, no copy
    return rendered_image, means2D, radiis, None


def baseline_accumGrads_impl(gaussians, scene, batched_cameras, background, scaling_modifier=1.0,
                             sparse_adam=False):
    """Activations once per batch, detached leaves accumulate over the cameras, then one
    backward through the activations.  Returns (losses, visibility | None); .grad lands on the
    model's six parameters (the optimizer step is the caller's, train.py:533-578)."""
    if getattr(utils.get_args(), "fused_front_end", True) and scaling_modifier == 1.0:
        return _baseline_fused(gaussians, scene, batched_cameras, background, sparse_adam)
    losses = []
    means3D = gaussians.get_xyz
    opacities_origin = gaussians.get_opacity
    scales_origin = gaussians.get_scaling * scaling_modifier
    rotations_origin = gaussians.get_rotation
    shs_origin = gaussians.get_features
    sh_degree = gaussians.active_sh_degree
    opacities = opacities_origin.detach().requires_grad_(True)
    scales = scales_origin.detach().requires_grad_(True)
    rotations = rotations_origin.detach().requires_grad_(True)
    shs = shs_origin.detach().requires_grad_(True)
    visibility = (torch.zeros((means3D.shape[0],), dtype=torch.bool, device=means3D.device)
                  if sparse_adam else None)
    H, W = int(utils.get_img_height()), int(utils.get_img_width())
    for camera in batched_cameras:
        rendered_image, means2D, radiis, gaussian_ids = baseline_accumGrads_micro_step(
            means3D, opacities, scales, rotations, shs, sh_degree, camera, background)
        loss = torch_compiled_loss(rendered_image, camera.original_image)
        loss.backward()
        losses.append(loss.detach())
        with torch.no_grad():
            update_densification_stats_baseline_accum_grads(scene, gaussians, H, W, means2D.grad,
                                                            radiis, gaussian_ids)
        if sparse_adam:
            visibility = visibility | (radiis > 0).squeeze()
        del loss, rendered_image, means2D, radiis
    opacities_origin.backward(opacities.grad)
    scales_origin.backward(scales.grad)
    rotations_origin.backward(rotations.grad)
    shs_origin.backward(shs.grad)
    return losses, visibility


def _baseline_fused(gaussians, scene, batched_cameras, background, sparse_adam):
    """Same batch through the fused front end (clm_gs_amd/fused.py): every camera runs over all N
    rows (filter = None), raw-parameter gradients are accumulated in place (activation VJPs are
    inside the kernel, so there is no separate "backward to origin" step), statistics follow the
    radii > 0 mask form of densification.py:105-147."""
    from ...fused import train_one_camera
    dev = gaussians._xyz.device
    N = gaussians._xyz.shape[0]
    with torch.no_grad():
        shs = gaussians.get_features.reshape(N, 48).contiguous()  # one [N,48] view of dc | rest
        g_shs = torch.zeros_like(shs)
        for p in (gaussians._xyz, gaussians._opacity, gaussians._scaling, gaussians._rotation):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        visibility = torch.zeros((N,), dtype=torch.bool, device=dev) if sparse_adam else None
        losses = []
        for camera in batched_cameras:
            losses.append(train_one_camera(gaussians, camera, None, shs, 1, g_shs, background,
                                           camera.original_image, stats_only_visible=True,
                                           visibility_out=visibility))
        g3 = g_shs.reshape(N, 16, 3)
        for p, g in ((gaussians._features_dc, g3[:, :1, :]), (gaussians._features_rest, g3[:, 1:, :])):
            p.grad = g.contiguous() if p.grad is None else p.grad + g
    return losses, visibility
