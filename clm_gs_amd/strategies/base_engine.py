"""Raster front end shared by the engines (reference: strategies/base_engine.py:15-207):
calculate_filters, the L1+SSIM training loss, and the one-camera forward used by eval."""
import math

import torch

from .. import utils
from ..clm_kernels import fused_l1_ssim_loss, fused_ssim
from ..gsplat import (fully_fused_projection, isect_offset_encode, isect_tiles,
                      rasterize_to_pixels, spherical_harmonics, visibility_radii)

LAMBDA_DSSIM = 0.2
TILE_SIZE = 16


def select_filters(batched_cameras, xyz_gpu, scaling_raw_gpu, rotation_raw_gpu, block_flags=None):
    """calculate_filters on the stored (raw) parameters, selected entirely on the GPU
    (gsplat.visibility_select) -> (filters, touched_rows): the same index sets as calculate_filters
    plus the union over the batch's cameras."""
    from ..gsplat import visibility_select
    args = utils.get_args()
    with torch.no_grad():
        Ks = torch.stack([c.create_k_on_gpu() if getattr(c, "K", None) is None else c.K for c in batched_cameras])
        viewmats = torch.stack([c.world_view_transform.transpose(0, 1) for c in batched_cameras])
        filters, touched_rows = visibility_select(
            xyz_gpu, rotation_raw_gpu, scaling_raw_gpu, viewmats, Ks, int(utils.get_img_width()),
            int(utils.get_img_height()), radius_clip=args.radius_clip, block_flags=block_flags)
    assert all(f.numel() > 0 for f in filters), (
        "every camera must see at least one gaussian (base_engine.py:64-67)")
    return filters, touched_rows


def calculate_filters(batched_cameras, xyz_gpu, opacity_gpu, scaling_gpu, rotation_gpu,
                      return_ids=False, raw=False):
    """Per-camera visible index lists: ONE cull pass over (bsz cameras x N Gaussians).

    Same index sets as the packed projection the reference runs (base_engine.py:18-76) -- the
    cull is the projection's own (near plane, blur determinant, radius, off-screen) -- but only
    radii are written, and the (camera, gaussian) pairs are compacted with a single one-column
    nonzero over the flat [C*N] mask (camera boundaries by binary search on the sorted result).
    Returns (filters, camera_ids, gaussian_ids); the id vectors only when return_ids.
    raw=True: scaling_gpu / rotation_gpu are the stored parameters (log-scales, un-normalised
    quaternions) and the activations happen inside the cull kernel."""
    args = utils.get_args()
    with torch.no_grad():
        Ks = torch.stack([c.create_k_on_gpu() if getattr(c, "K", None) is None else c.K for c in batched_cameras])
        viewmats = torch.stack([c.world_view_transform.transpose(0, 1) for c in batched_cameras])
        radii = visibility_radii(xyz_gpu, rotation_gpu, scaling_gpu, viewmats, Ks,
                                 int(utils.get_img_width()), int(utils.get_img_height()),
                                 radius_clip=args.radius_clip, raw=raw)
        C, N = radii.shape
        flat = torch.nonzero(radii.reshape(-1) > 0).flatten()  # sorted: camera-major, then gaussian
        edges = torch.searchsorted(flat, torch.arange(0, C + 1, device=flat.device) * N)
        e = edges.tolist()  # the one host sync of the filter stage
        counts_cpu = [e[i + 1] - e[i] for i in range(C)]
        assert all(c > 0 for c in counts_cpu), (
            "every camera must see at least one gaussian (base_engine.py:64-67)")
        pieces = torch.split(flat, counts_cpu)
        filters = tuple(p - i * N if i else p for i, p in enumerate(pieces))
        camera_ids = gaussian_ids = None
        if return_ids:
            camera_ids = torch.div(flat, N, rounding_mode="floor")
            gaussian_ids = flat - camera_ids * N
    return filters, camera_ids, gaussian_ids


def loss_combined(image, image_gt, ssim_loss):
    Ll1 = torch.abs(image - image_gt).mean()
    return (1.0 - LAMBDA_DSSIM) * Ll1 + LAMBDA_DSSIM * (1.0 - ssim_loss)


def torch_compiled_loss(image, image_gt_original):
    """0.8 * L1 + 0.2 * (1 - SSIM) against clamp(u8/255) (base_engine.py:79-103).  The
    reference torch.compile's the L1 mix next to a fused SSIM kernel; here the whole loss
    (u8 -> float GT, L1, SSIM, mix) is one fused HIP kernel each way when the GT is uint8."""
    if image_gt_original.dtype == torch.uint8:
        return fused_l1_ssim_loss(image, image_gt_original, LAMBDA_DSSIM)
    image_gt = torch.clamp(image_gt_original / 255.0, 0.0, 1.0)
    ssim_loss = fused_ssim(image.unsqueeze(0), image_gt.unsqueeze(0))
    return loss_combined(image, image_gt, ssim_loss)


def _tile_counts(w, h):
    return math.ceil(w / float(TILE_SIZE)), math.ceil(h / float(TILE_SIZE))


def pipeline_forward_one_step(filtered_opacity_gpu, filtered_scaling_gpu, filtered_rotation_gpu,
                              filtered_xyz_gpu, filtered_shs, camera, scene, gaussians, background,
                              pipe_args, eval=False):
    """One camera over the gathered rows, SH through autograd (base_engine.py:106-207)."""
    image_width, image_height = int(utils.get_img_width()), int(utils.get_img_height())
    fx = image_width / (2 * math.tan(camera.FoVx * 0.5))
    fy = image_height / (2 * math.tan(camera.FoVy * 0.5))
    K = torch.tensor([[fx, 0, image_width / 2.0], [0, fy, image_height / 2.0], [0, 0, 1]],
                     device=filtered_xyz_gpu.device)
    viewmat = camera.world_view_transform.transpose(0, 1)
    n_selected = filtered_xyz_gpu.shape[0]
    radiis, means2D, depths, conics, _ = fully_fused_projection(
        means=filtered_xyz_gpu, covars=None, quats=filtered_rotation_gpu,
        scales=filtered_scaling_gpu, viewmats=viewmat.unsqueeze(0), Ks=K.unsqueeze(0),
        width=image_width, height=image_height, packed=False)
    if not eval:
        means2D.retain_grad()
    camtoworlds = torch.inverse(viewmat.unsqueeze(0))
    dirs = filtered_xyz_gpu[None, :, :] - camtoworlds[:, None, :3, 3]
    colors = spherical_harmonics(degrees_to_use=gaussians.active_sh_degree, dirs=dirs,
                                 coeffs=filtered_shs.reshape(1, n_selected, 16, 3))
    colors = torch.clamp_min(colors + 0.5, 0.0)
    opacities = filtered_opacity_gpu.squeeze(1).unsqueeze(0)
    tile_width, tile_height = _tile_counts(image_width, image_height)
    _, isect_ids, flatten_ids = isect_tiles(means2d=means2D, radii=radiis, depths=depths,
                                            tile_size=TILE_SIZE, tile_width=tile_width,
                                            tile_height=tile_height, packed=False)
    isect_offsets = isect_offset_encode(isect_ids, 1, tile_width, tile_height)
    backgrounds = background.reshape(1, 3) if background is not None else None
    rendered_image, _ = rasterize_to_pixels(
        means2d=means2D, conics=conics, colors=colors, opacities=opacities,
        image_width=image_width, image_height=image_height, tile_size=TILE_SIZE,
        isect_offsets=isect_offsets, flatten_ids=flatten_ids, backgrounds=backgrounds)
    rendered_image = rendered_image.squeeze(0).permute(2, 0, 1)  # [3,H,W] view, no copy
    return rendered_image, means2D, radiis
