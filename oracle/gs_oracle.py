"""CPU oracle (torch, any float dtype) for the CLM-GS rasterization hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``clm_gs_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use ``oracle/`` -- and only as the checker.

PARITY UNPINNED at the native-kernel boundary: the reference's arithmetic for
this path lives in third-party CUDA submodules that are *absent* from
/root/reference (empty dirs, ``.gitmodules:1-15``): gsplat (only pin: permalink
commit b60e917c95afc449c5be33a634f1f457e116ff5e, ``strategies/no_offload/README.md:15``),
clm_kernels, cpu-adam, fast-tsp, simple-knn (unpinned).  The reference has no
tests and no golden vectors for those kernels.  This file restates their
*published* algorithms and is pinned where the reference
tree allows it:
  * SH basis           -> utils/sh_utils.py:57-103 (eval_sh), golden fixture
  * SSIM / L1 / PSNR   -> utils/loss_utils.py:18-85, utils/image_utils.py:19-21
  * quat -> R, R S     -> utils/general_utils.py:311-346
  * camera conventions -> utils/graphics_utils.py:42-84, scene/cameras.py:87-126
  * engine call order  -> the reference's own strategies/no_offload/engine.py:15-177,
                          strategies/clm_offload/engine.py:30-979, base_engine.calculate_filters,
                          densification.py and both GaussianModels executed in the build container
                          with this oracle plugged in as `gsplat` / `clm_kernels` (+ oracle/clm_oracle.py
                          as cpu_adam / fast_tsp): tests/golden/make_engine_golden.py ->
                          tests/golden/engine_*.npz.  That pins the ORCHESTRATION (what is called
                          with what, accumulation, optimizer scaling, densification); the arithmetic
                          of projection / tile intersection / alpha blend / Adam stays pinned to
                          nothing but the published algorithms (no reference-produced vector exists).

Every function is differentiable through torch autograd; autograd of this
restatement is the gradient oracle for the hand-written HIP backward kernels.
"""

import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# A1: projection   (call sites: strategies/base_engine.py:36-47,139-151;
#                   strategies/no_offload/engine.py:49-60;
#                   strategies/clm_offload/engine.py:51-63)
# ----------------------------------------------------------------------------


def quat_to_rotmat(quats):
    """(w,x,y,z) -> R, normalising first.  Same matrix as
    utils/general_utils.py:311-334 (build_rotation)."""
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z),
            2 * (x * y - w * z),
            2 * (x * z + w * y),
            2 * (x * y + w * z),
            1 - 2 * (x * x + z * z),
            2 * (y * z - w * x),
            2 * (x * z - w * y),
            2 * (y * z + w * x),
            1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.reshape(quats.shape[:-1] + (3, 3))


def quat_scale_to_covar(quats, scales):
    """Sigma = R diag(s^2) R^T  (utils/general_utils.py:337-346 gives L = R S)."""
    R = quat_to_rotmat(quats)
    M = R * scales[..., None, :]
    return M @ M.transpose(-1, -2)


def fully_fused_projection(
    means,
    covars,
    quats,
    scales,
    viewmats,
    Ks,
    width,
    height,
    eps2d=0.3,
    near_plane=0.01,
    far_plane=1e10,
    radius_clip=0.0,
    packed=False,
):
    """EWA projection of N Gaussians into C pinhole cameras.

    Unpacked: radii[C,N] i32, means2d[C,N,2], depths[C,N], conics[C,N,3], None.
    Culled entries have radii == 0 and all-zero float outputs.
    Packed: (camera_ids, gaussian_ids, radii, means2d, depths, conics, None),
    sorted by (camera, gaussian) -- base_engine.py:64-73 relies on that order.
    """
    assert covars is None
    C = viewmats.shape[0]
    N = means.shape[0]
    dt = means.dtype
    Rv = viewmats[:, :3, :3]  # [C,3,3]
    tv = viewmats[:, :3, 3]  # [C,3]
    mean_c = torch.einsum("cij,nj->cni", Rv, means) + tv[:, None, :]  # [C,N,3]
    z_raw = mean_c[..., 2]
    ok_z = (z_raw >= near_plane) & (z_raw <= far_plane)
    # benign substitute so culled rows never produce inf/nan in autograd
    safe = torch.where(ok_z[..., None], mean_c, torch.tensor([0.0, 0.0, 1.0], dtype=dt))
    x, y, z = safe.unbind(-1)

    covar = quat_scale_to_covar(quats, scales)  # [N,3,3]
    covar_c = torch.einsum("cij,njk,clk->cnil", Rv, covar, Rv)  # [C,N,3,3]

    fx = Ks[:, 0, 0][:, None]
    fy = Ks[:, 1, 1][:, None]
    cx = Ks[:, 0, 2][:, None]
    cy = Ks[:, 1, 2][:, None]
    tan_fovx = 0.5 * width / fx
    tan_fovy = 0.5 * height / fy
    lim_x_pos = (width - cx) / fx + 0.3 * tan_fovx
    lim_x_neg = cx / fx + 0.3 * tan_fovx
    lim_y_pos = (height - cy) / fy + 0.3 * tan_fovy
    lim_y_neg = cy / fy + 0.3 * tan_fovy
    rz = 1.0 / z
    rz2 = rz * rz
    tx = z * torch.minimum(lim_x_pos, torch.maximum(-lim_x_neg, x * rz))
    ty = z * torch.minimum(lim_y_pos, torch.maximum(-lim_y_neg, y * rz))
    zero = torch.zeros_like(z)
    J = torch.stack(
        [fx * rz, zero, -fx * tx * rz2, zero, fy * rz, -fy * ty * rz2], dim=-1
    ).reshape(C, N, 2, 3)
    cov2d = J @ covar_c @ J.transpose(-1, -2)  # [C,N,2,2]
    mu_x = fx * x * rz + cx
    mu_y = fy * y * rz + cy

    c00 = cov2d[..., 0, 0] + eps2d
    c01 = cov2d[..., 0, 1]
    c11 = cov2d[..., 1, 1] + eps2d
    det = c00 * c11 - c01 * c01
    ok_det = det > 0
    det_s = torch.where(ok_det, det, torch.ones_like(det))
    conic = torch.stack([c11 / det_s, -c01 / det_s, c00 / det_s], dim=-1)
    b = 0.5 * (c00 + c11)
    v1 = b + torch.sqrt(torch.clamp(b * b - det, min=0.01))
    radius = torch.ceil(3.0 * torch.sqrt(v1)).detach()
    ok_r = radius > radius_clip
    ok_img = ~(
        (mu_x + radius <= 0)
        | (mu_x - radius >= width)
        | (mu_y + radius <= 0)
        | (mu_y - radius >= height)
    )
    valid = ok_z & ok_det & ok_r & ok_img
    radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)
    means2d = torch.stack([mu_x, mu_y], dim=-1) * valid[..., None]
    depths = z * valid
    conics = conic * valid[..., None]
    if not packed:
        return radii, means2d, depths, conics, None
    cam_ids, g_ids = torch.nonzero(valid, as_tuple=True)
    return (
        cam_ids,
        g_ids,
        radii[cam_ids, g_ids],
        means2d[cam_ids, g_ids],
        depths[cam_ids, g_ids],
        conics[cam_ids, g_ids],
        None,
    )


# ----------------------------------------------------------------------------
# A2: spherical harmonics  (basis constants: utils/sh_utils.py:26-43, 57-103)
# ----------------------------------------------------------------------------

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [
    1.0925484305920792,
    -1.0925484305920792,
    0.31539156525252005,
    -1.0925484305920792,
    0.5462742152960396,
]
SH_C3 = [
    -0.5900435899266435,
    2.890611442640554,
    -0.4570457994644658,
    0.3731763325901154,
    -0.4570457994644658,
    1.445305721320277,
    -0.5900435899266435,
]


def sh_basis(degree, dirs):
    """[..., (degree+1)^2] real SH basis on normalised dirs, the polynomial
    set of utils/sh_utils.py:73-103 with the signs folded into the basis."""
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    out = [torch.full_like(x, SH_C0)]
    if degree > 0:
        out += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if degree > 1:
        xx, yy, zz = x * x, y * y, z * z
        xy, yz, xz = x * y, y * z, x * z
        out += [
            SH_C2[0] * xy,
            SH_C2[1] * yz,
            SH_C2[2] * (2.0 * zz - xx - yy),
            SH_C2[3] * xz,
            SH_C2[4] * (xx - yy),
        ]
    if degree > 2:
        out += [
            SH_C3[0] * y * (3 * xx - yy),
            SH_C3[1] * xy * z,
            SH_C3[2] * y * (4 * zz - xx - yy),
            SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
            SH_C3[4] * x * (4 * zz - xx - yy),
            SH_C3[5] * z * (xx - yy),
            SH_C3[6] * x * (xx - 3 * yy),
        ]
    return torch.stack(out, dim=-1)


def spherical_harmonics(degrees_to_use, dirs, coeffs, masks=None):
    """colors[...,3] = sum_k basis_k(dirs/|dirs|) * coeffs[...,k,:]; no +0.5
    (the caller adds it: strategies/base_engine.py:164).  coeffs [...,K,3]."""
    nb = (degrees_to_use + 1) ** 2
    B = sh_basis(degrees_to_use, dirs)  # [..., nb]
    col = (B[..., :, None] * coeffs[..., :nb, :]).sum(dim=-2)
    if masks is not None:
        col = col * masks[..., None]
    return col


# ----------------------------------------------------------------------------
# A3/A4: tile binning   (call sites: base_engine.py:175-186)
# ----------------------------------------------------------------------------


def isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, packed=False):
    """-> tiles_per_gauss[C,N] i32, isect_ids[I] i64 (sorted), flatten_ids[I] i32.

    key = (cam << tile_bits | tile_id) << 32 | float_bits(depth); stable sort,
    ties keep (camera, gaussian, tile-row-major) emit order."""
    assert not packed
    C, N = radii.shape
    mu = means2d.detach().to(torch.float32)
    r = radii.to(torch.float32)
    tr = r / tile_size
    t = mu / tile_size
    minx = torch.clamp(torch.floor(t[..., 0] - tr), 0, tile_width).to(torch.int64)
    miny = torch.clamp(torch.floor(t[..., 1] - tr), 0, tile_height).to(torch.int64)
    maxx = torch.clamp(torch.ceil(t[..., 0] + tr), 0, tile_width).to(torch.int64)
    maxy = torch.clamp(torch.ceil(t[..., 1] + tr), 0, tile_height).to(torch.int64)
    cnt = (maxx - minx) * (maxy - miny)
    cnt = torch.where(radii > 0, cnt, torch.zeros_like(cnt))
    tiles_per_gauss = cnt.to(torch.int32)
    n_tiles = tile_width * tile_height
    tile_bits = int(math.floor(math.log2(n_tiles))) + 1
    depth_bits = depths.detach().to(torch.float32).contiguous().view(torch.int32).to(torch.int64)
    keys, vals = [], []
    cntf = cnt.flatten().tolist()
    minxf, minyf, maxxf, maxyf = (
        minx.flatten().tolist(),
        miny.flatten().tolist(),
        maxx.flatten().tolist(),
        maxy.flatten().tolist(),
    )
    dbf = depth_bits.flatten().tolist()
    for flat in range(C * N):
        if cntf[flat] == 0:
            continue
        cam = flat // N
        for ty in range(minyf[flat], maxyf[flat]):
            for tx in range(minxf[flat], maxxf[flat]):
                tile = ty * tile_width + tx
                keys.append((((cam << tile_bits) | tile) << 32) | (dbf[flat] & 0xFFFFFFFF))
                vals.append(flat)
    keys_t = torch.tensor(keys, dtype=torch.int64)
    vals_t = torch.tensor(vals, dtype=torch.int32)
    if keys_t.numel():
        order = torch.sort(keys_t, stable=True).indices
        keys_t, vals_t = keys_t[order], vals_t[order]
    return tiles_per_gauss, keys_t, vals_t


def isect_offset_encode(isect_ids, n_cameras, tile_width, tile_height):
    """offsets[C,th,tw] i32: first sorted index whose (cam,tile) >= that tile."""
    n_tiles = tile_width * tile_height
    tile_bits = int(math.floor(math.log2(n_tiles))) + 1
    ids = isect_ids >> 32
    cam = ids >> tile_bits
    tile = ids & ((1 << tile_bits) - 1)
    lin = cam * n_tiles + tile
    q = torch.arange(n_cameras * n_tiles, dtype=torch.int64)
    off = torch.searchsorted(lin.contiguous(), q, right=False)
    return off.to(torch.int32).reshape(n_cameras, tile_height, tile_width)


# ----------------------------------------------------------------------------
# A5/A6: rasterize   (call sites: base_engine.py:192-203)
# ----------------------------------------------------------------------------


def rasterize_to_pixels(
    means2d,
    conics,
    colors,
    opacities,
    image_width,
    image_height,
    tile_size,
    isect_offsets,
    flatten_ids,
    backgrounds=None,
    return_last_ids=False,
):
    """Front-to-back alpha blend per 16x16 tile.  -> image[C,H,W,3], alpha[C,H,W,1].

    Per pixel, walking the tile's depth-sorted list:
      sigma = .5(a dx^2 + c dy^2) + b dx dy, dx = mu - (j+.5, i+.5)
      alpha = min(.999, o * exp(-sigma)); skip if sigma < 0 or alpha < 1/255
      stop *before* the Gaussian whose T(1-alpha) <= 1e-4
    Vectorised per tile with cumprod; autograd supplies the backward (A6).
    """
    C, N = opacities.shape
    dt = means2d.dtype
    th, tw = isect_offsets.shape[1:]
    n_isects = flatten_ids.shape[0]
    off = isect_offsets.flatten().tolist() + [n_isects]
    m2 = means2d.reshape(C * N, 2)
    cn = conics.reshape(C * N, 3)
    col = colors.reshape(C * N, -1)
    op = opacities.reshape(C * N)
    fid = flatten_ids.to(torch.int64)
    H, W = image_height, image_width
    img = torch.zeros(C, th * tile_size, tw * tile_size, 3, dtype=dt)
    if backgrounds is not None:  # empty tiles: T = 1 -> pure background
        img = img + backgrounds.to(dt)[:, None, None, :]
    alp = torch.zeros(C, th * tile_size, tw * tile_size, 1, dtype=dt)
    last = torch.zeros(C, th * tile_size, tw * tile_size, dtype=torch.int32)
    tile_rows, tile_cols = [], []
    ii, jj = torch.meshgrid(
        torch.arange(tile_size), torch.arange(tile_size), indexing="ij"
    )
    for c in range(C):
        for ty in range(th):
            for tx in range(tw):
                tid = (c * th + ty) * tw + tx
                s, e = off[tid], off[tid + 1]
                y0, x0 = ty * tile_size, tx * tile_size
                if e <= s:
                    continue
                g = fid[s:e]
                px = (x0 + jj.reshape(-1)).to(dt) + 0.5  # [256]
                py = (y0 + ii.reshape(-1)).to(dt) + 0.5
                dx = m2[g, 0][None, :] - px[:, None]  # [256,K]
                dy = m2[g, 1][None, :] - py[:, None]
                a_, b_, c_ = cn[g, 0][None], cn[g, 1][None], cn[g, 2][None]
                sigma = 0.5 * (a_ * dx * dx + c_ * dy * dy) + b_ * dx * dy
                alpha = torch.clamp(op[g][None] * torch.exp(-sigma), max=0.999)
                valid = (sigma.detach() >= 0) & (alpha.detach() >= 1.0 / 255.0)
                a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
                one_m = 1.0 - a_eff
                incl = torch.cumprod(one_m, dim=1)  # next_T after k
                done = (incl.detach() <= 1e-4) & valid
                # first k at which the pixel terminates -> exclude k and later
                anyd = done.any(dim=1)
                first = torch.where(
                    anyd, done.to(torch.int64).argmax(dim=1), torch.full_like(anyd, e - s, dtype=torch.int64)
                )
                keep = torch.arange(e - s)[None, :] < first[:, None]
                a_eff = a_eff * keep
                one_m = 1.0 - a_eff
                incl = torch.cumprod(one_m, dim=1)
                T_before = torch.cat([torch.ones(incl.shape[0], 1, dtype=dt), incl[:, :-1]], dim=1)
                wgt = a_eff * T_before  # [256,K]
                rgb = wgt @ col[g]  # [256,3]
                T_fin = incl[:, -1]
                contrib = (valid & keep)
                # absolute index in the sorted isect list of the last contributor
                kidx = torch.arange(e - s)[None, :].expand_as(contrib)
                lastk = torch.where(contrib, kidx, torch.full_like(kidx, -1)).max(dim=1).values
                lastabs = torch.where(lastk >= 0, lastk + s, torch.zeros_like(lastk))
                if backgrounds is not None:
                    rgb = rgb + T_fin[:, None] * backgrounds[c][None, :]
                img[c, y0 : y0 + tile_size, x0 : x0 + tile_size] = rgb.reshape(tile_size, tile_size, 3)
                alp[c, y0 : y0 + tile_size, x0 : x0 + tile_size, 0] = (1.0 - T_fin).reshape(tile_size, tile_size)
                last[c, y0 : y0 + tile_size, x0 : x0 + tile_size] = lastabs.reshape(tile_size, tile_size).to(torch.int32)
    img = img[:, :H, :W].contiguous()
    alp = alp[:, :H, :W].contiguous()
    if return_last_ids:
        return img, alp, last[:, :H, :W].contiguous()
    return img, alp


# ----------------------------------------------------------------------------
# A7: SSIM + loss   (utils/loss_utils.py:18-85; strategies/base_engine.py:79-103)
# ----------------------------------------------------------------------------


def _ssim_window(dtype):
    g = torch.tensor(
        [math.exp(-((x - 5) ** 2) / (2 * 1.5**2)) for x in range(11)], dtype=torch.float64
    )
    g = g / g.sum()
    return (g[:, None] * g[None, :]).to(dtype)


def fused_ssim(img1, img2):
    """Mean SSIM over [B,3,H,W]; 11x11 sigma=1.5 zero-padded window,
    C1=0.01^2, C2=0.03^2 (utils/loss_utils.py:26-85)."""
    ch = img1.shape[1]
    w = _ssim_window(img1.dtype)[None, None].expand(ch, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t, w, padding=5, groups=ch)
    mu1, mu2 = conv(img1), conv(img2)
    s11 = conv(img1 * img1) - mu1 * mu1
    s22 = conv(img2 * img2) - mu2 * mu2
    s12 = conv(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01**2, 0.03**2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean()


def training_loss(image, gt_u8, lambda_dssim=0.2):
    """strategies/base_engine.py:79-103: gt = clamp(u8/255); 0.8 L1 + 0.2 (1-ssim)."""
    gt = torch.clamp(gt_u8.to(image.dtype) / 255.0, 0.0, 1.0)
    ssim = fused_ssim(image[None], gt[None])
    l1 = (image - gt).abs().mean()
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim)


def psnr(img1, img2):
    """utils/image_utils.py:19-21."""
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


# ----------------------------------------------------------------------------
# one camera end to end, as strategies/no_offload/engine.py:15-101 composes it
# ----------------------------------------------------------------------------


def render_one_camera(
    means3D, opacities, scales, rotations, shs, sh_degree, viewmat, K, width, height,
    background=None, tile_size=16, radius_clip=0.0, sh_mask=True,
):
    radii, means2d, depths, conics, _ = fully_fused_projection(
        means3D, None, rotations, scales, viewmat[None], K[None], width, height,
        radius_clip=radius_clip,
    )
    camtoworld = torch.inverse(viewmat[None])
    dirs = means3D[None] - camtoworld[:, None, :3, 3]
    colors = spherical_harmonics(
        sh_degree, dirs, shs[None], masks=(radii > 0) if sh_mask else None
    )
    colors = torch.clamp_min(colors + 0.5, 0.0)
    tw = math.ceil(width / float(tile_size))
    th = math.ceil(height / float(tile_size))
    _, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size, tw, th)
    offsets = isect_offset_encode(isect_ids, 1, tw, th)
    bg = None
    if background is not None:
        bg = background.reshape(1, 3)
    img, alpha = rasterize_to_pixels(
        means2d, conics, colors, opacities.reshape(1, -1), width, height, tile_size,
        offsets, flatten_ids, backgrounds=bg,
    )
    return img[0].permute(2, 0, 1).contiguous(), means2d, radii, dict(
        depths=depths, conics=conics, colors=colors, isect_ids=isect_ids,
        flatten_ids=flatten_ids, offsets=offsets, alpha=alpha,
    )


# ----------------------------------------------------------------------------
# A8: Adam variants  (optimizer.py:6-184; clm_offload/gaussian_model.py:161-211)
# ----------------------------------------------------------------------------


def adam_rows(p, g, m, v, rows, col_lr, beta1, beta2, eps, step, scale=1.0,
              bias_correction=True, zero_grad=False):
    """Row-sparse Adam with per-column learning rate (FusedCPUAdam semantics):
    rows=None -> every row.  `step` is the 1-based global step used for bias
    correction (DeepSpeed cpu_adam convention: step_size = lr / bc1,
    denom = sqrt(v)/sqrt(bc2) + eps)."""
    idx = slice(None) if rows is None else rows.long()
    gg = g[idx] * scale
    m[idx] = beta1 * m[idx] + (1 - beta1) * gg
    v[idx] = beta2 * v[idx] + (1 - beta2) * gg * gg
    if bias_correction:
        bc1 = 1 - beta1**step
        bc2s = math.sqrt(1 - beta2**step)
    else:
        bc1, bc2s = 1.0, 1.0
    denom = v[idx].sqrt() / bc2s + eps
    p[idx] = p[idx] - (col_lr[None, :] / bc1) * (m[idx] / denom)
    if zero_grad:
        g[idx] = 0


def selective_adam(p, g, m, v, visibility, lr, beta1, beta2, eps):
    """optimizer.py:76-88 / Taming-3DGS: masked Adam, no bias correction."""
    rows = torch.nonzero(visibility).flatten()
    p2, g2, m2, v2 = (t.reshape(visibility.numel(), -1) for t in (p, g, m, v))
    gg = g2[rows]
    m2[rows] = beta1 * m2[rows] + (1 - beta1) * gg
    v2[rows] = beta2 * v2[rows] + (1 - beta2) * gg * gg
    p2[rows] = p2[rows] - lr * m2[rows] / (v2[rows].sqrt() + eps)
