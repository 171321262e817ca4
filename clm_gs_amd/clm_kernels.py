"""clm_kernels operator surface used by the CLM-GS engines, on gfx950.

Same names and argument meaning as the reference's call sites
(strategies/base_engine.py:5,93; strategies/clm_offload/engine.py:14-20,152-153,
200-204,227-232,499-505,622-636,709-716,789-825; optimizer.py:3,76-88).
"""
import ctypes

import torch

from . import _lib, utils
from ._lib import check, dptr, stream

F32, I32, I64, U8 = torch.float32, torch.int32, torch.int64, torch.uint8


# ------------------------------------------------------------------- fused SSIM
class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        L = _lib.lib()
        img1, img2 = img1.contiguous(), img2.contiguous()
        B, CH, H, W = img1.shape
        need = img1.requires_grad
        ssim_sum = torch.zeros((1024,), dtype=F32, device=img1.device)
        maps = [torch.empty_like(img1) for _ in range(3)] if need else [None, None, None]
        check(L.clmgs_ssim_fwd(stream(), B, CH, H, W, dptr(img1, F32), dptr(img2, F32),
                               dptr(ssim_sum), dptr(maps[0], F32, True), dptr(maps[1], F32, True),
                               dptr(maps[2], F32, True)))
        ctx.shape = (B, CH, H, W)
        if need:
            ctx.save_for_backward(img1, img2, *maps)
        return ssim_sum.sum() / float(img1.numel())

    @staticmethod
    def backward(ctx, v):
        L = _lib.lib()
        img1, img2, m0, m1, m2 = ctx.saved_tensors
        B, CH, H, W = ctx.shape
        v_img1 = torch.empty_like(img1)
        v = v.reshape(1).to(F32).contiguous()
        check(L.clmgs_ssim_bwd(stream(), B, CH, H, W, dptr(img1), dptr(img2), dptr(v, F32),
                               1.0 / float(img1.numel()), dptr(m0), dptr(m1), dptr(m2),
                               dptr(v_img1)))
        return v_img1, None


def fused_ssim(img1, img2):
    """Mean SSIM of [B,CH,H,W] images (11x11, sigma 1.5, zero padded), differentiable in img1."""
    return _FusedSSIM.apply(img1, img2)


class _FusedL1SSIMLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt_u8, lambda_dssim):
        L = _lib.lib()
        assert image.dim() == 3 and image.shape[0] == 3 and image.dtype == F32 and image.is_cuda
        _, H, W = image.shape
        gt_u8 = gt_u8.contiguous()
        assert gt_u8.dtype == U8 and gt_u8.shape == image.shape
        need = image.requires_grad
        slots = L.clmgs_loss_slots()
        partials = torch.zeros((slots, 2), dtype=F32, device=image.device)
        maps = torch.empty((3, 3, H, W), dtype=F32, device=image.device) if need else None
        sc, sy, sx = image.stride()
        check(L.clmgs_l1_ssim_loss_fwd(stream(), H, W, ctypes.c_void_p(image.data_ptr()), sc, sy, sx,
                                       dptr(gt_u8, U8), dptr(partials),
                                       dptr(maps[0] if need else None, F32, True),
                                       dptr(maps[1] if need else None, F32, True),
                                       dptr(maps[2] if need else None, F32, True)))
        tot = partials.sum(dim=0) / float(image.numel())
        ctx.lam = float(lambda_dssim)
        if need:
            ctx.save_for_backward(image, gt_u8, maps)
        return (1.0 - ctx.lam) * tot[0] + ctx.lam * (1.0 - tot[1])

    @staticmethod
    def backward(ctx, v):
        L = _lib.lib()
        image, gt_u8, maps = ctx.saved_tensors
        _, H, W = image.shape
        v = v.reshape(1).to(F32).contiguous()
        v_img = torch.empty_strided(image.shape, image.stride(), dtype=F32, device=image.device)
        sc, sy, sx = image.stride()
        check(L.clmgs_l1_ssim_loss_bwd(stream(), H, W, ctypes.c_void_p(image.data_ptr()), sc, sy, sx,
                                       dptr(gt_u8, U8), dptr(v, F32), ctx.lam, dptr(maps[0]),
                                       dptr(maps[1]), dptr(maps[2]),
                                       ctypes.c_void_p(v_img.data_ptr())))
        return v_img, None, None


def fused_l1_ssim_loss(image, gt_u8, lambda_dssim=0.2):
    """(1-lambda) * L1 + lambda * (1 - SSIM) of image [3,H,W] (any strides, e.g. a permuted view
    of the rasterizer's [H,W,3] output) against a uint8 [3,H,W] ground truth; one forward and one
    backward kernel (strategies/base_engine.py:79-103 semantics)."""
    return _FusedL1SSIMLoss.apply(image, gt_u8, lambda_dssim)


# ------------------------------------------------------------- SH row movement
def _idx64(t):
    return 1 if t is not None and t.dtype == I64 else 0


def _rows(fn, dst, src, dst_idx, src_idx, grid):
    L = _lib.lib()
    n = (dst_idx if dst_idx is not None else src_idx).numel() if (dst_idx is not None or src_idx is not None) else dst.shape[0]
    if dst_idx is not None and src_idx is not None and dst_idx.dtype != src_idx.dtype:
        src_idx = src_idx.to(dst_idx.dtype)
    is64 = _idx64(dst_idx if dst_idx is not None else src_idx)
    cols = src.shape[-1]
    check(getattr(L, fn)(stream(), dptr(dst, F32, allow_host=True), dptr(src, F32, allow_host=True),
                         dptr(dst_idx, None, True), dptr(src_idx, None, True), is64, int(n),
                         int(cols), int(grid)))


def send_shs2gpu_stream(shs, parameters, filter_idx, grid_size=0, block_size=256):
    """shs[i] = parameters[filter[i]]  (parameters may be pinned host memory)."""
    _rows("clmgs_rows_gather", shs, parameters, None, filter_idx, grid_size)


def send_shs2gpu_stream_retention(shs_next, parameters, shs_retent, host_indices_to_param,
                                  rtnt_indices_to_param, param_indices_from_host,
                                  param_indices_from_rtnt, grid_size=0, block_size=256,
                                  grid_size_D=0, block_size_D=256):
    """shs_next[param_indices_from_host[i]] = parameters[host_indices_to_param[i]] and
    shs_next[param_indices_from_rtnt[j]] = shs_retent[rtnt_indices_to_param[j]]."""
    if host_indices_to_param.numel():
        _rows("clmgs_rows_gather", shs_next, parameters, param_indices_from_host,
              host_indices_to_param, grid_size)
    if rtnt_indices_to_param.numel():
        _rows("clmgs_rows_gather", shs_next, shs_retent, param_indices_from_rtnt,
              rtnt_indices_to_param, grid_size_D)


def send_shs2cpu_grad_buffer_stream(shs_grad, grad_buffer, filter_idx, accum=True, grid_size=0,
                                    block_size=256):
    """grad_buffer[filter[i]] (+)= shs_grad[i]."""
    _rows("clmgs_rows_scatter_add" if accum else "clmgs_rows_gather", grad_buffer, shs_grad,
          filter_idx, None, grid_size)


def send_shs2cpu_grad_buffer_stream_retention(shs_grad, grad_buffer, shs_grad_next,
                                              host_indices_from_grad, rtnt_indices_from_grad,
                                              grad_indices_to_host, grad_indices_to_rtnt,
                                              accum=True, grid_size=0, block_size=256,
                                              grid_size_D=0, block_size_D=256):
    """grad_buffer[host_indices_from_grad[i]] += shs_grad[grad_indices_to_host[i]] and
    shs_grad_next[rtnt_indices_from_grad[j]] = shs_grad[grad_indices_to_rtnt[j]]."""
    if host_indices_from_grad.numel():
        _rows("clmgs_rows_scatter_add" if accum else "clmgs_rows_gather", grad_buffer, shs_grad,
              host_indices_from_grad, grad_indices_to_host, grid_size)
    if rtnt_indices_from_grad.numel():
        _rows("clmgs_rows_gather", shs_grad_next, shs_grad, rtnt_indices_from_grad,
              grad_indices_to_rtnt, grid_size_D)


@torch.no_grad()
def spherical_harmonics_bwd_inplace(degrees_to_use, dirs, coeffs, v_coeffs, v_colors):
    """SH backward that ACCUMULATES into the persistent v_coeffs[n,48] buffer and returns
    v_dirs (clm_offload/engine.py:709-716)."""
    L = _lib.lib()
    n = dirs.numel() // 3
    d2, c2, vc = dirs.contiguous(), coeffs.contiguous(), v_colors.contiguous()
    v_dirs = torch.empty_like(d2)
    check(L.clmgs_sh_bwd(stream(), n, int(degrees_to_use), dptr(d2, F32), dptr(c2, F32), None,
                         dptr(vc, F32), dptr(v_coeffs, F32), 1, dptr(v_dirs)))
    return v_dirs


# ------------------------------------------------------------------- bitmaps
def scatter_to_bit(bitmap, filter_idx, bit):
    L = _lib.lib()
    check(L.clmgs_scatter_to_bit(stream(), dptr(bitmap), bitmap.element_size(),
                                 dptr(filter_idx.contiguous(), I64), filter_idx.numel(), int(bit)))


def extract_ffs(bitmap, ffs):
    L = _lib.lib()
    check(L.clmgs_extract_ffs(stream(), dptr(bitmap), bitmap.element_size(), bitmap.numel(),
                              dptr(ffs, U8)))


def compute_cnt_h(bitmap, tmp_buffer, grid_size=64, block_size=256):
    """Reference contract (clm_offload/engine.py:227-233): fill tmp_buffer[bsz-1, T] with
    partial counts whose row sums are cnt_d[i] = #(F_i & F_{i+1}).  Here the full count lands
    in column 0 and the remaining columns are zero."""
    L = _lib.lib()
    bsz = tmp_buffer.shape[0] + 1
    cnt = torch.zeros((bsz - 1,), dtype=I32, device=bitmap.device)
    check(L.clmgs_pair_overlap_count(stream(), dptr(bitmap), bitmap.element_size(), bitmap.numel(),
                                     bsz, dptr(cnt)))
    tmp_buffer.zero_()
    tmp_buffer[:, 0] = cnt
    return cnt


def pair_overlap_count(bitmap, bsz):
    L = _lib.lib()
    cnt = torch.zeros((bsz - 1,), dtype=I32, device=bitmap.device)
    check(L.clmgs_pair_overlap_count(stream(), dptr(bitmap), bitmap.element_size(), bitmap.numel(),
                                     int(bsz), dptr(cnt)))
    return cnt


def set_signal(signal_tensor_pinned, idx, value):
    L = _lib.lib()
    check(L.clmgs_set_signal(stream(), ctypes.c_void_p(signal_tensor_pinned.data_ptr()), int(idx),
                             int(value)))


# ----------------------------------------------------------------------- Adam
def selective_adam_update(param, grad, exp_avg, exp_avg_sq, visibility, lr, beta1, beta2, eps, N, M):
    """Adam on rows where visibility is True, no bias correction (optimizer.py:76-88)."""
    L = _lib.lib()
    col_lr = torch.full((M,), float(lr), dtype=F32, device=param.device)
    vis = visibility.contiguous().view(U8)
    check(L.clmgs_adam_rows(stream(), dptr(param, F32), dptr(grad, F32), dptr(exp_avg, F32),
                            dptr(exp_avg_sq, F32), None, 0, dptr(vis, U8), int(N), int(M),
                            dptr(col_lr), float(beta1), float(beta2), float(eps), 1, 0, 1.0, 0))


def adam_rows(p, g, m, v, rows, col_lr, beta1, beta2, eps, step, bias_correction=True,
              grad_scale=1.0, zero_grad=False, mask=None):
    """Row-sparse Adam with per-column learning rate on device tensors [*, cols]."""
    L = _lib.lib()
    n_rows = rows.numel() if rows is not None else p.shape[0]
    cols = p.shape[-1] if p.dim() > 1 else 1
    check(L.clmgs_adam_rows(stream(), dptr(p, F32), dptr(g, F32, True), dptr(m, F32), dptr(v, F32),
                            dptr(rows, None, True), _idx64(rows),
                            dptr(mask.view(U8) if mask is not None else None, U8, True),
                            int(n_rows), int(cols), dptr(col_lr, F32), float(beta1), float(beta2),
                            float(eps), int(step), int(bool(bias_correction)), float(grad_scale),
                            int(bool(zero_grad))))


def adam_catch_up(p, m, v, last_step, rows, col_lr, beta1, beta2, eps, to_step, bias_correction=True,
                  max_replay=256, g=None, g_step=None, grad_scale=1.0, keep_grad=False, moment_row0=0):
    """Replay the deferred zero-gradient Adam steps of `rows` (None = all) up to `to_step` and
    stamp them; with g / g_step also apply the gradient step that is waiting for a row, at its own
    step (see clmgs_adam_catch_up).  moment_row0 > 0: m / v are a SHARD of the moment tables whose first row is
    global row `moment_row0` (camera-DP, moments held by the owner of a row range only); the explicit row list
    must then stay inside the shard -- the library indexes every table by global row id from the base pointer it
    is given (include/clmgs.h), so the shard travels as a base moved back by moment_row0 rows."""
    L = _lib.lib()
    n_rows = rows.numel() if rows is not None else p.shape[0]
    m_ptr, v_ptr = dptr(m, F32), dptr(v, F32)
    if moment_row0:
        assert rows is not None, "a moment shard needs an explicit row list"
        back = int(moment_row0) * int(p.shape[-1]) * 4
        m_ptr, v_ptr = ctypes.c_void_p(m_ptr.value - back), ctypes.c_void_p(v_ptr.value - back)
    check(L.clmgs_adam_catch_up(stream(), dptr(p, F32), m_ptr, v_ptr, dptr(last_step, I32),
                                dptr(rows, None, True), _idx64(rows), int(n_rows), int(p.shape[-1]),
                                dptr(col_lr, F32), float(beta1), float(beta2), float(eps), int(to_step),
                                int(bool(bias_correction)), int(max_replay), dptr(g, F32, True),
                                dptr(g_step, I32, True), float(grad_scale), int(bool(keep_grad))))
    if rows is None:
        last_step[: p.shape[0]].fill_(int(to_step))
    else:
        utils.fill_rows(last_step, rows.long(), int(to_step))  # scalar as kernel argument: no blocking H2D copy


def densify_stats(filter_idx, v_means2d, radii, width, height, max_radii2D, xyz_gradient_accum,
                  denom, only_visible=True):
    """Fused form of gsplat_add_densification_stats[_exact_filter]
    (clm_offload/gaussian_model.py:833-851, no_offload/gaussian_model.py:767-783)."""
    L = _lib.lib()
    n = radii.numel()
    check(L.clmgs_densify_stats(stream(), n, dptr(filter_idx, I64, True),
                                dptr(v_means2d.contiguous(), F32), dptr(radii.contiguous(), I32),
                                int(bool(only_visible)), float(width) * 0.5, float(height) * 0.5, dptr(max_radii2D, F32),
                                dptr(xyz_gradient_accum, F32), dptr(denom, F32)))
