"""The five gsplat operators the CLM-GS engines import
(strategies/base_engine.py:6-12, strategies/no_offload/engine.py:4-10,
strategies/clm_offload/engine.py:7-13), with the same names, argument meaning
and return shapes, served by the gfx950 kernels of libclmgs_hip.so.

Differentiable operators are ``torch.autograd.Function``s whose forward and
backward call the C ABI; there is no eager fallback.
"""

import time

import torch

from . import _lib
from ._lib import check, dptr, stream

F32, I32, I64 = torch.float32, torch.int32, torch.int64


# --------------------------------------------------------------------- projection
class _Projection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane,
                far_plane, radius_clip):
        L = _lib.lib()
        means, quats, scales = means.contiguous(), quats.contiguous(), scales.contiguous()
        viewmats, Ks = viewmats.contiguous(), Ks.contiguous()
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        radii = torch.empty((C, N), dtype=I32, device=dev)
        means2d = torch.empty((C, N, 2), dtype=F32, device=dev)
        depths = torch.empty((C, N), dtype=F32, device=dev)
        conics = torch.empty((C, N, 3), dtype=F32, device=dev)
        check(L.clmgs_projection_fwd(
            stream(), C, N, dptr(means, F32), dptr(quats, F32), dptr(scales, F32),
            dptr(viewmats, F32), dptr(Ks, F32), int(width), int(height), float(eps2d),
            float(near_plane), float(far_plane), float(radius_clip), dptr(radii), dptr(means2d),
            dptr(depths), dptr(conics)))
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii)
        ctx.cfg = (int(width), int(height), float(eps2d))
        ctx.mark_non_differentiable(radii)
        return radii, means2d, depths, conics

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics):
        L = _lib.lib()
        means, quats, scales, viewmats, Ks, radii = ctx.saved_tensors
        width, height, eps2d = ctx.cfg
        C, N = radii.shape
        dev = means.device
        if v_means2d is None:
            v_means2d = torch.zeros((C, N, 2), dtype=F32, device=dev)
        if v_conics is None:
            v_conics = torch.zeros((C, N, 3), dtype=F32, device=dev)
        v_means2d, v_conics = v_means2d.contiguous(), v_conics.contiguous()
        v_depths = v_depths.contiguous() if v_depths is not None else None
        v_means = torch.empty_like(means)
        v_quats = torch.empty_like(quats)
        v_scales = torch.empty_like(scales)
        check(L.clmgs_projection_bwd(
            stream(), C, N, dptr(means), dptr(quats), dptr(scales), dptr(viewmats), dptr(Ks),
            width, height, eps2d, dptr(radii), dptr(v_means2d, F32), dptr(v_depths, F32, True),
            dptr(v_conics, F32), dptr(v_means), dptr(v_quats), dptr(v_scales)))
        return v_means, v_quats, v_scales, None, None, None, None, None, None, None, None


def fully_fused_projection(means, covars, quats, scales, viewmats, Ks, width, height, eps2d=0.3,
                           near_plane=0.01, far_plane=1e10, radius_clip=0.0, packed=False,
                           sparse_grad=False, calc_compensations=False):
    """EWA projection of N Gaussians into C cameras.

    Unpacked -> (radii[C,N] i32, means2d[C,N,2], depths[C,N], conics[C,N,3], None).
    Packed (no grad; strategies/base_engine.py:36-47 uses it under no_grad only)
    -> (camera_ids, gaussian_ids, radii, means2d, depths, conics, None), ordered
    by (camera, gaussian).
    """
    if covars is not None:
        raise NotImplementedError("the CLM-GS engines always pass covars=None")
    if calc_compensations:
        raise NotImplementedError("compensations are never requested by the CLM-GS engines")
    if not packed:
        radii, means2d, depths, conics = _Projection.apply(
            means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
            radius_clip)
        return radii, means2d, depths, conics, None
    with torch.no_grad():
        radii, means2d, depths, conics = _Projection.apply(
            means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
            radius_clip)
        camera_ids, gaussian_ids = torch.nonzero(radii > 0, as_tuple=True)
        return (camera_ids, gaussian_ids, radii[camera_ids, gaussian_ids],
                means2d[camera_ids, gaussian_ids], depths[camera_ids, gaussian_ids],
                conics[camera_ids, gaussian_ids], None)


def visibility_radii(means, quats, scales, viewmats, Ks, width, height, eps2d=0.3,
                     near_plane=0.01, far_plane=1e10, radius_clip=0.0, raw=False):
    """radii[C,N] only: the cull of fully_fused_projection without writing the 24 B of
    per-pair outputs (what calculate_filters, strategies/base_engine.py:18-76, needs).
    raw=True: `quats` un-normalised and `scales` as logs (the stored parameters)."""
    L = _lib.lib()
    means, quats, scales = means.contiguous(), quats.contiguous(), scales.contiguous()
    viewmats, Ks = viewmats.contiguous(), Ks.contiguous()
    C, N = viewmats.shape[0], means.shape[0]
    radii = torch.empty((C, N), dtype=I32, device=means.device)
    if raw:
        check(L.clmgs_visibility_raw(
            stream(), C, N, dptr(means, F32), dptr(quats, F32), dptr(scales, F32), dptr(viewmats, F32),
            dptr(Ks, F32), int(width), int(height), float(eps2d), float(near_plane), float(far_plane),
            float(radius_clip), dptr(radii)))
        return radii
    check(L.clmgs_projection_fwd(
        stream(), C, N, dptr(means, F32), dptr(quats, F32), dptr(scales, F32), dptr(viewmats, F32),
        dptr(Ks, F32), int(width), int(height), float(eps2d), float(near_plane), float(far_plane),
        float(radius_clip), dptr(radii), None, None, None))
    return radii


def visibility_select(means, quats_raw, log_scales, viewmats, Ks, width, height, eps2d=0.3,
                      near_plane=0.01, far_plane=1e10, radius_clip=0.0, block_flags=None):
    """The batch's visibility filters selected on the GPU: same cull as visibility_radii(raw=True)
    but without the radii[C,N] round trip and torch.nonzero -- one ballot word per (camera, 64
    Gaussians), a scan, and an emit pass.  -> (filters: tuple of C int64 index tensors (ascending),
    touched_rows: int64 indices of the union over the cameras).  One host read (the C+1 totals)."""
    L = _lib.lib()
    means, quats_raw, log_scales = means.contiguous(), quats_raw.contiguous(), log_scales.contiguous()
    viewmats, Ks = viewmats.contiguous(), Ks.contiguous()
    C, N = viewmats.shape[0], means.shape[0]
    dev = means.device
    tb = L.clmgs_visibility_select_temp_bytes(C, N)
    temp = torch.empty((tb,), dtype=torch.uint8, device=dev)
    cum = torch.empty((C + 1,), dtype=I64, device=dev)
    # block_flags (uint8 per 256 rows, 0 = no row of the block can be visible: clmgs_adam_small_deferred's candidate test)
    check(L.clmgs_visibility_select_count_blocks(
        stream(), C, N, dptr(means, F32), dptr(quats_raw, F32), dptr(log_scales, F32), dptr(viewmats, F32),
        dptr(Ks, F32), int(width), int(height), float(eps2d), float(near_plane), float(far_plane),
        float(radius_clip), dptr(temp), tb, dptr(cum), dptr(block_flags, torch.uint8, True)))
    _t0 = time.perf_counter()
    ends = cum.tolist()  # the one host sync of the filter stage
    _lib.STATS["host_wait_s"] += time.perf_counter() - _t0
    out = torch.empty((max(ends[-1], 1),), dtype=I64, device=dev)
    if ends[-1] > 0:
        check(L.clmgs_visibility_select_emit(stream(), C, N, dptr(temp), dptr(out)))
    starts = [0] + ends[:-1]
    pieces = tuple(out[a:b] for a, b in zip(starts, ends))
    return pieces[:C], pieces[C]


def visibility_candidates(means, log_scales, viewmats, Ks, width, height, pos_margin, scale_gain, own_lo, own_hi,
                          eps2d=0.3, near_plane=0.01, far_plane=1e10):
    """Camera-DP, small attributes computed by their owners: ascending int64 ids of the rows OUTSIDE [own_lo, own_hi)
    that may pass visibility_select's cull in any camera while their stored mean is off by up to pos_margin and their
    largest scale by a factor up to scale_gain (clmgs_visibility_candidates: a superset of what the exact cull keeps
    for the true values).  One host read inside torch.nonzero (the count)."""
    L = _lib.lib()
    means, log_scales = means.contiguous(), log_scales.contiguous()
    viewmats, Ks = viewmats.contiguous(), Ks.contiguous()
    C, N = viewmats.shape[0], means.shape[0]
    mask = torch.empty((N,), dtype=torch.uint8, device=means.device)
    check(L.clmgs_visibility_candidates(
        stream(), C, N, int(own_lo), int(own_hi), dptr(means, F32), dptr(log_scales, F32), dptr(viewmats, F32),
        dptr(Ks, F32), int(width), int(height), float(eps2d), float(near_plane), float(far_plane),
        float(pos_margin), float(scale_gain), dptr(mask)))
    return torch.nonzero(mask).flatten()


# ------------------------------------------------------------ spherical harmonics
class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree, dirs, coeffs, masks):
        L = _lib.lib()
        lead = dirs.shape[:-1]
        n = dirs.numel() // 3
        assert coeffs.shape[-2:] == (16, 3) or coeffs.shape[-1] == 48, coeffs.shape
        d2, c2 = dirs.contiguous(), coeffs.contiguous()
        m2 = masks.contiguous().view(torch.uint8) if masks is not None else None
        colors = torch.empty(lead + (3,), dtype=F32, device=dirs.device)
        check(L.clmgs_sh_fwd(stream(), n, int(degree), dptr(d2, F32), dptr(c2, F32),
                             dptr(m2, torch.uint8, True), dptr(colors)))
        ctx.save_for_backward(d2, c2, m2)
        ctx.degree = int(degree)
        ctx.n = n
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        L = _lib.lib()
        d2, c2, m2 = ctx.saved_tensors
        v_colors = v_colors.contiguous()
        v_coeffs = torch.empty_like(c2) if ctx.needs_input_grad[2] else None
        v_dirs = torch.empty_like(d2) if ctx.needs_input_grad[1] else None
        if v_coeffs is None:  # kernel always produces v_coeffs; give it scratch
            v_coeffs_buf = torch.empty_like(c2)
        else:
            v_coeffs_buf = v_coeffs
        check(L.clmgs_sh_bwd(stream(), ctx.n, ctx.degree, dptr(d2), dptr(c2),
                             dptr(m2, torch.uint8, True), dptr(v_colors, F32), dptr(v_coeffs_buf),
                             0, dptr(v_dirs, F32, True)))
        return None, v_dirs, v_coeffs, None


def spherical_harmonics(degrees_to_use, dirs, coeffs, masks=None):
    """colors[...,3] from dirs[...,3] (normalised inside) and coeffs[...,16,3]; rows with
    masks == False produce 0 and receive 0 gradient."""
    return _SphericalHarmonics.apply(degrees_to_use, dirs, coeffs, masks)


# ------------------------------------------------------------------- tile binning
@torch.no_grad()
def isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True,
                packed=False, n_cameras=None, camera_ids=None, gaussian_ids=None):
    """-> (tiles_per_gauss[C,N] i32, isect_ids[I] i64 sorted, flatten_ids[I] i32)."""
    if packed or not sort:
        raise NotImplementedError("the CLM-GS engines call isect_tiles(packed=False, sort=True)")
    L = _lib.lib()
    C, N = radii.shape
    dev = radii.device
    means2d, radii, depths = means2d.contiguous(), radii.contiguous(), depths.contiguous()
    tiles_per_gauss = torch.empty((C, N), dtype=I32, device=dev)
    cum = torch.empty((C * N,), dtype=I64, device=dev)
    if C * N == 0:
        return tiles_per_gauss, torch.empty(0, dtype=I64, device=dev), torch.empty(0, dtype=I32, device=dev)
    tb = L.clmgs_isect_count_temp_bytes(C * N)
    temp = torch.empty((tb,), dtype=torch.uint8, device=dev)
    check(L.clmgs_isect_count(stream(), C, N, dptr(means2d, F32), dptr(radii, I32), int(tile_size),
                              int(tile_width), int(tile_height), dptr(tiles_per_gauss), dptr(cum),
                              dptr(temp), tb))
    n_isects = int(cum[-1].item())  # data-dependent size: the one host sync of the front end
    _lib.STATS["n_isects"].append(n_isects)
    if len(_lib.STATS["n_isects"]) > 4096:
        del _lib.STATS["n_isects"][:2048]
        del _lib.STATS["n_emitted"][:2048]
    isect_ids = torch.empty((n_isects,), dtype=I64, device=dev)
    flatten_ids = torch.empty((n_isects,), dtype=I32, device=dev)
    if n_isects:
        sb = L.clmgs_isect_sort_temp_bytes(n_isects)
        temp2 = torch.empty((sb,), dtype=torch.uint8, device=dev)
        check(L.clmgs_isect_emit_sort(stream(), C, N, n_isects, dptr(means2d), dptr(radii),
                                      dptr(depths, F32), dptr(cum), int(tile_size),
                                      int(tile_width), int(tile_height), dptr(isect_ids),
                                      dptr(flatten_ids), dptr(temp2), sb))
    return tiles_per_gauss, isect_ids, flatten_ids


@torch.no_grad()
def isect_offset_encode(isect_ids, n_cameras, tile_width, tile_height):
    """offsets[C, tile_height, tile_width] i32 = first sorted index of each tile."""
    L = _lib.lib()
    offsets = torch.empty((n_cameras, tile_height, tile_width), dtype=I32, device=isect_ids.device)
    check(L.clmgs_isect_offsets(stream(), isect_ids.numel(), dptr(isect_ids.contiguous(), I64),
                                int(n_cameras), int(tile_width), int(tile_height), dptr(offsets)))
    return offsets


def bucket_size(n):
    """n rounded up to 1/8 steps of its leading power of two (<= 12.5 % more).  The data-dependent
    buffers of a camera (rows V, intersections I) are allocated at bucketed capacity and viewed
    [:n]: the same few sizes then recur batch after batch and the caching allocator serves them
    from its cache instead of calling hipMalloc in the middle of a step (a 9 ms stall when it
    happened inside the 5 timed steps of the bench)."""
    n = int(n)
    if n <= 4096:
        return 4096
    sh = n.bit_length() - 4
    return ((n + (1 << sh) - 1) >> sh) << sh


def empty_bucketed(n, trailing, dtype, device):
    """torch.empty((n, *trailing)) backed by a bucket_size(n)-row allocation."""
    return torch.empty((bucket_size(n),) + tuple(trailing), dtype=dtype, device=device)[:n]


_PINNED_TOTALS = None  # ring of pinned int64[2] slots for the asynchronous size readbacks
_PINNED_NEXT = 0


def _pinned_totals_slot():
    global _PINNED_TOTALS, _PINNED_NEXT
    if _PINNED_TOTALS is None:
        _PINNED_TOTALS = torch.empty((256, 2), dtype=I64).pin_memory()
    _PINNED_NEXT = (_PINNED_NEXT + 1) % _PINNED_TOTALS.shape[0]
    return _PINNED_TOTALS[_PINNED_NEXT]


class _Isect2:
    """State between isect2_begin and isect2_finish (one camera's binning in flight)."""
    __slots__ = ("args", "V", "dev", "depths", "order", "cum", "boxes", "totals", "host", "event",
                 "offsets", "temp", "means2d", "radii", "packed", "row_cum", "route")


@torch.no_grad()
def isect2_begin(means2d, radii, depths, tile_size, tile_width, tile_height, want_isect_ids=False,
                 want_slots=False, packed=None):
    """First half of isect_tiles_two_level: depth order + per-row tile counts on the CURRENT stream,
    then an asynchronous copy of the two totals into pinned memory and an event.  Nothing blocks:
    the caller can enqueue other work (the next camera's projection) before isect2_finish waits for
    the event -- unlike a `.tolist()`, which waits for everything enqueued on the stream so far."""
    L = _lib.lib()
    c = _Isect2()
    c.route = "sort"
    c.args = (int(tile_size), int(tile_width), int(tile_height), bool(want_isect_ids), bool(want_slots))
    V = c.V = radii.numel()
    dev = c.dev = radii.device
    c.means2d, c.radii, c.depths = means2d.contiguous(), radii.contiguous(), depths.contiguous()
    c.packed = packed
    c.offsets = torch.empty((1, tile_height, tile_width), dtype=I32, device=dev)
    c.event = None
    if V == 0:
        return c
    c.order = empty_bucketed(V, (), I32, dev)
    c.cum = empty_bucketed(V, (), I64, dev)
    c.boxes = empty_bucketed(V, (2,), I64, dev)
    c.totals = torch.empty((2,), dtype=I64, device=dev)
    c.row_cum = empty_bucketed(V, (), I64, dev) if want_slots else None
    tb = L.clmgs_isect2_order_temp_bytes(V)
    c.temp = empty_bucketed(tb, (), torch.uint8, dev)
    check(L.clmgs_isect2_order_count(stream(), V, dptr(c.means2d, F32), dptr(c.radii, I32), dptr(c.depths, F32),
                                     c.args[0], c.args[1], c.args[2], dptr(packed, F32, True), dptr(c.order),
                                     dptr(c.cum), dptr(c.boxes), dptr(c.totals), dptr(c.temp), tb,
                                     dptr(c.row_cum, I64, True)))
    c.host = _pinned_totals_slot()
    c.host.copy_(c.totals, non_blocking=True)
    c.event = torch.cuda.Event()
    c.event.record(torch.cuda.current_stream())
    return c


def _record_counts(n_isects, n_ref):
    _lib.STATS["n_isects"].append(n_ref)        # the reference's (3-sigma box) intersection count
    _lib.STATS["n_emitted"].append(n_isects)    # what is actually sorted and blended
    if len(_lib.STATS["n_isects"]) > 4096:
        del _lib.STATS["n_isects"][:2048]
        del _lib.STATS["n_emitted"][:2048]


def isect2_counts(c):
    """Wait for the totals of isect2_begin (an event on an asynchronous readback) -> (emitted, reference)."""
    _t0 = time.perf_counter()
    c.event.synchronize()
    n_isects, n_ref = int(c.host[0]), int(c.host[1])
    _lib.STATS["host_wait_s"] += time.perf_counter() - _t0
    return n_isects, n_ref


@torch.no_grad()
def isect2_finish(c, capacity=None):
    """Second half: emit + tile sort + offsets on the CURRENT stream (the stream isect2_begin ran on, or one
    ordered after it).
    capacity None: wait for the totals (the one host wait of the front end) and size everything exactly.
    capacity = K (device-count form, clmgs_isect2_emit_sort_dev): NOTHING is waited for -- buffers and launches
    are sized for K intersections and the kernels read the true count from c.totals on the device; the returned
    lists have K entries of which the first `count` are valid.  The caller must check isect2_counts(c)[0] <= K
    later (fused.camera_verify) and redo the camera exactly otherwise."""
    L = _lib.lib()
    tile_size, tile_width, tile_height, want_isect_ids, want_slots = c.args
    dev, V = c.dev, c.V
    if V == 0:
        c.offsets.zero_()
        e = torch.empty(0, dtype=I32, device=dev)
        res = (e, c.offsets, (torch.empty(0, dtype=I64, device=dev) if want_isect_ids else None))
        return res + ((e, torch.empty(0, dtype=I64, device=dev)),) if want_slots else res
    if capacity is None:
        n_isects, n_ref = isect2_counts(c)
        _record_counts(n_isects, n_ref)
    else:
        n_isects = int(capacity)
    fids = empty_bucketed(n_isects, (), I32, dev)
    ids = empty_bucketed(n_isects, (), I64, dev) if want_isect_ids else None
    sb = L.clmgs_isect2_sort_temp_bytes(n_isects)
    temp2 = empty_bucketed(sb, (), torch.uint8, dev)
    emit_slot = empty_bucketed(n_isects, (), I32, dev) if want_slots else None
    if capacity is None:
        check(L.clmgs_isect2_emit_sort(stream(), V, n_isects, dptr(c.depths), dptr(c.order), dptr(c.cum),
                                       dptr(c.boxes), tile_width, tile_height, dptr(fids), dptr(c.offsets),
                                       dptr(ids, I64, True), dptr(emit_slot, I32, True), dptr(temp2), sb,
                                       dptr(c.row_cum, I64, True)))
    else:
        check(L.clmgs_isect2_emit_sort_dev(stream(), V, n_isects, dptr(c.totals), dptr(c.depths), dptr(c.order),
                                           dptr(c.cum), dptr(c.boxes), tile_width, tile_height, dptr(fids),
                                           dptr(c.offsets), dptr(ids, I64, True), dptr(emit_slot, I32, True),
                                           dptr(temp2), sb, dptr(c.row_cum, I64, True)))
    return (fids, c.offsets, ids, (emit_slot, c.row_cum)) if want_slots else (fids, c.offsets, ids)


# ---- tile-major binning (csrc/isect3.hip): the same lists without a global sort; same two-phase calling pattern
@torch.no_grad()
def isect3_begin(means2d, radii, depths, tile_size, tile_width, tile_height, want_isect_ids=False,
                 want_slots=False, packed=None):
    """First half of the tile-major binning (clmgs_isect3_front): per-row tile boxes / masks, tile counters, row_cum and
    the totals on the CURRENT stream, then the asynchronous readback of the totals -- the calling pattern (and the
    state object) of isect2_begin; isect2_counts reads the totals of either."""
    L = _lib.lib()
    c = _Isect2()
    c.route = "tile"
    c.args = (int(tile_size), int(tile_width), int(tile_height), bool(want_isect_ids), bool(want_slots))
    V = c.V = radii.numel()
    dev = c.dev = radii.device
    c.means2d, c.radii, c.depths = means2d.contiguous(), radii.contiguous(), depths.contiguous()
    c.packed = packed
    c.offsets = torch.empty((1, tile_height, tile_width), dtype=I32, device=dev)
    c.event = None
    c.order = c.cum = c.boxes = None
    if V == 0:
        return c
    c.totals = torch.empty((2,), dtype=I64, device=dev)
    c.row_cum = empty_bucketed(V, (), I64, dev)
    tb = L.clmgs_isect3_front_temp_bytes(V, int(tile_width) * int(tile_height))
    c.temp = empty_bucketed(tb, (), torch.uint8, dev)
    check(L.clmgs_isect3_front(stream(), V, dptr(c.means2d, F32), dptr(c.radii, I32), c.args[0], c.args[1], c.args[2],
                               dptr(packed, F32, True), dptr(c.totals), dptr(c.row_cum), dptr(c.temp), tb))
    c.host = _pinned_totals_slot()
    c.host.copy_(c.totals, non_blocking=True)
    c.event = torch.cuda.Event()
    c.event.record(torch.cuda.current_stream())
    return c


@torch.no_grad()
def isect3_finish(c, capacity=None):
    """Second half (clmgs_isect3_bin / _bin_dev): tile scan, scatter, per-tile sort.  Same contract as isect2_finish."""
    L = _lib.lib()
    tile_size, tile_width, tile_height, want_isect_ids, want_slots = c.args
    dev, V = c.dev, c.V
    if V == 0:
        c.offsets.zero_()
        e = torch.empty(0, dtype=I32, device=dev)
        res = (e, c.offsets, (torch.empty(0, dtype=I64, device=dev) if want_isect_ids else None))
        return res + ((e, torch.empty(0, dtype=I64, device=dev)),) if want_slots else res
    if capacity is None:
        n_isects, n_ref = isect2_counts(c)
        _record_counts(n_isects, n_ref)
    else:
        n_isects = int(capacity)
    fids = empty_bucketed(n_isects, (), I32, dev)
    ids = empty_bucketed(n_isects, (), I64, dev) if want_isect_ids else None
    sb = L.clmgs_isect3_bin_temp_bytes(n_isects, tile_width * tile_height)
    temp2 = empty_bucketed(sb, (), torch.uint8, dev)
    emit_slot = empty_bucketed(n_isects, (), I32, dev) if want_slots else None
    if capacity is None:
        check(L.clmgs_isect3_bin(stream(), V, n_isects, dptr(c.depths), tile_width, tile_height, dptr(c.row_cum),
                                 dptr(c.temp), dptr(fids), dptr(c.offsets), dptr(ids, I64, True),
                                 dptr(emit_slot, I32, True), dptr(temp2), sb))
    else:
        check(L.clmgs_isect3_bin_dev(stream(), V, n_isects, dptr(c.totals), dptr(c.depths), tile_width, tile_height,
                                     dptr(c.row_cum), dptr(c.temp), dptr(fids), dptr(c.offsets), dptr(ids, I64, True),
                                     dptr(emit_slot, I32, True), dptr(temp2), sb))
    return (fids, c.offsets, ids, (emit_slot, c.row_cum)) if want_slots else (fids, c.offsets, ids)


@torch.no_grad()
def isect_tiles_two_level(means2d, radii, depths, tile_size, tile_width, tile_height,
                          want_isect_ids=False, want_slots=False, packed=None):
    """Single-camera binning through the two-level sort (depth sort of the rows, then one stable
    sort on tile-id bits).  -> (flatten_ids[I] i32, offsets[1,th,tw] i32, isect_ids[I] i64 | None),
    identical to isect_tiles + isect_offset_encode for C = 1.  want_slots: a 4th result
    (emit_slot[I] i32, row_cum[V] i64) for the atomic-free rasterize backward: row i owns the
    contiguous slot range [row_cum[i-1], row_cum[i]).
    packed: the [V,16] raster records -> EXACT per-tile culling (pairs whose tile cannot reach
    alpha >= 1/255 are not emitted; image and gradients unchanged, the list gets ~29 % shorter).
    = isect2_begin + isect2_finish back to back."""
    return isect2_finish(isect2_begin(means2d, radii, depths, tile_size, tile_width, tile_height,
                                      want_isect_ids, want_slots, packed))


# ---------------------------------------------------------------------- rasterize
class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, width, height, tile_size,
                isect_offsets, flatten_ids):
        L = _lib.lib()
        C, N = opacities.shape
        dev = means2d.device
        means2d, conics, colors, opacities = (t.contiguous() for t in (means2d, conics, colors, opacities))
        bg = backgrounds.contiguous() if backgrounds is not None else None
        offsets = isect_offsets.contiguous()
        fids = flatten_ids.contiguous()
        th, tw = offsets.shape[1:]
        out = torch.empty((C, height, width, 3), dtype=F32, device=dev)
        alphas = torch.empty((C, height, width, 1), dtype=F32, device=dev)
        last_ids = torch.empty((C, height, width), dtype=I32, device=dev)
        n_isects = fids.numel()
        packed = torch.empty((C * N, 16), dtype=F32, device=dev)  # one 64 B record per Gaussian
        check(L.clmgs_rasterize_fwd(
            stream(), C, N, n_isects, dptr(means2d, F32), dptr(conics, F32), dptr(colors, F32),
            dptr(opacities, F32), dptr(bg, F32, True), int(width), int(height), int(tile_size), tw,
            th, dptr(offsets, I32), dptr(fids, I32), dptr(packed), dptr(out), dptr(alphas),
            dptr(last_ids)))
        ctx.save_for_backward(packed, bg, offsets, fids, alphas, last_ids)
        ctx.shapes = (means2d.shape, conics.shape, colors.shape, opacities.shape)
        ctx.cfg = (int(width), int(height), int(tile_size))
        return out, alphas

    @staticmethod
    def backward(ctx, v_out, v_alphas):
        L = _lib.lib()
        packed, bg, offsets, fids, alphas, last_ids = ctx.saved_tensors
        width, height, tile_size = ctx.cfg
        s_m2, s_cn, s_col, s_op = ctx.shapes
        C, N = s_op
        th, tw = offsets.shape[1:]
        dev = packed.device
        v_out = v_out.contiguous()
        v_alphas = v_alphas.contiguous() if v_alphas is not None else None
        v_means2d = torch.empty(s_m2, dtype=F32, device=dev)
        v_conics = torch.empty(s_cn, dtype=F32, device=dev)
        v_colors = torch.empty(s_col, dtype=F32, device=dev)
        v_opacities = torch.empty(s_op, dtype=F32, device=dev)
        packed_grad = torch.empty_like(packed)
        check(L.clmgs_rasterize_bwd(
            stream(), C, N, fids.numel(), dptr(packed), dptr(bg, F32, True), width, height,
            tile_size, tw, th, dptr(offsets), dptr(fids), dptr(alphas), dptr(last_ids),
            dptr(v_out, F32), dptr(v_alphas, F32, True), dptr(packed_grad), dptr(v_means2d),
            dptr(v_conics), dptr(v_colors), dptr(v_opacities), None, None, None))
        return v_means2d, v_conics, v_colors, v_opacities, None, None, None, None, None, None


def rasterize_to_pixels(means2d, conics, colors, opacities, image_width, image_height, tile_size,
                        isect_offsets, flatten_ids, backgrounds=None, masks=None, packed=False,
                        absgrad=False):
    """-> (render_colors[C,H,W,3], render_alphas[C,H,W,1]).  backgrounds: None, [3] or [C,3]
    (the reference passes both shapes: no_offload/engine.py:96 vs base_engine.py:189-191)."""
    if packed or absgrad or masks is not None:
        raise NotImplementedError("packed/absgrad/masks are not used by the CLM-GS engines")
    if colors.shape[-1] != 3:
        raise NotImplementedError("3 colour channels only")
    C = opacities.shape[0]
    if backgrounds is not None:
        backgrounds = backgrounds.reshape(-1, 3).to(F32)
        if backgrounds.shape[0] != C:
            backgrounds = backgrounds.expand(C, 3)
    return _Rasterize.apply(means2d, conics, colors, opacities, backgrounds, int(image_width),
                            int(image_height), int(tile_size), isect_offsets, flatten_ids)
