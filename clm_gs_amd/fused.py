"""Engine-internal fast path: one camera forward + loss + backward with NO autograd tape and the
fused front-end kernels (csrc/preprocess.hip).  It computes exactly what the op-by-op chain of
strategies/clm_offload/engine.py:650-742 computes (gather -> activations -> projection -> SH ->
clamp -> isect -> rasterize -> loss -> backward of all of it -> scatter-add -> densification
stats) in 9 launches; the op-by-op path (clm_gs_amd.gsplat / clm_kernels) stays as the API-parity
surface and as the cross-check (tests/test_gpu_engines.py).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib, utils
from ._lib import check, dptr
from .gsplat import (_record_counts, bucket_size, empty_bucketed, isect2_begin, isect2_counts, isect2_finish, isect3_begin,
                     isect3_finish)

F32, I32, U8 = torch.float32, torch.int32, torch.uint8
TILE = 16


def _partial_line_floats():
    """Floats per partial-gradient line of the atomic-free rasterize backward, as the library reports it
    (csrc/common.h PART_F4: 16 = one 64 B line; the 48 B form was measured slower)."""
    global _PART_FLOATS
    if _PART_FLOATS is None:
        _PART_FLOATS = int(_lib.lib().clmgs_rasterize_partials_bytes(1)) // 4
    return _PART_FLOATS


_PART_FLOATS = None


def _cam_host(camera):
    """Host copies of viewmat (row-major world->camera), K and the camera centre, cached."""
    h = getattr(camera, "_clmgs_host", None)
    if h is None:
        vm = camera.world_view_transform.detach().t().contiguous().cpu().numpy().astype(np.float32)
        K = camera.K.detach().cpu().numpy().astype(np.float32) if getattr(camera, "K", None) is not None \
            else camera.create_k_on_gpu().cpu().numpy().astype(np.float32)
        c2w = getattr(camera, "camtoworlds", None)
        if c2w is not None:
            campos = c2w.detach().reshape(-1, 4, 4)[0, :3, 3].cpu().numpy().astype(np.float32)
        else:
            campos = np.linalg.inv(vm.astype(np.float64))[:3, 3].astype(np.float32)
        h = (np.ascontiguousarray(vm.reshape(16)), np.ascontiguousarray(K.reshape(9)),
             np.ascontiguousarray(campos.reshape(3)))
        camera._clmgs_host = h
    return h


def _np(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _CameraPass:
    """Everything one camera's forward leaves behind for its backward (kept alive by the caller
    until the streams involved have consumed it)."""
    __slots__ = ("V", "cam", "filt", "sh_rows", "sh_by_filter", "small_in", "small_packed", "radii",
                 "packed", "fids", "offsets", "emit_slot", "row_cum", "out", "alphas", "last_ids",
                 "bg", "v_out", "maps", "loss", "ev_loss", "streams", "deg", "aux", "loss_partials",
                 "lambda_dssim", "gt_u8", "background", "isect", "sh_index", "means2d", "n_dev")


def _sptr(torch_stream):
    return ctypes.c_void_p(torch_stream.cuda_stream)


# ---- device-side intersection counts (include/clmgs.h "device-count forms").  The reference reads every
# data-dependent size back before it can go on (base_engine.py:64-69, gsplat's cum[-1].item()); here the
# consumers of a camera's intersection list are enqueued against a PREDICTED capacity and read the true count
# on the device, and the host checks the count later (camera_verify, when it enqueues the camera's backward --
# by then the asynchronous readback has long arrived), redoing the camera exactly if the capacity was exceeded.
# Predictor: decaying maximum of the counts seen at this image size, x `isect_capacity_margin`.
_CAPACITY = {}   # (image size, model) -> decaying maximum of the counts seen
_CAP_HELD = {}   # (image size, model) -> the capacity the buffers are currently built for
_CAP_IDS = __import__("itertools").count(1)


def _cap_key(gaussians, W, H):
    """Predictions are per MODEL and image size: a second scene in the same process (a ground-truth renderer, a
    sub-scene of a test) must not inherit the first one's intersection counts."""
    k = getattr(gaussians, "_clmgs_cap_id", None)
    if k is None:
        k = gaussians._clmgs_cap_id = next(_CAP_IDS)
    return (int(W), int(H), k)


def _observe_count(key, n):
    _CAPACITY[key] = max(int(n), int(0.98 * _CAPACITY.get(key, 0)))


def _capacity_for(key, args):
    """Capacity for the next camera at this image size, with hysteresis: the capacity in use is kept while it
    still leaves 10 % over the largest recent count, and a new one is chosen `isect_capacity_margin` (1.25) above it
    -- buffer sizes then change once per ~15 % of growth of the scene's intersection count instead of at every
    1/8-octave bucket boundary, so a training run does not keep calling hipMalloc (20 ms stalls each on a box whose
    memory has not been touched yet: measured as a 90 ms hiccup in the first process of a fresh box)."""
    seen = _CAPACITY.get(key)
    if not seen or not getattr(args, "device_side_counts", True):
        return None
    margin = float(getattr(args, "isect_capacity_margin", 1.25))
    floor = int(getattr(args, "isect_capacity_floor", 4096))
    held = _CAP_HELD.get(key)
    if held is not None and margin > 1.0 and int(seen * 1.10) + floor <= held <= int(seen * margin * 1.5) + floor:
        return held
    cap = int(seen * margin) + floor
    cap = bucket_size(cap) if cap > 4096 else max(cap, 1)
    _CAP_HELD[key] = cap
    return cap


def camera_forward(gaussians, camera, this_filter, sh_rows, sh_by_filter, background, gt_u8,
                   lambda_dssim=0.2, small_packed=None, streams=None, sh_index=None, exact=True):
    """Projection + binning (stream `front`), alpha-blend forward (stream `raster`), loss forward +
    backward (stream `mem`) of one camera; returns the _CameraPass for camera_backward.  The three
    streams may be one and the same; distinct streams are chained by events, so the caller can put
    all cameras' tile kernels on one low-priority stream and order them RF0 RF1 RB0 RF2 RB1 ...
    (software pipelining over the cameras of a batch: the tile stream never waits for a loss).
    = camera_front (nothing blocks) + camera_forward_finish.
    exact=True (default: forward-only callers -- evaluation-style use, parity checks -- get a COMPLETE image): the host
    waits for the intersection count and sizes the lists exactly.  exact=False: lists built for the predicted capacity
    (device-count forms); the render is only valid once camera_verify / camera_backward has checked the count --
    train_one_camera, which runs the backward right after, passes False."""
    return camera_forward_finish(gaussians, camera_front(
        gaussians, camera, this_filter, sh_rows, sh_by_filter, background, gt_u8, lambda_dssim,
        small_packed, streams, sh_index), exact=exact)


def camera_front(gaussians, camera, this_filter, sh_rows, sh_by_filter, background, gt_u8,
                 lambda_dssim=0.2, small_packed=None, streams=None, sh_index=None):
    """Projection and the first half of the binning (depth order, per-row tile counts, asynchronous
    readback of the intersection count) on stream `front`; no host wait."""
    L = _lib.lib()
    args = utils.get_args()
    W, H = int(utils.get_img_width()), int(utils.get_img_height())
    dev = gaussians._xyz.device
    cur = torch.cuda.current_stream()
    s_front, s_mem, s_raster = streams if streams is not None else (cur, cur, cur)
    p = _CameraPass()
    p.streams = (s_front, s_mem, s_raster)
    V = p.V = int(this_filter.shape[0]) if this_filter is not None else int(gaussians._xyz.shape[0])
    p.cam = _cam_host(camera)
    vm, K, campos = p.cam
    deg = p.deg = int(gaussians.active_sh_degree)
    p.sh_rows, p.sh_by_filter, p.small_packed = sh_rows, sh_by_filter, small_packed
    p.sh_index = sh_index  # int32[V]: SH row of position i in a staging table (host-resident mode)
    p.gt_u8, p.lambda_dssim, p.background = gt_u8, float(lambda_dssim), background
    filt = p.filt = this_filter.contiguous() if this_filter is not None else None  # None: all rows
    if small_packed is not None:
        small_in = (dptr(small_packed, F32), None, None, None)
        p.aux = (small_packed,)
    else:
        xyz, opa = gaussians._xyz.detach(), gaussians._opacity.detach()
        sca, rot = gaussians._scaling.detach(), gaussians._rotation.detach()
        small_in = (dptr(xyz, F32), dptr(opa, F32), dptr(sca, F32), dptr(rot, F32))
        p.aux = (xyz, opa, sca, rot)
    p.small_in = small_in
    tw, th = math.ceil(W / float(TILE)), math.ceil(H / float(TILE))
    with torch.cuda.stream(s_front):
        # data-dependent sizes -> bucketed capacity (gsplat.bucket_size): no hipMalloc mid-step
        radii = p.radii = empty_bucketed(V, (), I32, dev).reshape(1, V)
        means2d = empty_bucketed(V, (2,), F32, dev).reshape(1, V, 2)
        depths = empty_bucketed(V, (), F32, dev).reshape(1, V)
        packed = p.packed = empty_bucketed(V, (16,), F32, dev)
        check(L.clmgs_preprocess_fwd(
            _sptr(s_front), V, dptr(filt, torch.int64, True), *small_in,
            dptr(sh_rows, F32, allow_host=True), int(sh_by_filter), _np(vm), _np(K), _np(campos), W, H, deg,
            0.3, 0.01, 1e10, float(getattr(args, "radius_clip", 0.0)), dptr(radii), dptr(means2d),
            dptr(depths), None, None, None, dptr(packed),  # conics/colours/opacities live in `packed`
            dptr(sh_index, I32, True)))
        # binning route: "tile" (default, csrc/isect3.hip: no global sort) | "sort" (csrc/isect.hip: depth sort + tile sort);
        # identical lists (tests/test_gpu_ops.py)
        begin = isect3_begin if getattr(args, "binning", "tile") == "tile" else isect2_begin
        p.isect = begin(means2d, radii, depths, TILE, tw, th, want_slots=True,
                        packed=packed if getattr(args, "exact_tile_cull", True) else None)
        p.aux = p.aux + (means2d, depths)
        p.means2d = means2d  # [1,V,2] pixel centres (parity checks read them; kept alive through p.aux)
    return p


def camera_forward_finish(gaussians, p, exact=False):
    """Second half of the binning, alpha-blend forward, loss forward + backward.  With a capacity prediction
    for this image size (device-side counts, see above) nothing is waited for; otherwise (first cameras of a
    run, `exact`, device_side_counts=False) the host waits for the intersection count and sizes exactly.
    OBLIGATION of the caller when a capacity was used (p.n_dev is not None): the image, loss and lists are built
    from capacity-sized lists whose overflow is DROPPED -- call camera_verify(gaussians, p) (camera_backward does)
    before consuming or accumulating anything of this camera; it redoes the forward exactly if the count exceeded
    the capacity."""
    L = _lib.lib()
    args = utils.get_args()
    W, H = int(utils.get_img_width()), int(utils.get_img_height())
    dev = gaussians._xyz.device
    s_front, s_mem, s_raster = p.streams
    V, packed, background, gt_u8, lambda_dssim = p.V, p.packed, p.background, p.gt_u8, p.lambda_dssim
    tw, th = math.ceil(W / float(TILE)), math.ceil(H / float(TILE))
    cap = None if (exact or V == 0) else _capacity_for(_cap_key(gaussians, W, H), args)
    with torch.cuda.stream(s_front):
        with _lib.host_region("fwd_isect"):
            finish = isect3_finish if p.isect.route == "tile" else isect2_finish
            p.fids, p.offsets, _, (p.emit_slot, p.row_cum) = finish(p.isect, capacity=cap)
        if cap is None:
            p.isect, p.n_dev = None, None
            if V:
                _observe_count(_cap_key(gaussians, W, H), p.fids.numel())
        else:
            p.n_dev = p.isect.totals  # int64[2] on the device: {emitted, reference}; p.isect stays for camera_verify
        p.out = torch.empty((H, W, 3), dtype=F32, device=dev)
        p.alphas = torch.empty((H, W), dtype=F32, device=dev)
        p.last_ids = torch.empty((H, W), dtype=I32, device=dev)
        p.bg = background.reshape(1, 3).to(F32).contiguous() if background is not None else None
    if s_raster is not s_front:
        s_raster.wait_stream(s_front)
    n_isects = p.fids.numel()
    if p.n_dev is None:
        check(L.clmgs_rasterize_fwd(_sptr(s_raster), 1, V, n_isects, None, None, None, None, dptr(p.bg, F32, True),
                                    W, H, TILE, tw, th, dptr(p.offsets), dptr(p.fids), dptr(packed), dptr(p.out),
                                    dptr(p.alphas), dptr(p.last_ids)))
    else:
        check(L.clmgs_rasterize_fwd_dev(_sptr(s_raster), 1, V, n_isects, dptr(p.n_dev), dptr(p.bg, F32, True),
                                        W, H, TILE, tw, th, dptr(p.offsets), dptr(p.fids), dptr(packed), dptr(p.out),
                                        dptr(p.alphas), dptr(p.last_ids)))
    if s_mem is not s_raster:
        ev = torch.cuda.Event()
        ev.record(s_raster)
        s_mem.wait_event(ev)
    with torch.cuda.stream(s_mem):
        # loss forward + backward straight on the [H,W,3] buffer viewed as [3,H,W]
        slots = L.clmgs_loss_slots()
        partials = torch.zeros((slots, 2), dtype=F32, device=dev)
        maps = p.maps = torch.empty((3, 3, H, W), dtype=F32, device=dev)
        sc, sy, sx = 1, 3 * W, 3
        gt = gt_u8.contiguous()
        sm = _sptr(s_mem)
        check(L.clmgs_l1_ssim_loss_fwd(sm, H, W, dptr(p.out), sc, sy, sx, dptr(gt, U8), dptr(partials),
                                       dptr(maps[0]), dptr(maps[1]), dptr(maps[2])))
        # the loss VALUE is not needed by the backward (d loss / d loss = 1): its handful of tiny
        # reduction kernels is enqueued by camera_loss(), off the forward -> backward chain
        p.loss_partials, p.loss = partials, None
        one = getattr(gaussians, "_clmgs_one", None)
        if one is None or one.device != dev:
            one = gaussians._clmgs_one = torch.ones((1,), dtype=F32, device=dev)
        p.v_out = torch.empty_like(p.out)
        check(L.clmgs_l1_ssim_loss_bwd(sm, H, W, dptr(p.out), sc, sy, sx, dptr(gt, U8), dptr(one),
                                       float(lambda_dssim), dptr(maps[0]), dptr(maps[1]), dptr(maps[2]),
                                       dptr(p.v_out)))
        p.ev_loss = None
        if s_raster is not s_mem:
            p.ev_loss = torch.cuda.Event()
            p.ev_loss.record(s_mem)
        p.aux = p.aux + (gt, one, partials)
        # the three derivative maps (573 MB at 4K) were allocated, written and read on s_mem only: dropping
        # them here hands the block back to that stream's pool, where the next camera's maps reuse it in
        # stream order (they used to stay alive until the next batch: 4 x 573 MB)
        p.maps = None
        del maps
    return p


def camera_loss(p):
    """The camera's loss value (0-dim tensor) from the partial sums its forward left; enqueued on
    the CURRENT stream, which must be ordered after the camera's loss kernel."""
    if p.loss is None:
        W, H = int(utils.get_img_width()), int(utils.get_img_height())
        tot = p.loss_partials.sum(dim=0) / float(3 * H * W)
        p.loss = ((1.0 - p.lambda_dssim) * tot[0] + p.lambda_dssim * (1.0 - tot[1])).detach()
    return p.loss


def camera_verify(gaussians, p):
    """Device-count form: the camera's intersection count against the capacity its lists were built for.  Called
    before anything of the camera is accumulated (camera_backward); the readback was issued a whole forward ago.
    Exceeded capacity (a view much denser than any seen so far): the camera's forward is repeated exactly."""
    c = p.isect
    if c is None or p.n_dev is None:
        return
    W, H = int(utils.get_img_width()), int(utils.get_img_height())
    n, n_ref = isect2_counts(c)
    _observe_count(_cap_key(gaussians, W, H), n)
    if n <= p.fids.numel():
        _record_counts(n, n_ref)
        p.isect = None
        return
    _lib.STATS["isect_capacity_redo"] = _lib.STATS.get("isect_capacity_redo", 0) + 1
    for st in set(p.streams):
        st.synchronize()
    p.n_dev = None
    camera_forward_finish(gaussians, p, exact=True)


def camera_backward(gaussians, p, g_sh_rows, small_grad=None, update_stats=True, stats_delta=None,
                    stats_only_visible=False, visibility_out=None, accumulate_after=None, sh_stamp=None,
                    cur_step=0, release=False):
    """Alpha-blend backward (stream `raster`) + projection / SH backward (stream `mem`) of the
    camera whose forward left `p`.  Gradients are ACCUMULATED (see train_one_camera); with `sh_stamp`
    (int32 [N], the deferred optimizer's per-row gradient-step table) + `cur_step` the SH gradient rows
    are STORED on their first touch of the step and stamped (clmgs_preprocess_bwd)."""
    L = _lib.lib()
    args = utils.get_args()
    W, H = int(utils.get_img_width()), int(utils.get_img_height())
    dev = gaussians._xyz.device
    camera_verify(gaussians, p)
    s_front, s_mem, s_raster = p.streams
    V = p.V
    vm, K, campos = p.cam
    tw, th = math.ceil(W / float(TILE)), math.ceil(H / float(TILE))
    n_isects = p.fids.numel()
    if p.small_packed is not None:
        assert small_grad is not None
        small_out = (dptr(small_grad, F32), None, None, None)
    else:
        small_out = (dptr(gaussians._xyz.grad, F32), dptr(gaussians._opacity.grad, F32),
                     dptr(gaussians._scaling.grad, F32), dptr(gaussians._rotation.grad, F32))
    with torch.cuda.stream(s_raster):
        # atomic-free accumulation: the tile kernel stores one 64 B line per intersection (row-ordered
        # slots); clmgs_preprocess_bwd sums each row's contiguous range while it gathers the row.
        # Allocated on the stream of its WRITER (the caching allocator orders reuse by allocation stream).
        partials = empty_bucketed(max(n_isects, 1), (_partial_line_floats(),), F32, dev)
    if p.ev_loss is not None:
        s_raster.wait_event(p.ev_loss)
    if p.n_dev is None:
        check(L.clmgs_rasterize_bwd(_sptr(s_raster), 1, V, n_isects, dptr(p.packed), dptr(p.bg, F32, True), W, H,
                                    TILE, tw, th, dptr(p.offsets), dptr(p.fids), dptr(p.alphas), dptr(p.last_ids),
                                    dptr(p.v_out), None, None, None, None, None, None,
                                    dptr(p.emit_slot), dptr(p.row_cum), dptr(partials)))
    else:
        check(L.clmgs_rasterize_bwd_dev(_sptr(s_raster), 1, V, n_isects, dptr(p.n_dev), dptr(p.packed),
                                        dptr(p.bg, F32, True), W, H, TILE, tw, th, dptr(p.offsets), dptr(p.fids),
                                        dptr(p.alphas), dptr(p.last_ids), dptr(p.v_out), None,
                                        dptr(p.emit_slot), dptr(p.row_cum), dptr(partials)))
    if s_mem is not s_raster:
        ev = torch.cuda.Event()
        ev.record(s_raster)
        s_mem.wait_event(ev)
    stats = update_stats and (not args.disable_auto_densification) and \
        utils.get_cur_iter() <= args.densify_until_iter
    if stats and stats_delta is not None:  # [N,4] delta table (see GaussianModelCLMOffload.stats_delta)
        stat_ptrs = (dptr(stats_delta, F32), None, None)
    else:
        stat_ptrs = (dptr(gaussians.max_radii2D if stats else None, F32, True),
                     dptr(gaussians.xyz_gradient_accum if stats else None, F32, True),
                     dptr(gaussians.denom if stats else None, F32, True))
    with torch.cuda.stream(s_mem):
        if accumulate_after is not None:
            s_mem.wait_event(accumulate_after)
        if visibility_out is not None:
            visibility_out |= (p.radii.reshape(-1) > 0)
        check(L.clmgs_preprocess_bwd(
            _sptr(s_mem), V, dptr(p.filt, torch.int64, True), *p.small_in, dptr(p.sh_rows, F32, allow_host=True),
            int(p.sh_by_filter), _np(vm), _np(K), _np(campos), W, H, p.deg, 0.3, dptr(p.radii),
            None, *small_out, dptr(g_sh_rows, F32, allow_host=True),
            *stat_ptrs, None, int(bool(stats_only_visible)), dptr(partials), dptr(p.row_cum),
            dptr(p.sh_index, I32, True), dptr(sh_stamp, I32, True), int(cur_step)))
    # partials (64 B per intersection, 576 MB at 4K) and the loss cotangent image are dead once the two
    # kernels above have run: hand them back now instead of at the next batch.  Each was used on a second
    # stream (partials: written on s_raster, read on s_mem; v_out: written on s_mem, read on s_raster), so
    # the allocator is told (record_stream) and delays the reuse until that stream has passed this point.
    if s_mem is not s_raster:
        partials.record_stream(s_mem)
        p.v_out.record_stream(s_raster)
    del partials
    p.v_out = None
    if release:
        # Nothing reads this camera's forward outputs after the two kernels above.  They were allocated on the
        # front stream and used on the tile and memory streams: tell the allocator, then drop them, so the next
        # cameras reuse the blocks in stream order instead of the whole batch's outputs staying alive until the
        # next batch (4 x 0.7 GB at 4K).  Only what camera_loss() needs is kept.
        for name in ("radii", "packed", "fids", "offsets", "emit_slot", "row_cum", "out", "alphas", "last_ids", "n_dev"):
            t = getattr(p, name)
            if t is not None:
                for st in {s_mem, s_raster} - {s_front}:
                    t.record_stream(st)
                setattr(p, name, None)
        for t in p.aux:
            if isinstance(t, torch.Tensor) and t.is_cuda and t is not p.loss_partials:
                for st in {s_mem, s_raster} - {s_front}:
                    t.record_stream(st)
        p.aux = (p.loss_partials,)
        p.means2d = None
    return p


def train_one_camera(gaussians, camera, this_filter, sh_rows, sh_by_filter, g_sh_rows, background,
                     gt_u8, lambda_dssim=0.2, update_stats=True, keep=None, accumulate_after=None,
                     return_event=False, stats_only_visible=False, visibility_out=None,
                     raster_stream=None, small_packed=None, small_grad=None, stats_delta=None, sh_index=None,
                     sh_stamp=None, cur_step=0):
    """Forward, loss, backward for one camera over the rows of `this_filter`.

    Gradients are ACCUMULATED into gaussians._xyz/_opacity/_scaling/_rotation .grad (full size, must
    exist) and into g_sh_rows (indexed like sh_rows: by row id when sh_by_filter else by position).
    Returns the detached loss (0-dim tensor).  `keep`, if a list, receives tensors that must stay
    alive until the stream has consumed them.  Everything is enqueued on the CURRENT torch stream;
    `accumulate_after` (an event) gates the gradient-accumulating kernel so two cameras can be in
    flight on two streams while their read-modify-write accumulations stay ordered.
    `raster_stream`: if given, the two ALU-bound tile kernels are enqueued there (event-chained
    with the current stream) so that a shared low-priority stream carries all cameras' tile work
    while the latency-bound kernels of the other camera get CU slots first.
    `small_packed` / `small_grad`: the [N,12] mirror of the four small parameter tensors and the
    packed [N,12] gradient table (GaussianModelCLMOffload.small_packed / small_grad); the kernels
    then gather one 48 B row per Gaussian and accumulate into one, instead of four pieces each."""
    cur = torch.cuda.current_stream()
    p = camera_forward(gaussians, camera, this_filter, sh_rows, sh_by_filter, background, gt_u8,
                       lambda_dssim, small_packed, (cur, cur, raster_stream if raster_stream is not None else cur),
                       sh_index, exact=False)  # verified by camera_backward below
    camera_backward(gaussians, p, g_sh_rows, small_grad, update_stats, stats_delta,
                    stats_only_visible, visibility_out, accumulate_after, sh_stamp, cur_step)
    loss = camera_loss(p)  # on `cur`, which the loss kernels ran on
    if keep is not None:
        keep.append(p)
    if return_event:
        ev = torch.cuda.Event()
        ev.record(cur)
        return loss, ev
    return loss
