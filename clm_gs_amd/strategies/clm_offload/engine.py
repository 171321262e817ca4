"""clm_offload engine (reference: strategies/clm_offload/engine.py:30-979).

``clm_offload_train_one_batch`` keeps the reference signature and return value.  Per batch:
visibility filters for all cameras (one cull pass) -> per micro-batch: gather the visible rows,
render, loss, backward, SH backward in place, scatter the small gradients back -> Adam.

Two residency modes for the [N,48] SH rows and their optimizer state (see gaussian_model.py):

* HBM-resident (MI355X default, `_train_one_batch_hbm`).  Rows are read / accumulated by row id inside HBM by
  the fused front end; the SH-row optimizer is DEFERRED (a row's step of batch b is applied, with the
  zero-gradient steps it skipped, the next time the row is rendered / evaluated / saved: clmgs_adam_catch_up);
  the cameras of a batch run as a software pipeline over three streams by kernel type.
* host-resident (`_train_one_batch_host`, the offloading configuration of the metric).  Pinned host rows; every
  touched row is staged once per direction: helper threads drive the host pool (deferred host row optimizer + copy
  into pinned staging) and ship 48 MB chunks with hipMemcpyAsync on a side stream, grouped by the camera that uses
  a row first; gradient rows go home as zero-copy stores after the camera that uses a row last.  Two staging forms,
  each its own module of four stage functions: per-camera windows (`host_window.py`, the default; `sh_hbm_budget_gb`
  keeps a prefix of the rows resident and stepped in HBM) and the union of the batch (`host_batch.py`).  No retention
  sets, no signal flags, no host Adam thread: the modules' docstrings say what replaced them and why.

Camera order: both modes process the cameras in the order given (the batch gradient does not depend on it and
every touched row crosses the host link once per direction whatever the order).  `order_calculation` -- the
reference's TSP order + last-use groups + retention-set sizes (engine.py:135-298) -- is kept as the
reference-compatible ordering entry and is what `reference_camera_order=True` runs: the cameras are then
processed, and `ordered_cams` / `sparsity` reported, in that order, as the reference does.
"""
import math
import os

import torch

from ... import _lib, clm_kernels, dp, fast_tsp, utils
from ...clm_kernels import (send_shs2cpu_grad_buffer_stream, send_shs2gpu_stream,
                            spherical_harmonics_bwd_inplace)
from ...densification import update_densification_stats_offload_accum_grads
from ...gsplat import (fully_fused_projection, isect_offset_encode, isect_tiles,
                       rasterize_to_pixels, spherical_harmonics)
from ...host import pinned_empty
from ..base_engine import (select_filters, TILE_SIZE, calculate_filters, pipeline_forward_one_step,
                           torch_compiled_loss)

_BITMAP_DTYPE = {4: torch.int8, 8: torch.int8, 16: torch.int16, 32: torch.int32, 64: torch.int64}


def pipeline_forward_one_step_shs_inplace(filtered_opacity_gpu, filtered_scaling_gpu,
                                          filtered_rotation_gpu, filtered_xyz_gpu, filtered_shs,
                                          camera, scene, gaussians, background, pipe_args):
    """One camera over the gathered rows; SH evaluated under no_grad and the colours re-leafed so
    the SH backward can later accumulate in place (engine.py:30-127).
    Returns (image[3,H,W], means2D[1,V,2], radii[1,V], colors_detached[1,V,3], dirs[1,V,3])."""
    viewmat = camera.world_view_transform.transpose(0, 1)
    K = camera.K
    n_selected = filtered_xyz_gpu.shape[0]
    image_width, image_height = int(utils.get_img_width()), int(utils.get_img_height())
    radiis, means2D, depths, conics, _ = fully_fused_projection(
        means=filtered_xyz_gpu, covars=None, quats=filtered_rotation_gpu,
        scales=filtered_scaling_gpu, viewmats=viewmat.unsqueeze(0), Ks=K.unsqueeze(0),
        width=image_width, height=image_height, packed=False)
    means2D.retain_grad()
    dirs = filtered_xyz_gpu[None, :, :] - camera.camtoworlds[:, None, :3, 3]
    filtered_shs = filtered_shs.reshape(1, n_selected, 16, 3)
    with torch.no_grad():
        colors_origin = spherical_harmonics(degrees_to_use=gaussians.active_sh_degree, dirs=dirs,
                                            coeffs=filtered_shs)
    colors_detached = colors_origin.detach().requires_grad_()
    colors = torch.clamp_min(colors_detached + 0.5, 0.0)
    opacities = filtered_opacity_gpu.squeeze(1).unsqueeze(0)
    tile_width = math.ceil(image_width / float(TILE_SIZE))
    tile_height = math.ceil(image_height / float(TILE_SIZE))
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
    return rendered_image, means2D, radiis, colors_detached, dirs


# --------------------------------------------------------------------------- ordering
def _encode_bitmap(filters, n_gaussians, bsz):
    bitmap = torch.zeros((n_gaussians,), dtype=_BITMAP_DTYPE[bsz], device=filters[0].device)
    for i, f in enumerate(filters):  # MSB = first micro-batch (engine.py:150-153)
        clm_kernels.scatter_to_bit(bitmap, f, bsz - 1 - i)
    return bitmap


def finish_groups(filters, n_gaussians, bsz, sparse_adam=False):
    """For filters ALREADY in processing order: the row groups by last use (`update_ls`, engine.py:194-213), the
    sparse_adam visibility mask (:215-222) and the retention-set sizes cnt_h / cnt_d / cnt_g (:224-235).
    -> (groups[bsz+1] device i64, visibility_mask | None, cnt_h, cnt_d, cnt_g (python int lists), bitmap)."""
    dev = filters[0].device
    bitmap = _encode_bitmap(filters, n_gaussians, bsz)
    ffs = torch.empty(n_gaussians, dtype=torch.uint8, device=dev)
    clm_kernels.extract_ffs(bitmap, ffs)
    # group 0: untouched; group k (k>=1): last used in micro-batch k-1  <=> ffs == bsz - (k-1)
    groups = [torch.nonzero(ffs == 0).flatten()]
    for mb in range(bsz):
        groups.append(torch.nonzero(ffs == (bsz - mb)).flatten())
    visibility_mask = (ffs != 0) if sparse_adam else None
    cnt_d = clm_kernels.pair_overlap_count(bitmap, bsz).tolist()
    lens = [len(f) for f in filters]
    cnt_h = [lens[i + 1] - cnt_d[i] for i in range(bsz - 1)]
    cnt_g = [lens[i] - cnt_d[i] for i in range(bsz - 1)]
    return groups, visibility_mask, cnt_h, cnt_d, cnt_g, bitmap


def order_calculation(filters, batched_cameras, n_gaussians, bsz, perm_generator, args):
    """Camera order + row groups by last use + retention-set sizes (engine.py:135-298).

    Returns (finish_indices_filters[bsz+1] pinned-host i32, cameras, filters, sparsity,
    ordered_cams, cnt_h, cnt_d, cnt_g (python int lists, len bsz-1), visibility_mask | None,
    bitmap)."""
    if bsz not in _BITMAP_DTYPE:
        raise ValueError("Currently supported bsz: (4, 8, 16, 32, 64).")
    dev = filters[0].device
    bitmap = _encode_bitmap(filters, n_gaussians, bsz)
    # sampled Hamming distance between the cameras' visibility sets (deterministic stride
    # sample instead of the reference's randperm over N)
    stride = bsz * bsz if bsz >= 32 else 32
    sample = bitmap[::stride].to(torch.int64)
    shifts = torch.arange(bsz - 1, -1, -1, device=dev)
    unz = ((sample[None, :] >> shifts[:, None]) & 1).to(torch.float32)  # [bsz, n_s], row i = mb i
    ones = unz.sum(dim=1)
    inter = unz @ unz.t()
    distance_matrix = (ones[:, None] + ones[None, :] - 2 * inter).round().to(torch.int64).tolist()
    ordered_cams = fast_tsp.find_tour(distance_matrix, 0.001)
    if args.reorder_by_min_sparsity_at_end:
        k_min = bsz - 1
        for k in range(bsz - 1):
            if len(filters[ordered_cams[k]]) < len(filters[ordered_cams[k_min]]):
                k_min = k
        ordered_cams = ordered_cams[k_min + 1:] + ordered_cams[:k_min + 1]
    batched_cameras = [batched_cameras[i] for i in ordered_cams]
    filters = [filters[i] for i in ordered_cams]
    sparsity = [len(f) / float(n_gaussians) for f in filters]

    groups, visibility_mask, cnt_h, cnt_d, cnt_g, bitmap = finish_groups(filters, n_gaussians, bsz, args.sparse_adam)

    # one pinned D2H copy of every index list, as int32 (engine.py:246-260)
    sizes = [g.numel() for g in groups]
    cat = torch.cat(groups).to(torch.int32)
    cat_h = pinned_empty((max(cat.numel(), 1),), dtype=torch.int32)[:cat.numel()]
    cat_h.copy_(cat)
    finish_indices_filters = list(torch.split(cat_h, sizes))
    assert len(finish_indices_filters) == bsz + 1
    assert sum(sizes) == n_gaussians, (sum(sizes), n_gaussians)
    return (finish_indices_filters, batched_cameras, filters, sparsity, ordered_cams, cnt_h, cnt_d,
            cnt_g, visibility_mask, bitmap)


# ------------------------------------------------------------------ shared micro-batch
def _gather_small(gaussians, this_filter):
    idx = this_filter
    xyz = gaussians._xyz.detach()[idx].requires_grad_(True)
    opa = gaussians._opacity.detach()[idx].requires_grad_(True)
    sca = gaussians._scaling.detach()[idx].requires_grad_(True)
    rot = gaussians._rotation.detach()[idx].requires_grad_(True)
    return xyz, opa, sca, rot


def _scatter_small_grads(gaussians, this_filter, xyz, opa, sca, rot):
    with torch.no_grad():
        gaussians._xyz.grad.index_add_(0, this_filter, xyz.grad)
        gaussians._opacity.grad.index_add_(0, this_filter, opa.grad)
        gaussians._scaling.grad.index_add_(0, this_filter, sca.grad)
        gaussians._rotation.grad.index_add_(0, this_filter, rot.grad)


def _render_and_backward(gaussians, scene, camera, background, pipe_args, this_filter, shs,
                         shs_grad, before_sh_backward=None):
    """Forward + loss + backward for one camera on the gathered rows; SH gradients are
    accumulated into shs_grad[V,48]; small gradients are scatter-added into the model's .grad."""
    if getattr(utils.get_args(), "fused_front_end", True):
        from ...fused import train_one_camera
        if before_sh_backward is not None:
            before_sh_backward()
        return train_one_camera(gaussians, camera, this_filter, shs, 0, shs_grad, background,
                                camera.original_image)
    xyz, opa_raw, sca_raw, rot_raw = _gather_small(gaussians, this_filter)
    opa = gaussians.opacity_activation(opa_raw)
    sca = gaussians.scaling_activation(sca_raw)
    rot = gaussians.rotation_activation(rot_raw)
    image, means2D, radiis, colors_detached, dirs = pipeline_forward_one_step_shs_inplace(
        opa, sca, rot, xyz, shs, camera, scene, gaussians, background, pipe_args)
    loss = torch_compiled_loss(image, camera.original_image)
    loss.backward()
    if before_sh_backward is not None:
        before_sh_backward()
    v_dirs = spherical_harmonics_bwd_inplace(
        degrees_to_use=gaussians.active_sh_degree, dirs=dirs, coeffs=shs.reshape(1, -1, 16, 3),
        v_coeffs=shs_grad, v_colors=colors_detached.grad)
    dirs.backward(v_dirs)
    _scatter_small_grads(gaussians, this_filter, xyz, opa_raw, sca_raw, rot_raw)
    update_densification_stats_offload_accum_grads(
        scene, gaussians, int(utils.get_img_height()), int(utils.get_img_width()), this_filter,
        means2D.grad.squeeze(0), radiis.squeeze(0))
    return loss.detach()


def _zero_small_grads(gaussians):
    """Full-size gradient accumulators of the four GPU-resident tensors.  The dense Adam pass
    (_gpu_adam_step) leaves them zeroed and marks them clean, so they are reused batch after
    batch; anything else (first batch, densification replaced the parameter, sparse / frozen
    modes) gets fresh zeros."""
    clean = getattr(gaussians, "_clean_grad_ptrs", ())
    for p in (gaussians._xyz, gaussians._opacity, gaussians._scaling, gaussians._rotation):
        g = p.grad
        if g is None or g.shape != p.shape or g.data_ptr() not in clean:
            p.grad = torch.zeros_like(p)
    gaussians._clean_grad_ptrs = ()


def _gpu_adam_step(gaussians, args, visibility_mask, grad_div=None):
    """grad_div: what the accumulated gradients are divided by (bsz, or bsz x ranks when camera-DP
    summed them over ranks)."""
    grad_div = float(grad_div or args.bsz)
    small = gaussians.all_parameters()[:4]
    if args.stop_update_param or args.sparse_adam:
        for param in small:
            if param.grad is not None:
                param.grad /= grad_div
        if not args.stop_update_param:
            gaussians.optimizer.gpu_adam.step(visibility=visibility_mask)
        gaussians.optimizer.gpu_adam.zero_grad(set_to_none=True)
        return
    # dense: grad / bsz, Adam and gradient zeroing fused into one pass per tensor
    gaussians.optimizer.gpu_step_scaled(1.0 / grad_div)
    gaussians._clean_grad_ptrs = tuple(p.grad.data_ptr() for p in small if p.grad is not None)


# ----------------------------------------------------------------------- HBM-resident
def _pipeline_streams(gaussians):
    """The HIP streams of the camera pipeline, created once per model: `mem` (high priority: front end / loss /
    projection backward), `raster` (low priority: the alpha-blend kernels), `aux`."""
    sts = getattr(gaussians, "_clmgs_streams", None)
    if sts is None:
        try:
            lo, hi = torch.cuda.Stream.priority_range()  # (lowest, highest), e.g. (0, -1)
        except AttributeError:
            lo, hi = 0, -1
        sts = gaussians._clmgs_streams = {
            "aux": torch.cuda.Stream(),
            "mem": [torch.cuda.Stream(priority=hi) for _ in range(2)],
            "raster": [torch.cuda.Stream(priority=lo) for _ in range(2)]}
    return sts


def reserve_working_set(gaussians, n_cameras_in_flight=2):
    """Allocator warm-up of a training run (the product's own form of what bench.py used to do for itself): ONE block per
    stream pool, sized for the per-camera buffers that stream allocates (fused.py) and freed at once, so that the
    caching allocator serves the first batches -- and every later, slightly larger request -- by splitting it instead of
    calling hipMalloc (15-20 ms each in the first process of a fresh box; the 400-image trainer run of round 4 made 45).
    Sizes from the image size and the model size alone; the data-dependent intersection count is taken as the pixel
    count (Rubble 4K slab: 0.6-0.8 of it).  The memory stays RESERVED by the allocator, not allocated: it does not count
    in max_memory_allocated, and on a 288 GB device a few GB of reserve are the cheapest thing there is.
    -> bytes reserved per pool."""
    W, H = int(utils.get_img_width()), int(utils.get_img_height())
    P, N = W * H, int(gaussians._xyz.shape[0])
    I = int(1.0 * P)
    V = min(N, max(1, N // 6))
    k = int(n_cameras_in_flight)
    sts = _pipeline_streams(gaussians)
    plan = {
        # (the lists grow with the model: 1.25x head room, or the growth of iteration ~370 of the bench leg asks the device)
        "front": (sts["mem"][0], k * (20 * P + 70 * I + 96 * V)),      # image / alpha / last ids, lists + sort space, records
        "mem": (sts["mem"][1], k * 48 * P + (1 << 20)),                 # SSIM derivative maps + the loss cotangent
        # one partial-gradient line per intersection; a camera's buffer lives until its row sums are taken (preprocess_bwd,
        # on the other stream type): up to 2 k buffers are alive at once
        "raster": (sts["raster"][0], 2 * k * 64 * int(0.8 * I)),
        # default stream: filters and touched-row lists of a batch, and the temporaries of a densification -- masks,
        # selections, the Z-order keys and their sort, and the per-row tensors (parameters + moments 132 B, packed mirror /
        # gradient / statistics tables 112 B per row), which are re-created BEFORE their predecessors are freed
        "default": (torch.cuda.current_stream(), 240 * N + 64 * V * 4),
    }
    # A convenience, never a requirement: every reservation is capped so that all of them together take at most a third
    # of the memory that is free NOW (a model whose tables nearly fill the device simply runs without the warm-up and
    # pays the hipMallocs it would have paid), and an out-of-memory answer skips the block instead of ending the setup.
    out = {}
    try:
        free_now = torch.cuda.mem_get_info(gaussians._xyz.device)[0]
    except Exception:  # noqa: BLE001
        free_now = None
    total = sum(int(nb) for _, nb in plan.values())
    shrink = 1.0 if not free_now or total <= free_now / 3 else (free_now / 3) / total
    for name, (st, nbytes) in plan.items():
        nbytes = int(nbytes * shrink)
        if nbytes < (1 << 20):
            out[name] = 0
            continue
        try:
            with torch.cuda.stream(st):
                blk = torch.empty((nbytes,), dtype=torch.uint8, device=gaussians._xyz.device)
                # ... and the SMALL pool of the stream (requests <= 1 MB are carved from 2 MB segments of their own: the
                # counters, count read-backs and per-camera scalars of a batch; round 5's trainer leg still made 8 such
                # hipMallocs in its first iterations): 8 MB = four segments, held together so that four are created
                small = [torch.empty((1 << 19,), dtype=torch.uint8, device=gaussians._xyz.device) for _ in range(16)]
                del blk, small
            out[name] = nbytes
        except torch.cuda.OutOfMemoryError:
            torch.cuda.empty_cache()
            out[name] = 0
    return out


class _Batch:
    """State of one HBM-resident batch, handed from stage to stage (see _train_one_batch_hbm)."""


def _stage_visibility(b):
    """Stage 1: (camera-DP step S) + the batch's visibility filters and the union of touched rows, selected on the GPU."""
    gaussians, args, cams = b.gaussians, b.args, b.cameras
    N = b.N
    b.touched = b.touched_rows = None
    # camera-DP, small attributes at their owners (gaussian_model.small_owner): step S -- the current values of every
    # foreign row that may be visible this batch arrive BEFORE the visibility pass reads them
    b.small_owner = bool(getattr(gaussians, "small_owner", False) and gaussians.lazy_rows and dp.active()
                         and not args.stop_update_param)
    if b.small_owner:
        with torch.no_grad(), _lib.host_region("dp_small_fetch"), dp.phase("S"):
            gaussians.small_prepare(cams)
    # single GPU, small attributes stepped per block (gaussian_model.small_deferred): the blocks this batch's cameras may
    # see are brought up to date BEFORE the visibility pass reads them
    b.small_deferred = bool(getattr(gaussians, "small_deferred", False) and b.fused and gaussians.lazy_rows
                            and not args.stop_update_param and getattr(args, "packed_small", True))
    blk_flags = None
    if b.small_deferred:
        with torch.no_grad():
            blk_flags = gaussians.small_catch_up(cams)  # (also: which blocks of 256 rows can hold a visible row at all)
    with torch.no_grad():
        if b.fused:
            # same fast exp as the fused front end -> filter and render agree on every cull;
            # filters AND the union of touched rows are selected on the GPU in one pass
            with _lib.host_region("select_filters"):
                b.filters, b.touched_rows = select_filters(cams, gaussians._xyz.detach(), gaussians._scaling.detach(),
                                                           gaussians._rotation.detach(), block_flags=blk_flags)
        else:
            b.filters, _, _ = calculate_filters(cams, gaussians.get_xyz, gaussians.get_opacity,
                                                gaussians.get_scaling, gaussians.get_rotation)
    if b.small_owner and os.environ.get("CLMGS_DP_DEBUG") and b.touched_rows is not None:
        # debug: every foreign row the exact pass selected must have been a candidate of step S
        lo_, hi_ = dp.owner_range(N)
        foreign = b.touched_rows[(b.touched_rows < lo_) | (b.touched_rows >= hi_)]
        cand_ = getattr(gaussians, "_dbg_cand", None)
        miss = -1
        if cand_ is not None and gaussians._small_since > 0:
            isc = torch.zeros((N,), dtype=torch.bool, device=foreign.device)
            isc[cand_] = True
            miss = int((~isc[foreign]).sum())
        print("DPDEBUG batch rank %d since %d foreign touched rows %d not candidates %d" % (
            dp.rank(), gaussians._small_since, foreign.numel(), miss), flush=True)
        assert miss <= 0
    b.sparsity = [len(f) / float(N) for f in b.filters]
    b.ordered_cams = list(range(b.bsz))
    if getattr(args, "reference_camera_order", False):  # the reference's TSP order (engine.py:135-298)
        _, b.cameras, b.filters, b.sparsity, b.ordered_cams = order_calculation(
            list(b.filters), list(cams), N, b.bsz, None, args)[:5]


def _stage_plan(b):
    """Stage 2: which optimizer / exchange form this batch runs in; the touched mask where a form needs it; the row
    optimizer's step counter."""
    gaussians, args, N, bsz = b.gaussians, b.args, b.N, b.bsz
    b.lazy = gaussians.lazy_rows and not args.stop_update_param
    mode = getattr(args, "overlap_cameras", True)
    b.pipelined = b.fused and mode in (True, "pipeline")
    # packed [N,12] mirror + packed gradient table for the four small tensors (dense fused path)
    b.use_packed = (b.fused and getattr(args, "packed_small", True) and not args.sparse_adam
                    and not args.stop_update_param)
    # camera-DP, locality exchange (dp.py): every rank works on ITS cameras' rows; no global touched mask
    b.locality = bool(b.lazy and dp.active() and getattr(args, "dp_locality", False))
    # ... and its sparse_adam form (config 5 as the reference scripts it: SelectiveAdam for the small attributes, the
    # SH rows of visible Gaussians only): eager row optimizer at the owner, clearing-policy gradient tables
    b.locality_sparse = bool(dp.active() and getattr(args, "dp_locality", False) and args.sparse_adam and b.fused
                             and not args.stop_update_param and not b.lazy)
    if getattr(args, "dp_locality", False) and dp.active():
        assert b.touched_rows is not None and (b.locality_sparse or (b.locality and b.use_packed and gaussians.first_touch_grads)), (
            "dp_locality needs the fused front end and either the dense deferred row optimizer with the packed "
            "small-attribute tables and first-touch gradient stores (defaults) or sparse_adam")
    need_mask = b.touched_rows is None or args.sparse_adam or (dp.active() and not b.locality) or not b.lazy
    if need_mask:
        b.touched = torch.zeros((N,), dtype=torch.bool, device=gaussians._xyz.device)
        if b.touched_rows is not None:  # index_fill_: scalar as kernel argument, no blocking H2D copy
            utils.fill_rows(b.touched, b.touched_rows, True)
        else:
            for f in b.filters:
                utils.fill_rows(b.touched, f, True)
        # camera-DP: rows touched by ANY rank get their (reduced) gradient at the end of the batch;
        # only globally untouched rows may take the early zero-gradient update
        if dp.active() and not b.locality_sparse:  # (locality: the global mask is assembled from what the owners publish)
            b.local_touched = b.touched          # this rank's cameras' rows (mask), before the OR over the ranks
            b.touched = dp.allreduce_touched(b.touched)
            b.touched_rows = None
    b.row_adam = gaussians.optimizer.cpu_adam
    b.params = gaussians._parameters
    b.grad_buf = b.parameters_grad_buffer[:N]
    b.params.grad = b.grad_buf
    b.row_adam.global_step += 1
    b.step = b.row_adam.global_step
    _mark_batch(gaussians)
    b.group = b.row_adam.param_groups[0]
    b.st = b.row_adam.state[b.params]
    b.default_stream = torch.cuda.current_stream()
    b.side_event = None
    b.col_lr = b.row_adam._col_lr(b.params.device)
    if b.touched_rows is None:
        b.touched_rows = torch.nonzero(b.touched).flatten()
    b.touched_rows = b.touched_rows.to(torch.int32)
    _lib.STATS.setdefault("touched_rows", []).append(int(b.touched_rows.shape[0]))  # shape known on the host
    # first-touch gradient stores (gaussian_model.first_touch_grads): the projection/SH backward stamps
    # `_row_g_step` itself and stores instead of accumulating on a row's first touch of this step
    b.ft_stamp = gaussians._row_g_step if (b.lazy and b.fused and gaussians.first_touch_grads) else None


def _mark_batch(gaussians):
    """Every engine path that advances the row optimizer's step counter calls this: deferred row steps will be waiting
    (flush_lazy_rows clears the flag), and positions move -- "the rows are in Z-order of their current positions" ends
    here (gaussian_model.spatial_sort keys its shortcut on the mutation counter)."""
    gaussians._lazy_dirty = True
    gaussians._mutations = getattr(gaussians, "_mutations", 0) + 1
    gaussians._sorted_tag = None


def _row_update(b, rows, zero_grad_rows=False):
    # explicit row lists: the kernel walks |rows| x 48 elements instead of scanning all N rows;
    # rows the batch never touches have an all-zero gradient -> the grad buffer is not read
    clm_kernels.adam_rows(b.params.data, None if zero_grad_rows else b.grad_buf, b.st["exp_avg"],
                          b.st["exp_avg_sq"], rows, b.col_lr,
                          b.group["betas"][0], b.group["betas"][1], b.group["eps"], b.step,
                          b.group["bias_correction"], 1.0 / (b.bsz * dp.world_size()), True)


def _stage_exchange_head(b):
    """Stage 3, before rendering: the rows the batch renders from are brought up to date (deferred row optimizer) and,
    camera-DP, fetched from their owners (steps A + B of the locality exchange / the all-gather of owner-computes)."""
    gaussians, args, N, bsz, step = b.gaussians, b.args, b.N, b.bsz, b.step
    params, touched_rows = b.params, b.touched_rows
    b.owner = b.border = None  # owner-computes camera-DP (dp.py): rows owned by index range
    b.split_catch = False
    b.dp_ev_b0 = b.dp_ev_b1 = b.dp_comm = None
    b.dp_split = False
    if b.locality:
        # A + B of the locality exchange: who needs which of my rows; bring every own row anybody renders from
        # up to date (waiting gradient step + replays); parameter rows out to the ranks that asked for them.
        # OVERLAP (round 4, dp_overlap): under the camera pipeline the exchange is split (dp.border_plan parts) --
        #   B  parameters of the FIRST camera's border rows travel first, the (larger) rest while camera 0 renders:
        #      both on a side stream, the front stream waits for an event per part;
        #   D  gradient lines of the border rows the LAST camera does not touch leave right after the second-to-last
        #      backward, i.e. under the last camera's backward (side stream, read-only on rows nobody writes any more);
        #      the rest, the owner-side accumulation and F (the owners' published sums) stay at the tail.
        b.dp_split = bool(b.pipelined and bsz >= 2 and getattr(args, "dp_overlap", True))
        with _lib.host_region("dp_border_plan"), dp.phase("plan"):
            b.border = dp.border_plan(touched_rows.long(), N, first_rows=b.filters[0] if b.dp_split else None,
                                      last_rows=b.filters[bsz - 1] if b.dp_split else None,
                                      publish_counts=not b.small_owner)
        gaussians._owner_dirty = True  # replicas are partial until the next flush_lazy_rows()
        own_rows = b.border.own_rows
        if own_rows.numel():
            with dp.phase("catch_up_own"):
                gaussians.catch_up_rows(own_rows.to(torch.int32), to_step=step - 1)
        if b.dp_split:
            b.dp_comm = getattr(gaussians, "_dp_comm_stream", None)
            if b.dp_comm is None:
                b.dp_comm = gaussians._dp_comm_stream = torch.cuda.Stream()
            b.dp_comm.wait_stream(b.default_stream)  # the owners' rows are current
            with torch.cuda.stream(b.dp_comm):
                with dp.phase("B0"):
                    dp.border_params_out(params.data, b.border, "params0")
                b.dp_ev_b0 = torch.cuda.Event()
                b.dp_ev_b0.record(b.dp_comm)
                with dp.phase("B1"):
                    dp.border_params_out(params.data, b.border, "params1")
                b.dp_ev_b1 = torch.cuda.Event()
                b.dp_ev_b1.record(b.dp_comm)
        else:
            with dp.phase("B"):
                dp.border_params_out(params.data, b.border)
    elif b.locality_sparse:
        with _lib.host_region("dp_border_plan"), dp.phase("plan"):
            b.border = dp.border_plan(touched_rows.long(), N)
        gaussians._owner_dirty = True
        with dp.phase("B"):
            dp.border_params_out(params.data, b.border)  # the owners' rows are always current (eager row optimizer)
    elif b.lazy and dp.active() and getattr(args, "dp_owner_computes", False):
        b.owner = dp.owner_plan(touched_rows.long(), N)
        gaussians._owner_dirty = True  # replicas are partial until the next flush_lazy_rows()
        own_rows = touched_rows[b.owner.lo:b.owner.hi]
        if b.owner.hi > b.owner.lo:  # waiting gradient step + replays, on the rows this rank owns
            gaussians.catch_up_rows(own_rows, to_step=step - 1)
        dp.owner_gather_rows(params.data, b.owner)  # every rank renders from the owners' current rows
    elif b.lazy:
        # deferred dense Adam: rows this batch renders replay the zero-gradient steps they skipped
        # (exactly the updates the eager pass would have streamed through HBM every batch);
        # untouched rows are not visited at all.
        if (b.pipelined and bsz >= 2 and getattr(args, "split_catch_up", True) and b.filters is not None):
            # (all-reduce camera-DP, round 6: the rows only OTHER ranks touch must consume their waiting gradient before
            #  this batch's sum lands on them -- they get one extra pass of their own, enqueued by _cameras_pipelined
            #  behind the last camera's rows, i.e. underneath the tile kernels and ahead of the tail exchange)
            b.foreign_rows = None
            if dp.active():
                lt = getattr(b, "local_touched", None)
                assert lt is not None
                b.foreign_rows = touched_rows[~lt[touched_rows.long()]]
            # CAMERA BY CAMERA (round 5; enqueued by _cameras_pipelined on the front stream, each call right before its
            # camera's projection): the first camera's chain waits for ITS rows only (0.8 ms of the 2.7 ms pass at 28 M
            # rows), the other cameras' rows are brought up to date underneath the tile kernels of the cameras before them
            b.split_catch = True
            s_front = _pipeline_streams(gaussians)["mem"][0]
            s_front.wait_stream(b.default_stream)
            b.first_catch_done = bool(getattr(args, "early_first_catch_up", True))
            if b.first_catch_done and b.filters[0].numel():  # (issued here, ahead of the host work of the camera stage)
                with torch.cuda.stream(s_front):
                    gaussians.catch_up_rows(b.filters[0], to_step=step - 1)
        else:
            gaussians.catch_up_rows(touched_rows, to_step=step - 1)
    elif not args.stop_update_param and not args.sparse_adam:
        # rows this batch never touches: zero gradient, pure momentum decay -> overlap with render
        untouched_rows = torch.nonzero(~b.touched).flatten().to(torch.int32)
        b.comm_stream.wait_stream(b.default_stream)
        with torch.cuda.stream(b.comm_stream):
            _row_update(b, untouched_rows, zero_grad_rows=True)
            b.side_event = torch.cuda.Event()
            b.side_event.record(b.comm_stream)
        untouched_rows.record_stream(b.comm_stream)


def _cameras_pipelined(b):
    """Software pipeline over the cameras of the batch, streams by kernel TYPE:
      front  (high priority): projection + binning of camera k, one camera ahead
      mem    (high priority): loss of camera k, projection/SH backward of camera k-1
      raster (low priority) : the ALU-bound tile kernels, enqueued RF0 RF1 RB0 RF2 RB1 ...
    so the tile stream always has the next forward to run while a loss is being computed, and the latency-bound kernels
    fill the memory system underneath it.  Measured-out variants of this schedule (a CU mask on the tile stream, a
    second front stream running a camera ahead of the host's count wait, two forwards ahead of the oldest backward,
    all streams at one priority, one stream pair per camera) are recorded in DESIGN.md section 6 and gone from the code."""
    from ...fused import camera_backward, camera_forward_finish, camera_front, camera_loss
    gaussians, bsz, step = b.gaussians, b.bsz, b.step
    sts = _pipeline_streams(gaussians)
    s_front, s_mem, s_raster = sts["mem"][0], sts["mem"][1], sts["raster"][0]
    if getattr(gaussians, "_clmgs_one", None) is None:  # the loss cotangent (1.0), made on the
        gaussians._clmgs_one = torch.ones((1,), dtype=torch.float32, device=b.params.device)  # default stream
    for st_ in (s_front, s_mem, s_raster):
        st_.wait_stream(b.default_stream)
    # the previous batch's per-camera tensors: every stream that read them has been joined
    # into the default stream, which the three streams now wait for -> safe to recycle
    gaussians._clmgs_passes = None
    passes = []
    dp_recv0 = [None]

    def backward(k):
        with _lib.host_region("camera_backward"):
            camera_backward(gaussians, passes[k], b.grad_buf, b.small_gk, stats_delta=b.stats_d,
                            sh_stamp=b.ft_stamp, cur_step=step, release=True)
        # camera-DP, locality exchange: the gradient lines of the border rows the last camera does not touch are
        # final once camera bsz-2's backward is enqueued -- they travel now, under the last camera's backward
        if b.dp_split and k == bsz - 2:
            ev = torch.cuda.Event()
            ev.record(s_mem)
            b.dp_comm.wait_event(ev)
            with torch.cuda.stream(b.dp_comm), dp.phase("D0"):
                dp_recv0[0] = dp.border_grads_send([b.grad_buf, b.small_gk], b.ft_stamp, step, b.border, "grads0")

    # Deferred row steps camera by camera (_stage_exchange_head), on the FRONT stream, each right before its camera's
    # projection: catch(0) front(0) catch(1) front(1) ... -- camera k+1's rows are brought up to date while camera k's
    # forward tile kernel runs, and the first camera waits for its own rows only.  A row seen by several cameras is
    # stepped by the first list that holds it -- the wrapper stamps `_row_last_step` after every call, in stream order,
    # later lists find the row current and skip it -- so every row takes exactly the step the single pass over the union
    # gave it.  No camera's backward touches a row an unfinished call may still write: a camera's rows are current, and
    # skipped by all later calls, before its chain starts.  (On a stream of its own the calls shared a hardware queue
    # with the loss kernels and ran beside the first camera's projection, which took 1.16 ms instead of 0.27.)
    def catch(k):
        if b.filters[k].numel():
            with torch.cuda.stream(s_front):
                gaussians.catch_up_rows(b.filters[k], to_step=step - 1)
    for k in range(bsz):
        if b.dp_split:  # parameters of this camera's border rows have landed (part 0: camera 0, part 1: the rest)
            s_front.wait_event(b.dp_ev_b0 if k == 0 else b.dp_ev_b1)
        if b.split_catch and (k or not b.first_catch_done):  # (camera 0: usually done by _stage_exchange_head)
            catch(k)
        if b.split_catch and k == bsz - 1 and getattr(b, "foreign_rows", None) is not None and b.foreign_rows.numel():
            with torch.cuda.stream(s_front):  # all-reduce camera-DP: the rows only other ranks touch (see _stage_exchange_head)
                gaussians.catch_up_rows(b.foreign_rows, to_step=step - 1)
        with _lib.host_region("camera_front"):
            cur_pass = camera_front(gaussians, b.cameras[k], b.filters[k], b.params.data, 1, b.background,
                                    b.cameras[k].original_image, small_packed=b.small_pk,
                                    streams=(s_front, s_mem, s_raster))
        with _lib.host_region("camera_forward"):
            passes.append(camera_forward_finish(gaussians, cur_pass))
        if k >= 1:  # the forward runs one camera ahead of the backward
            backward(k - 1)
    backward(bsz - 1)
    if b.dp_split:
        b.default_stream.wait_stream(b.dp_comm)
        gaussians._dp_recv0 = dp_recv0[0]
    for st_ in (s_front, s_mem, s_raster):
        b.default_stream.wait_stream(st_)
    losses = [camera_loss(p_) for p_ in passes]  # loss values: after the join, off the chain
    gaussians._clmgs_passes = passes  # released at the start of the next batch (see above)
    return losses


def _stage_cameras(b):
    """Stage 4: every camera of the batch -- forward, loss, backward; gradients accumulate by row id."""
    gaussians, args, bsz = b.gaussians, b.args, b.bsz
    b.small_pk = b.small_gk = b.stats_d = None
    if b.use_packed:
        b.small_pk, b.small_gk = gaussians.small_packed(), gaussians.small_grad()
        if (getattr(args, "packed_stats", True) and (not args.disable_auto_densification)
                and utils.get_cur_iter() <= args.densify_until_iter):
            b.stats_d = gaussians.stats_delta()
    else:
        _zero_small_grads(gaussians)
    if b.pipelined:
        return _cameras_pipelined(b)
    losses = []
    if b.fused:  # overlap_cameras=False: one camera after the other on the current stream
        from ...fused import train_one_camera
        for k in range(bsz):
            losses.append(train_one_camera(
                gaussians, b.cameras[k], b.filters[k], b.params.data, 1, b.grad_buf, b.background,
                b.cameras[k].original_image, small_packed=b.small_pk, small_grad=b.small_gk,
                stats_delta=b.stats_d, sh_stamp=b.ft_stamp, cur_step=b.step))
        return losses
    for k in range(bsz):  # op-by-op path (fused_front_end=False): the reference's chain of gsplat / clm_kernels calls
        this_filter = b.filters[k]
        with torch.no_grad():
            shs = torch.empty((this_filter.shape[0], 48), device=b.params.device)
            send_shs2gpu_stream(shs, b.params.data, this_filter)
            shs_grad = torch.zeros_like(shs)
        loss = _render_and_backward(gaussians, b.scene, b.cameras[k], b.background, b.pipe_args, this_filter, shs,
                                    shs_grad)
        with torch.no_grad():
            send_shs2cpu_grad_buffer_stream(shs_grad, b.grad_buf, this_filter, True)
        losses.append(loss)
    return losses


def _stage_exchange_tail(b):
    """Stage 5 (camera-DP only): the one gradient exchange of the batch (sums; 1/ranks rides on the Adam gradient
    scale, so no tensor is touched just to be divided)."""
    if not dp.active():
        return
    gaussians, N, step = b.gaussians, b.N, b.step
    grad_buf, small_gk, ft_stamp, border = b.grad_buf, b.small_gk, b.ft_stamp, b.border
    if border is not None and b.locality_sparse:
        # D + F, clearing policy: SH gradient rows and the four small gradients of the border rows are added
        # at their owners; the owners publish the summed small gradients of every row anybody touched, which
        # also tells every rank the global visibility set SelectiveAdam steps
        small_grads = [gaussians._xyz.grad, gaussians._opacity.grad, gaussians._scaling.grad, gaussians._rotation.grad]
        dp.border_grads_home([grad_buf] + small_grads, None, 0, border)
        own_rows = border.own_rows
        _, got = dp.publish_rows(small_grads, own_rows, N, counts=border.own_counts)
        b.touched = torch.zeros((N,), dtype=torch.bool, device=b.params.device)
        for ids in [own_rows] + got:
            if ids.numel():
                utils.fill_rows(b.touched, ids, True)
        b.touched_rows = own_rows.to(torch.int32)  # the SH rows THIS rank steps
    elif border is not None:
        # D + F of the locality exchange: border rows' gradient lines (SH row | packed small row) go home to
        # their owners; the owners publish the summed small-attribute gradients of their touched rows
        if b.dp_split:
            r0 = gaussians._dp_recv0
            gaussians._dp_recv0 = None
            for t_ in (r0[0], r0[1], r0[3]):  # allocated on the side stream, consumed here
                if isinstance(t_, torch.Tensor) and t_.is_cuda:
                    t_.record_stream(b.default_stream)
            with dp.phase("D1"):
                r1 = dp.border_grads_send([grad_buf, small_gk], ft_stamp, step, border, "grads1")
            with dp.phase("D_apply"):
                dp.border_grads_apply([grad_buf, small_gk], ft_stamp, step, border, r0)
                dp.border_grads_apply([grad_buf, small_gk], ft_stamp, step, border, r1)
        else:
            with dp.phase("D"):
                dp.border_grads_home([grad_buf, small_gk], ft_stamp, step, border)
        if not b.small_owner:  # (small_owner: the summed lines are home, and only the owner steps the row)
            dp.publish_small(small_gk, ft_stamp, step, N, border)
    elif b.owner is not None:
        # small gradients: all-reduce over the touched rows (their dense Adam stays replicated: the
        # next batch's visibility pass needs every row's xyz / scale / rotation on every rank);
        # SH gradient rows: summed AT THEIR OWNER, which alone will step them
        if b.use_packed:
            dp.allreduce_tables_rows([small_gk], b.touched_rows, N, average=False)
        else:
            dp.allreduce_small_grads([gaussians._xyz.grad, gaussians._opacity.grad,
                                      gaussians._scaling.grad, gaussians._rotation.grad], average=False)
        dp.owner_reduce_rows(grad_buf, b.owner)
    elif b.use_packed:
        # one collective: packed small gradients + SH gradient rows of the globally touched set
        with dp.phase("allreduce"):
            dp.allreduce_tables_rows([small_gk, grad_buf], b.touched_rows, N, average=False)
    else:
        dp.allreduce_small_grads([gaussians._xyz.grad, gaussians._opacity.grad,
                                  gaussians._scaling.grad, gaussians._rotation.grad], average=False)
        dp.allreduce_rows(grad_buf, b.touched, average=False, rows=b.touched_rows)


def _stage_optimizer(b):
    """Stage 6: the small attributes' Adam step now; the SH rows' step now (eager modes) or deferred to their next touch."""
    gaussians, args, N, bsz, step = b.gaussians, b.args, b.N, b.bsz, b.step
    if b.use_packed and b.small_deferred and b.ft_stamp is not None and not dp.active():
        gaussians.small_defer_record(step)  # applied block by block when a camera comes near (small_catch_up)
    elif b.use_packed:
        if getattr(gaussians, "small_deferred", False):
            gaussians.flush_small()  # (a batch in another mode: nothing may wait across it)
        gaussians.optimizer.gpu_step_packed(b.small_pk, b.small_gk, 1.0 / (bsz * dp.world_size()),
                                            g_stamp=b.ft_stamp, cur_step=step,
                                            row_range=dp.owner_range(N) if (b.small_owner and b.locality) else None)
        if b.small_owner and b.locality:
            gaussians.small_after_step()
    else:
        _gpu_adam_step(gaussians, args, b.touched if args.sparse_adam else None,
                       grad_div=bsz * dp.world_size())
        gaussians.invalidate_small_packed()
    if b.lazy:
        # DEFERRED: the touched rows' Adam step of this batch is not run now.  Their (reduced) gradient
        # rows stay in the gradient table, stamped with this step, and catch_up_rows applies them -- at
        # this step, before the zero-gradient replays -- the next time a row is rendered / evaluated /
        # saved / densified: p, m, v of a row make one round trip per touch instead of two.
        # (index_fill_ takes the scalar as a kernel argument; `t[rows] = step` would copy a host scalar
        # to the device and block the host until the whole batch has drained)
        stamp = b.touched_rows[b.owner.lo:b.owner.hi] if b.owner is not None else b.touched_rows
        if stamp.numel() and b.ft_stamp is None:  # first-touch mode: the backward kernels stamped their rows
            utils.fill_rows(gaussians._row_g_step, stamp.long(), step)
    elif not args.stop_update_param:
        _row_update(b, b.touched_rows)
    b.st["step"] = step


def _train_one_batch_hbm(gaussians, scene, batched_cameras, parameters_grad_buffer, background,
                         pipe_args, comm_stream, args):
    """One batch with the SH rows and their optimizer state in HBM, in six stages (each a function above):
    visibility -> plan -> exchange head (camera-DP: parameter rows in; deferred row steps of the rows rendered from)
    -> cameras (software pipeline over three kernel-type streams) -> exchange tail (camera-DP: gradient lines home)
    -> optimizer (small attributes now, SH rows deferred)."""
    b = _Batch()
    b.gaussians, b.scene, b.cameras, b.args = gaussians, scene, batched_cameras, args
    b.parameters_grad_buffer, b.background, b.pipe_args, b.comm_stream = parameters_grad_buffer, background, pipe_args, comm_stream
    b.bsz, b.N = len(batched_cameras), gaussians._xyz.shape[0]
    b.fused = bool(getattr(args, "fused_front_end", True))
    _stage_visibility(b)
    _stage_plan(b)
    _stage_exchange_head(b)
    losses = _stage_cameras(b)
    with dp.phase("tail_exchange"):  # (bench.py dp.phase_ms: what the exchange adds after the last backward)
        _stage_exchange_tail(b)
    if getattr(args, "debug_skip_optimizer", False):
        # test hook: the batch ran exactly as in production (packed tables, lazy catch-up, DP exchange)
        # but no optimizer consumes the accumulated gradients; they stay in parameters_grad_buffer[:N],
        # the packed small-gradient table / the four .grad tensors, UNSCALED (sum over the cameras)
        b.row_adam.global_step -= 1
        if b.side_event is not None:
            b.default_stream.wait_event(b.side_event)
        return losses, b.ordered_cams, b.sparsity
    _stage_optimizer(b)
    if b.side_event is not None:
        b.default_stream.wait_event(b.side_event)
    # No device synchronisation here: everything above is ordered on the default stream, so the
    # host can already prepare the next batch (its first host wait is the filter sizes) while the
    # optimizer kernels run; callers that read the losses synchronise by doing so.
    if getattr(args, "sync_each_batch", False):
        torch.cuda.synchronize()
    return losses, b.ordered_cams, b.sparsity


# ---------------------------------------------------------------------- host-resident
def _bit_of(bitmap, ids, bit):
    return ((bitmap[ids].to(torch.int64) >> bit) & 1).to(torch.bool)


_HOST_CHUNK_ROWS = 262144  # staging granularity: 48 MB of parameter rows per hipMemcpyAsync


def _host_tables(gaussians, dev):
    """Row-indexed scratch of the host-resident mode (independent of the batch size): row -> slot, two row masks."""
    N = gaussians._xyz.shape[0]
    ht = getattr(gaussians, "_host_tabs", None)
    if ht is None or ht["N"] != N:
        ht = gaussians._host_tabs = dict(N=N, slot_of=torch.zeros((N,), dtype=torch.int32, device=dev),
                                         mark=torch.zeros((N,), dtype=torch.bool, device=dev),
                                         in_spec=torch.zeros((N,), dtype=torch.bool, device=dev))
    return ht


def _host_buffers(gaussians, T, dev):
    """Per-batch buffers of the host-resident mode, kept between batches (bucketed capacity):
    pinned row lists + pinned staging rows on the host (one set for the batch's late rows, one for the
    speculative rows of the next batch), TWO staging parameter tables on the GPU (the batch in flight renders
    from one while the next batch's speculative rows land in the other) and the gradient staging table.
    `gen` counts re-allocations: rows staged into an older generation are gone."""
    from ...gsplat import bucket_size
    hb = getattr(gaussians, "_host_bufs", None)
    N = gaussians._xyz.shape[0]
    if hb is None or hb["cap"] < T or hb["N"] != N:
        # Two capacities.  DEVICE tables (they count against the GPU peak of the offloading mode): 8 % over the need,
        # grown when a batch exceeds them.  PINNED host twins: 50 % over the need -- host memory is cheap, and a new
        # pinned table is a hipHostMalloc of 2 GB at 28 M (300+ ms: ONE such re-allocation inside a 20-batch run was
        # 21 of the 24 ms of average "host_groups" time rounds 2-3 reported); they are kept when only the device tables
        # grow.  The union of a batch's filters varies by several percent from batch to batch (shuffled cameras), and
        # staged-but-unused rows of a hinted batch ride along.
        cap = bucket_size(max(int(T * 1.08), 1))
        gen = hb["gen"] + 1 if hb else 0
        keep_pinned = None
        if hb is not None:
            # The staging tables are written by hipMemcpyAsync issued through ctypes on the side streams (speculative
            # prefetch, feeder): the caching allocator knows nothing of that use, and a dropped speculation is only
            # JOINED (its last copy may still be in flight).  Growing the tables is rare: drain the device before the
            # old blocks go back to the allocator, so no late copy can land in a recycled block.
            torch.cuda.synchronize()
            if hb["N"] == N and hb["rows_h"].shape[0] >= cap:
                keep_pinned = (hb["rows_h"], hb["stage_h"], hb["spec_rows_h"], hb["spec_stage_h"])
            hb.clear()  # the old tables go back to the allocator BEFORE the new ones are requested
        gaussians._host_bufs = None
        if keep_pinned is None:
            cap_h = bucket_size(max(int(T * 1.5), cap))
            keep_pinned = (pinned_empty((cap_h,), dtype=torch.int32), pinned_empty((cap_h, 48)), None, None)
        hb = gaussians._host_bufs = dict(
            cap=cap, N=N, cur=0, gen=gen,
            rows_h=keep_pinned[0], stage_h=keep_pinned[1], spec_rows_h=keep_pinned[2], spec_stage_h=keep_pinned[3],
            sh_stage=[torch.empty((cap, 48), device=dev), None],
            g_stage=torch.empty((cap, 48), device=dev))
    return hb


def _host_spec_buffers(hb, dev):
    """The second staging table + its pinned twin: allocated when the first hint arrives (a caller that never
    hints pays nothing for the speculative prefetch)."""
    if hb["sh_stage"][1] is None:
        hb["sh_stage"][1] = torch.empty((hb["cap"], 48), device=dev)
    if hb["spec_rows_h"] is None:
        cap_h = hb["rows_h"].shape[0]
        hb["spec_rows_h"] = pinned_empty((cap_h,), dtype=torch.int32)
        hb["spec_stage_h"] = pinned_empty((cap_h, 48))


def hint_next_batch(gaussians, cameras):
    """Host-resident mode: tell the engine which cameras the NEXT call of clm_offload_train_one_batch will get
    (a data loader knows; `trainer.training` and `bench.py` call this).  The engine then selects that batch's rows
    early (on the positions current NOW), and while the present batch still renders, the otherwise idle host pool
    brings the rows the present batch does not touch up to date and ships them -- speculation that is verified
    against the exact selection when the batch arrives (late rows are staged then, unused ones un-stamped).
    Without a hint, or with a wrong one, the engine behaves as before."""
    gaussians._next_batch_hint = list(cameras) if cameras else None


def _drop_speculation(gaussians):
    gaussians.drop_host_speculation()


def _train_one_batch_host(gaussians, scene, batched_cameras, parameters_grad_buffer, background,
                          pipe_args, comm_stream, perm_generator, args):
    """Host-resident SH rows + Adam state (sh_residency="host"), in one of two staging forms, each a module of four stage
    functions (plan -> feeders -> cameras -> the small attributes' optimizer step): per-camera staging windows
    (host_window.py, the default; optionally with an HBM budget that keeps a prefix of the rows resident) or the union of the
    batch's rows (host_batch.py, rounds 2-5's form)."""
    if getattr(args, "host_staging", "window") == "window":
        from .host_window import train_one_batch_host_windowed as run
    else:
        from .host_batch import train_one_batch_host_batched as run
    return run(gaussians, scene, batched_cameras, parameters_grad_buffer, background, pipe_args, comm_stream,
               perm_generator, args)


def clm_offload_train_one_batch(gaussians, scene, batched_cameras, parameters_grad_buffer,
                                background, pipe_args, comm_stream, perm_generator):
    """-> (losses: list[Tensor0d], ordered_cams: list[int], sparsity: list[float])
    (train.py:362-371)."""
    args = utils.get_args()
    bsz = len(batched_cameras)
    assert bsz > 1 and bsz in _BITMAP_DTYPE, "clm_offload supports bsz in (4, 8, 16, 32, 64)"
    # the deferred SH-row step applies a batch's gradient later with the scale 1 / (args.bsz x ranks), the
    # small-attribute Adam of the same batch uses len(batched_cameras): they must be the same number
    assert bsz == args.bsz, f"batch of {bsz} cameras but args.bsz = {args.bsz} (optimizer hyper-parameters are scaled by args.bsz)"
    if gaussians._parameters.is_cuda:
        with _lib.host_region("batch_total"):
            return _train_one_batch_hbm(gaussians, scene, batched_cameras, parameters_grad_buffer,
                                        background, pipe_args, comm_stream, args)
    return _train_one_batch_host(gaussians, scene, batched_cameras, parameters_grad_buffer,
                                 background, pipe_args, comm_stream, perm_generator, args)


def clm_offload_eval_one_cam(camera, gaussians, background, scene):
    """Single-camera render of the visible rows (engine.py:928-979) -> image[3,H,W]."""
    with torch.no_grad():
        a_ = utils.get_args()
        dp_partial = bool(getattr(gaussians, "lazy_rows", False) and dp.active()
                          and (getattr(a_, "dp_owner_computes", False) or getattr(a_, "dp_locality", False)))
        if dp_partial:  # before the filter: with small_owner the positions of foreign rows are stale until then
            gaussians.flush_lazy_rows()  # collective (no-op unless a batch ran since the last flush)
        if getattr(gaussians, "small_deferred", False):
            gaussians.flush_small()  # single GPU: the small attributes' waiting steps, before anything reads them
        filters, _, _ = calculate_filters([camera], gaussians.get_xyz, gaussians.get_opacity,
                                          gaussians.get_scaling, gaussians.get_rotation)
        f = filters[0]
        if getattr(gaussians, "lazy_rows", False):
            if not getattr(gaussians, "moments_sharded", False):  # (sharded moments: the flush left every row current)
                gaussians.catch_up_rows(f.to(torch.int32))
        if getattr(gaussians, "deferred_host_rows", False):  # host rows: apply what is waiting for them
            gaussians.host_rows_prepare(f.to(torch.int32).cpu().contiguous(), None)
        xyz = gaussians._xyz.detach()[f]
        opa = gaussians.opacity_activation(gaussians._opacity.detach()[f])
        sca = gaussians.scaling_activation(gaussians._scaling.detach()[f])
        rot = gaussians.rotation_activation(gaussians._rotation.detach()[f])
        shs = torch.empty((f.shape[0], 48), device=xyz.device)
        send_shs2gpu_stream(shs, gaussians._parameters.data, f)
        image, _, _ = pipeline_forward_one_step(opa, sca, rot, xyz, shs, camera, scene, gaussians,
                                                background, None, eval=True)
    return image
