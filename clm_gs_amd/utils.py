"""Process-wide run state and small host helpers the engines read.

Mirrors the accessor names of the reference's utils/general_utils.py:23-100
(get_args, get_img_width/height, get_cur_iter, get_log_file, get_timers) and
restates check_update_at_this_iter (:130-142), inverse_sigmoid (:145-146) and
get_expon_lr_func (:259-292) so engine code reads like the reference's.
"""
import io
import time
from types import SimpleNamespace

import numpy as np
import torch

ARGS = None
LOG_FILE = None
CUR_ITER = 0
IMG_W = 0
IMG_H = 0
TIMERS = None
DENSIFY_ITER = 0


def default_args(**over):
    """Flag names and defaults of arguments/__init__.py (the subset the hot path reads)."""
    a = SimpleNamespace(
        # AuxiliaryParams (arguments/__init__.py:60-150)
        no_offload=False, naive_offload=False, clm_offload=False, prealloc_capacity=-1,
        comm_stream_priority=-1, grid_size_H=32, grid_size_D=128,
        reorder_by_min_sparsity_at_end=True, sparse_adam=False, gpu=0, packed=False,
        # ModelParams
        sh_degree=3, radius_clip=0.0, white_background=False,
        # OptimizationParams (:194-235)
        iterations=30_000, position_lr_init=0.00016, position_lr_final=0.0000016,
        position_lr_delay_mult=0.01, position_lr_max_steps=30_000, feature_lr=0.0025,
        opacity_lr=0.05, scaling_lr=0.005, lr_scale_loss=1.0, lr_scale_pos_and_scale=1.0,
        rotation_lr=0.001, percent_dense=0.01, lambda_dssim=0.2, densification_interval=100,
        opacity_reset_interval=3000, densify_from_iter=500, densify_until_iter=15_000,
        densify_grad_threshold=0.0002, disable_auto_densification=False, min_opacity=0.005,
        lr_scale_mode="sqrt", bsz=1, exact_filter=True, log_cpu_adam_trailing_overhead=False,
        # Debug
        stop_update_param=False, drop_initial_3dgs_p=0.0,
        debug_skip_optimizer=False,  # this build, tests only: run the batch, leave the gradients unconsumed
        # this build: where the SH rows + their optimizer state live
        sh_residency="hbm",
        # this build: fused front-end kernels + no autograd tape inside the engines (fused.py);
        # False = the op-by-op gsplat/clm_kernels chain the reference engines spell out
        fused_front_end=True,
        overlap_cameras=True,  # False: one camera after the other on one stream (A/B, kernel timing); True: the pipeline
        packed_small=True, packed_stats=True, exact_tile_cull=True,
        sync_each_batch=False,  # True: torch.cuda.synchronize() at the end of every batch
        dp_overlap=True,        # camera-DP locality exchange: split B / D so that they hide behind the first / last camera
        dp_shard_moments=True,  # camera-DP locality exchange (dense deferred row optimizer): m / v of the SH row table only for the owned row range
        dp_small_owner=True,    # ... and xyz / opacity / scaling / rotation stepped by the owner of a row range only (no step F)
        dp_small_refresh=8,     # ... batches between two all-gathers of the owned small-attribute ranges (bounds the staleness)
        dp_small_max_log_gain=0.7,  # ... or earlier, once the bound on the growth of a stale scale exceeds exp(this)
        binning="tile",  # per-camera tile binning: "tile" = per-tile counters + scatter + one LDS sort per tile (isect3.hip,
                         # 7 launches); "sort" = depth sort of the rows + stable sort on the tile id (isect.hip, 23 launches)
        deferred_small_adam=True,  # single GPU, dense optimizer: xyz / opacity / scaling / rotation are stepped per block of 256
                                   # Z-ordered rows when a batch's cameras may see it (GaussianModelCLMOffload.small_deferred)
        allocator_reservoir=True,  # trainer: one block per stream pool allocated and freed before the first batch
                                   # (engine.reserve_working_set): the caching allocator splits it instead of calling hipMalloc
        split_catch_up=True,    # camera pipeline: the deferred SH-row steps run camera by camera on a side stream; the first
                                # camera starts after its own rows' pass instead of the whole batch's
        early_first_catch_up=True,  # ... and the first camera's call is issued before the host work of the camera stage
        warm_structural_ops=True,  # trainer: the torch operators of a densification run once on small tensors before the
                                   # end-to-end clock (lazily loaded device code: 110-130 ms inside the first densification)
        defer_loss_log=True,    # trainer: a batch's loss line is written once the NEXT batch is enqueued (no device drain)
        spatial_row_order=True,   # trainer: keep the rows in Z-order of (x, y) (after loading / densification)
        first_touch_grads=True,  # fused HBM engine: SH gradient rows stored on first touch, never cleared
        host_staging="window",           # host-resident mode: "window" = per-camera staging tables + the rows several cameras
                                         # share (strategies/clm_offload/host_window.py), "batch" = the union of the batch's rows
        placement_candidates=1,          # HBM row tables of >= 1 GB: keep the best-placed of this many allocations, judged by a
                                         # gather probe at set-up time (clm_gs_amd/placement.py); 1 = take what comes
        sh_hbm_budget_gb=0.0,            # host-resident mode, window staging: this much HBM (768 B per row: parameters, two
                                         # moments, gradient) keeps rows [0, K) of the Z-ordered SH table RESIDENT -- rendered
                                         # from and stepped in HBM, never crossing the link (gaussian_model.hbm_prefix_*)
        host_speculative_prefetch=True,  # host-resident mode: stage the hinted next batch's untouched rows early
        device_side_counts=True,   # fused engine: consumers of a camera's intersection list read its length on the device
        isect_capacity_margin=1.25,  # ... from buffers sized (largest count seen at this image size) x margin
        isect_capacity_floor=4096,   # ... + this many entries
        reference_camera_order=False,  # clm_offload: process (and report) a batch's cameras in the reference's TSP order
        dp_locality=False,  # camera-DP: owner-computes with point-to-point traffic only (dp.py "locality exchange"):
        # a rank fetches just the rows its cameras touch outside its own index range and returns their gradients
        dp_owner_computes=False,  # camera-DP: rows are owned by index range; all-gather of parameter rows before
                                  # rendering, reduce-scatter of gradient rows after it, only the owner steps a row
        lazy_dense_adam=True,   # HBM rows: replay zero-gradient Adam steps on demand (exact)   # two cameras of a batch in flight on two HIP streams
    )
    for k, v in over.items():
        setattr(a, k, v)
    return a


class _NullLog(io.StringIO):
    def write(self, s):
        return len(s)


class Timers:
    def __init__(self):
        self.t = {}
        self.acc = {}

    def start(self, k):
        self.t[k] = time.perf_counter()

    def stop(self, k):
        if k in self.t:
            self.acc[k] = self.acc.get(k, 0.0) + time.perf_counter() - self.t.pop(k)


def set_args(a):
    global ARGS
    ARGS = a


def get_args():
    global ARGS
    if ARGS is None:
        ARGS = default_args()
    return ARGS


def set_log_file(f):
    global LOG_FILE
    LOG_FILE = f


def get_log_file():
    global LOG_FILE
    if LOG_FILE is None:
        LOG_FILE = _NullLog()
    return LOG_FILE


def set_cur_iter(i):
    global CUR_ITER
    CUR_ITER = i


def get_cur_iter():
    return CUR_ITER


def set_img_size(h, w):
    global IMG_H, IMG_W
    IMG_H, IMG_W = int(h), int(w)


def get_img_width():
    return IMG_W


def get_img_height():
    return IMG_H


def get_timers():
    global TIMERS
    if TIMERS is None:
        TIMERS = Timers()
    return TIMERS


def inc_densify_iter():
    global DENSIFY_ITER
    DENSIFY_ITER += 1


def check_update_at_this_iter(iteration, bsz, update_interval, update_residual):
    """True when some image index in [iteration, iteration+bsz) is congruent to
    update_residual mod update_interval (general_utils.py:130-142)."""
    lo = iteration % update_interval
    hi = lo + bsz
    return (lo <= update_residual < hi) or (lo <= update_residual + update_interval < hi)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear decay with optional warm-up (general_utils.py:259-292)."""

    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay = 1.0
        t = np.clip(step / max_steps, 0, 1)
        return delay * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)

    return helper


def build_rotation(r):
    """(w,x,y,z) -> R [n,3,3] (general_utils.py:311-334)."""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


SH_C0 = 0.28209479177387814


def RGB2SH(rgb):
    return (rgb - 0.5) / SH_C0


def SH2RGB(sh):
    return sh * SH_C0 + 0.5


def morton_order(xyz, bits=16):
    """Permutation that sorts points along a Z-order (Morton) curve of their (x, y) coordinates
    (`bits` bits per axis; z is ignored: the scenes this engine trains are aerial / terrestrial slabs and
    a camera's footprint is an area of the ground plane).  Rows stored in this order make the rows a
    camera sees -- and the rows a batch touches -- contiguous runs of the row tables instead of isolated
    192 B rows scattered over gigabytes: coalesced gathers, TLB reach, streaming host walks.
    Row order is not part of the model: any permutation of the Gaussians renders the same image (up to
    the tie order of equal depths).

    Points on the GPU (bits = 16): the library's one-pass form, clmgs_morton_order -- the same IEEE double
    arithmetic and a stable radix sort, hence the SAME permutation as the torch form below (kept as
    `morton_order_torch`: host tensors, other bit counts, and the test's reference); ~1 ms at 28 M rows instead of
    ~10 ms of elementwise passes and a 64-bit sort, inside every densification of a training run."""
    xyz = xyz.detach()
    n = xyz.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=xyz.device)
    if xyz.is_cuda and bits == 16 and 0 < n < 2 ** 31 and xyz.dtype == torch.float32:
        from . import _lib
        with torch.no_grad():
            xy = xyz[:, :2]
            lo_hi = torch.cat((xy.amin(dim=0), xy.amax(dim=0))).double()  # (exact: float -> double is monotonic)
            p = xyz.contiguous()
            L = _lib.lib()
            nbytes = int(L.clmgs_morton_order_temp_bytes(n))
            temp = torch.empty((nbytes,), dtype=torch.uint8, device=xyz.device)
            order = torch.empty((n,), dtype=torch.int64, device=xyz.device)
            _lib.check(L.clmgs_morton_order(_lib.stream(), n, _lib.dptr(p), lo_hi.data_ptr(), order.data_ptr(),
                                            temp.data_ptr(), nbytes))
            return order
    return morton_order_torch(xyz, bits)


def morton_order_torch(xyz, bits=16):
    """morton_order in plain torch ops (any device)."""
    with torch.no_grad():
        p = xyz.detach()[:, :2].double()
        lo, hi = p.min(dim=0).values, p.max(dim=0).values
        q = ((p - lo) / (hi - lo).clamp_min(1e-30) * (2 ** bits - 1)).round().to(torch.int64)

        def spread(v):  # 16 bits -> every second bit of 32
            v = (v | (v << 8)) & 0x00FF00FF
            v = (v | (v << 4)) & 0x0F0F0F0F
            v = (v | (v << 2)) & 0x33333333
            v = (v | (v << 1)) & 0x55555555
            return v
        code = spread(q[:, 0]) | (spread(q[:, 1]) << 1)
        return torch.sort(code, stable=True).indices


def select_rows(t, mask, chunk=1 << 25):
    """t[mask] (bool row mask) for row tables of any size: tables of more than 2^25 rows are selected in
    chunks of source rows (see gather_rows: advanced indexing of [N,4] tables misbehaved beyond ~35 M rows)."""
    n = t.shape[0]
    if n <= chunk:
        return t[mask]
    return torch.cat([t[a:a + chunk][mask[a:a + chunk]] for a in range(0, n, chunk)], dim=0)


def take_rows(t, idx, chunk=1 << 23):
    """t[idx] (int64 row ids, any count) in chunks of `chunk` indices; see gather_rows."""
    n = idx.numel()
    if n <= chunk:
        return t[idx]
    out = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    for a in range(0, n, chunk):
        out[a:a + chunk] = t[idx[a:a + chunk]]
    return out


def put_rows(t, idx, src, chunk=1 << 23):
    """t[idx] = src in chunks of `chunk` indices (unique row ids)."""
    n = idx.numel()
    for a in range(0, n, chunk):
        t[idx[a:a + chunk]] = src[a:a + chunk]


def fill_rows(t, idx, value, chunk=1 << 23):
    """t[idx] = value (a scalar) in chunks of `chunk` indices; index_fill_ takes the scalar as a kernel
    argument (no host -> device copy of a 0-dim tensor, which would block the enqueueing thread)."""
    n = idx.numel()
    for a in range(0, n, chunk):
        t.index_fill_(0, idx[a:a + chunk], value)


def gather_rows(t, order, chunk=1 << 23):
    """t[order] for row tables of any size, gathered in chunks of rows.

    Why chunks (profiles/repro_index_defect.py, profiles/r03_index_defect.json): on torch 2.10 / ROCm 7,
    `t[idx]` and `torch.index_select(t, 0, idx)` with MORE THAN 2^26 INDICES into a table whose rows are a
    multiple of 16 bytes ([N,4] fp32; [N,3] is fine) write only the first (len(idx) mod 2^26) output rows --
    102 231 360 indices: rows 35 122 496.. are garbage.  The fast path launches one 64-thread workgroup per
    index, and HIP caps gridDim.x * blockDim.x at 2^32, i.e. 2^26 such workgroups.  Every row selection /
    permutation / exchange of the package stays below that by construction (2^23 indices per call);
    tests/test_gpu_row_indexing.py pins the helpers at the failing size."""
    out = torch.empty_like(t)
    n = order.numel()
    for a in range(0, n, chunk):
        out[a:a + chunk] = t[order[a:a + chunk]]
    return out
