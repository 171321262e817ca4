"""ctypes binding of libclmgs_hip.so (C ABI declared in include/clmgs.h).

The library is the product: if it is missing, or if a tensor handed to an
operator is not a contiguous CUDA(HIP) tensor of the declared dtype, we raise --
there is deliberately no eager/PyTorch fallback path.
"""
import ctypes
import os
import time

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLMGS_LIB_PATH") or os.path.join(_HERE, "libclmgs_hip.so")  # override: A/B builds

_vp, _i, _i64, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
_d = ctypes.c_double

# name -> (restype, argtypes); mirrors include/clmgs.h one to one
SIGNATURES = {
    "clmgs_version": (_i, []),
    "clmgs_last_error": (ctypes.c_char_p, []),
    "clmgs_projection_fwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "clmgs_visibility_select_temp_bytes": (_sz, [_i, _i]),
    "clmgs_visibility_select_count": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _sz, _vp]),
    "clmgs_visibility_select_count_blocks": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _sz, _vp, _vp]),
    "clmgs_visibility_select_emit": (_i, [_vp, _i, _i, _vp, _vp]),
    "clmgs_visibility_raw": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp]),
    "clmgs_projection_bwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clmgs_sh_fwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "clmgs_sh_bwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "clmgs_isect_count_temp_bytes": (_sz, [_i]),
    "clmgs_isect_count": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz]),
    "clmgs_isect_sort_temp_bytes": (_sz, [_i64]),
    "clmgs_isect_emit_sort": (_i, [_vp, _i, _i, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz]),
    "clmgs_isect_offsets": (_i, [_vp, _i64, _vp, _i, _i, _i, _vp]),
    "clmgs_isect2_order_temp_bytes": (_sz, [_i]),
    "clmgs_isect2_order_count": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "clmgs_isect2_sort_temp_bytes": (_sz, [_i64]),
    "clmgs_isect2_emit_sort": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "clmgs_isect2_emit_sort_dev": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "clmgs_isect3_front_temp_bytes": (_sz, [_i, _i]),
    "clmgs_isect3_front": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz]),
    "clmgs_isect3_bin_temp_bytes": (_sz, [_i64, _i]),
    "clmgs_isect3_bin": (_i, [_vp, _i, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "clmgs_isect3_bin_dev": (_i, [_vp, _i, _i64, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "clmgs_rasterize_fwd_dev": (_i, [_vp, _i, _i, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clmgs_rasterize_bwd_dev": (_i, [_vp, _i, _i, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clmgs_rasterize_pack_bytes": (_sz, [_i, _i]),
    "clmgs_rasterize_fwd": (_i, [_vp, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clmgs_rasterize_partials_bytes": (_sz, [_i64]),
    "clmgs_rasterize_bwd": (_i, [_vp, _i, _i, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clmgs_preprocess_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clmgs_preprocess_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i]),
    "clmgs_ssim_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clmgs_ssim_bwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "clmgs_loss_slots": (_i, []),
    "clmgs_l1_ssim_loss_fwd": (_i, [_vp, _i, _i, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "clmgs_l1_ssim_loss_bwd": (_i, [_vp, _i, _i, _vp, _i64, _i64, _i64, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "clmgs_rows_gather": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i]),
    "clmgs_rows_scatter_add": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i]),
    "clmgs_scatter_to_bit": (_i, [_vp, _vp, _i, _vp, _i64, _i]),
    "clmgs_extract_ffs": (_i, [_vp, _vp, _i, _i64, _vp]),
    "clmgs_pair_overlap_count": (_i, [_vp, _vp, _i, _i64, _i, _vp]),
    "clmgs_set_signal": (_i, [_vp, _vp, _i, ctypes.c_int32]),
    "clmgs_adam_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _vp, _d, _d, _d, _i, _i, _f, _i]),
    "clmgs_pack_small": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "clmgs_adam_small_packed": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _i, _i, _f, _vp, _i]),
    "clmgs_adam_catch_up": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _d, _d, _d, _i, _i, _i, _vp, _vp, _f, _i]),
    "clmgs_host_adam_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _d, _d, _d, _i, _i, _f, _i, _vp, _i]),
    "clmgs_host_pool_start": (_i, [_i]),
    "clmgs_host_usable_cpus": (_i, []),
    "clmgs_memcpy_async": (_i, [_vp, _vp, _vp, _sz, _i]),
    "clmgs_host_rows_prepare": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _d, _d, _d, _i, _i, _i, _f, _i, _vp, _i]),
    "clmgs_densify_stats": (_i, [_vp, _i64, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp]),
    "clmgs_morton_order_temp_bytes": (_sz, [_i64]),
    "clmgs_morton_order": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _sz]),
    "clmgs_tsp_tour": (_i, [_i, _vp, _vp]),
    "clmgs_knn3_mean_dist2": (_i, [_vp, _i, _vp, _vp, _f, _f, _f, _f, _i, _i, _i, _i, _vp]),
    "clmgs_host_groups_temp_bytes": (_sz, [_i64]),
    "clmgs_host_groups": (_i, [_vp, _i64, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz]),
    "clmgs_publish_pack": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i64, _i64, _i]),
    "clmgs_visibility_candidates": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _f, _vp]),
    "clmgs_small_rows_scatter": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clmgs_adam_small_packed_range": (_i, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _i, _i, _f, _vp, _i]),
    "clmgs_small_deferred_kmax": (_i, []),
    "clmgs_adam_small_deferred": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _d, _d, _d,
                                       _f, _i, _vp, _vp, _i, _i, _f, _f, _f, _i, _vp]),
    "clmgs_debug_counters": (_i, [_vp, _i]),
    "clmgs_device_errors": (_i, [_vp, _i]),
    "clmgs_pinned_alloc": (_vp, [_sz]),
    "clmgs_pinned_free": (_i, [_vp]),
}

_lib = None

# Optional per-entry-point device timing (bench.py's roofline leg): when TIMING is a dict every
# stream-taking call is bracketed by two events recorded on the stream it is launched on.
TIMING = None
# Data-dependent sizes seen by the front end (intersections per image), for the same purpose.
STATS = {"n_isects": [], "n_emitted": [], "host_wait_s": 0.0}  # host_wait_s: time blocked in size readbacks


class ClmgsError(RuntimeError):
    pass


class _Entry:
    """One C entry point; times itself on the launch stream when TIMING is enabled."""

    def __init__(self, name, fn, takes_stream):
        self.name, self.fn, self.takes_stream = name, fn, takes_stream

    def __call__(self, *args):
        if TIMING is None or not self.takes_stream:
            return self.fn(*args)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h = getattr(args[0], "value", args[0])  # events go on the stream the kernel is launched on
        st = torch.cuda.ExternalStream(h) if h else torch.cuda.current_stream()
        s.record(st)
        rc = self.fn(*args)
        e.record(st)
        TIMING.setdefault(self.name, []).append((s, e))
        return rc


class _Namespace:
    pass


_NO_STREAM = {"clmgs_version", "clmgs_loss_slots", "clmgs_small_deferred_kmax", "clmgs_isect3_front_temp_bytes",
              "clmgs_isect3_bin_temp_bytes", "clmgs_last_error", "clmgs_isect_count_temp_bytes",
              "clmgs_isect_sort_temp_bytes", "clmgs_host_groups_temp_bytes", "clmgs_isect2_order_temp_bytes", "clmgs_isect2_sort_temp_bytes", "clmgs_visibility_select_temp_bytes", "clmgs_rasterize_pack_bytes", "clmgs_rasterize_partials_bytes", "clmgs_host_adam_rows", "clmgs_host_pool_start", "clmgs_host_usable_cpus", "clmgs_host_rows_prepare", "clmgs_tsp_tour",
              "clmgs_pinned_alloc", "clmgs_pinned_free", "clmgs_debug_counters", "clmgs_device_errors"}


def timing_summary():
    """name -> (n_calls, total_ms); call after torch.cuda.synchronize()."""
    out = {}
    for name, evs in (TIMING or {}).items():
        out[name] = (len(evs), sum(s.elapsed_time(e) for s, e in evs))
    return out


def lib():
    """Load (once) and return the shared library; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ClmgsError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C clm_gs_amd/csrc).  clm_gs_amd has no CPU/eager fallback."
            )
        l = ctypes.CDLL(LIB_PATH)
        ns = _Namespace()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError = ABI drift, also loud
            fn.restype = res
            fn.argtypes = args
            setattr(ns, name, _Entry(name, fn, name not in _NO_STREAM))
        _lib = ns
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().clmgs_last_error()
        raise ClmgsError(f"libclmgs_hip error {rc}: {msg.decode() if msg else ''}")


def check_device_errors():
    """Raises if a look-back scan / sort pass of the binning chain timed out since the last check
    (clmgs_device_errors; synchronises the device).  Called where the host synchronises anyway."""
    bits = ctypes.c_uint32(0)
    check(lib().clmgs_device_errors(ctypes.byref(bits), 1))
    if bits.value & 4:
        raise ClmgsError("deferred small-attribute Adam: a block of rows was further behind than the recorded step history "
                         "(clmgs_adam_small_deferred, device error bit 4): its parameters are wrong")
    if bits.value:
        raise ClmgsError(f"binning chain: look-back timed out on the device (bits {bits.value}: 1 = scan, 2 = sort pass); "
                         "the intersection lists built since the last check are invalid")


HOST_REGIONS = None  # set to {} to accumulate host wall time per engine region (diagnostics)
REGION_TRACE = None  # set to {} (with HOST_REGIONS) to keep every occurrence's time as well
_ROCTX = None        # libroctx64 handle when CLMGS_ROCTX=1 (ranges show up in `rocprofv3 --marker-trace`)


def roctx():
    """roctx range markers around the engine's regions (the reference's torch.cuda.nvtx.range_push / pop,
    train.py:238-250): off unless CLMGS_ROCTX=1, so the product path makes no extra calls."""
    global _ROCTX
    if _ROCTX is None:
        _ROCTX = False
        if os.environ.get("CLMGS_ROCTX") == "1":
            for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    l = ctypes.CDLL(name)
                    l.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    l.roctxRangePushA.restype = ctypes.c_int
                    l.roctxRangePop.restype = ctypes.c_int
                    _ROCTX = l
                    break
                except (OSError, AttributeError):
                    continue
    return _ROCTX


class host_region:
    """with host_region("name"): ...   -- adds the block's host wall time to HOST_REGIONS[name]; with
    CLMGS_ROCTX=1 the block is also a roctx range."""
    __slots__ = ("name", "t0", "rx")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.rx = roctx()
        if self.rx:
            self.rx.roctxRangePushA(self.name.encode())
        if HOST_REGIONS is not None:
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if HOST_REGIONS is not None:
            dt = time.perf_counter() - self.t0
            HOST_REGIONS[self.name] = HOST_REGIONS.get(self.name, 0.0) + dt
            if REGION_TRACE is not None:  # per-occurrence times (diagnostics: which batch paid)
                REGION_TRACE.setdefault(self.name, []).append(round(dt * 1e3, 2))
        if self.rx:
            self.rx.roctxRangePop()
        return False


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t, dtype=None, allow_none=False, allow_host=False):
    """Raw pointer of a contiguous device tensor (or pinned host tensor if allowed)."""
    if t is None:
        if allow_none:
            return None
        raise ClmgsError("required tensor is None")
    if dtype is not None and t.dtype != dtype:
        raise ClmgsError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ClmgsError("tensor must be contiguous")
    if not t.is_cuda:
        if not (allow_host and (t.is_pinned() or getattr(t, "_clmgs_pinned", False))):
            raise ClmgsError("tensor must live on the GPU (no CPU fallback in clm_gs_amd)")
    return ctypes.c_void_p(t.data_ptr())
