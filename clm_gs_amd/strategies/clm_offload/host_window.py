"""Host-resident SH rows with PER-CAMERA staging windows (sh_residency="host", host_staging="window": the default).

The reference's offloading pipeline keeps, per micro-batch, only what that camera needs on the GPU and retains the rows the
next camera shares with it (retention sets H / D / G, strategies/clm_offload/engine.py:566-636; 13.0 GB of GPU memory at
28 M Gaussians, release_scripts/rubble4k_README.md:113).  `host_batch.py` (host_staging="batch") stages the
UNION of a batch's rows instead -- three [T, 48] tables, 6.6 GB at 28 M.  This module keeps that mode's properties

  * every touched row crosses the host link ONCE per direction and batch (parameters in before the first camera that
    uses it, gradient home after the last one; plain stores, no read-modify-write over the link),
  * the deferred host row optimizer + staging copy on the host pool, chunked hipMemcpyAsync on a side stream,
  * speculative prefetch of the hinted next batch while the present one renders (here: of its FIRST camera's rows),

with tables sized for ONE camera plus the rows that several cameras of the batch share:

    params   PT = [ P: rows used by >= 2 cameras, kept from their first to their last use | A: the current camera's arrivals ]
    grads    GT = [ GP                                                                    | GA ]       (same slot numbers)
    bounce   BB = [ the NEXT camera's arrivals land here (hipMemcpyAsync) while the current camera renders ]

Per camera k:  BB -> A (one device copy; BB is free for camera k+1's rows)  ->  render + backward from / into PT / GT by slot
->  the rows whose LAST camera this was send their gradient row home (zero-copy stores on a side stream)  ->  the arrivals
that a later camera uses again ("survivors") move, parameters and partial gradient, from A to their P slot.
Slots are planned once per batch from the visibility bitmap (a row is a survivor iff more than one bit is set); nothing is
recycled inside a batch, so the plan is a handful of scans over the batch's row list and two small read-backs.

28 M Gaussians, 4K, bsz 4: 3.4 GB of staging instead of 6.6 GB (peak 14.9 GB instead of 18.1 GB).
"""
import ctypes
import threading
import time

import torch

from ... import _lib, clm_kernels, utils
from ...gsplat import bucket_size
from ...host import pinned_empty
from ..base_engine import select_filters

_CHUNK_ROWS = 262144  # staging granularity: 48 MB of parameter rows per hipMemcpyAsync


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class _Win:
    """State of one windowed host-resident batch, handed from stage to stage."""


def _buffers(gaussians, need_a, need_p, need_late, dev, K=0):
    """Device tables (bucketed capacity, 8 % head room) + pinned host twins (50 %: host memory is cheap, and a pinned
    re-allocation is a hipHostMalloc of hundreds of ms).  `gen` counts device re-allocations: rows staged into an older
    generation are gone.  K > 0 (sh_hbm_budget_gb): the first K rows of the parameter and of the gradient table are the
    HBM-resident prefix of the model (slot r = row r), the P and A regions follow; a re-allocation carries them over."""
    hb = getattr(gaussians, "_hwin_bufs", None)
    N = gaussians._xyz.shape[0]
    if hb is None or hb["N"] != N or hb.get("K", 0) != K or hb["cap_a"] < need_a or hb["cap_p"] < need_p:
        gen = hb["gen"] + 1 if hb else 0
        keep, old_pt = None, None
        if hb is not None:
            # the tables are written by hipMemcpyAsync issued through ctypes on side streams (the caching allocator does not
            # know): drain the device before the old blocks go back to it
            torch.cuda.synchronize()
            same = hb["N"] == N and hb.get("K", 0) == K
            if same:
                keep = hb
                px = getattr(gaussians, "_hbm_prefix", None)
                if K and px is not None and not px["fill"]:
                    # the resident rows move to the new table; their waiting gradients are applied first (one pass of the
                    # deferred row optimizer over the K rows, ~4 ms at 14 M), so the gradient table holds zeros there and
                    # is not carried over: the only transient is the old parameter table beside the new one
                    if not gaussians.args.sparse_adam:
                        gaussians.hbm_prefix_catch_up(None, int(gaussians.optimizer.cpu_adam.global_step))
                    old_pt = hb["pt"]
            cap_a = max(hb["cap_a"] if same else 0, bucket_size(max(int(need_a * 1.08), 1)))
            cap_p = max(hb["cap_p"] if same else 0, bucket_size(max(int(need_p * 1.08), 1)))
            for k in ("pt", "gt", "bb"):
                hb[k] = None
        else:
            cap_a, cap_p = bucket_size(max(int(need_a * 1.08), 1)), bucket_size(max(int(need_p * 1.08), 1))
        gaussians._hwin_bufs = None
        new = dict(N=N, K=K, gen=gen, cap_a=cap_a, cap_p=cap_p, bb=torch.empty((cap_a, 48), device=dev),
                   rows_h=keep["rows_h"] if keep else None, stage_h=keep["stage_h"] if keep else None,
                   spec_rows_h=keep["spec_rows_h"] if keep else None, spec_stage_h=keep["spec_stage_h"] if keep else None)
        new["pt"] = torch.empty((K + cap_p + cap_a, 48), device=dev)
        if old_pt is not None:
            new["pt"][:K].copy_(old_pt[:K])
            old_pt = None
        new["gt"] = torch.empty((K + cap_p + cap_a, 48), device=dev)
        if K:
            new["gt"][:K].zero_()
        hb = gaussians._hwin_bufs = new
    if hb["rows_h"] is None or hb["rows_h"].shape[0] < need_late:
        cap_h = bucket_size(max(int(need_late * 1.5), 1))
        hb["rows_h"], hb["stage_h"] = pinned_empty((cap_h,), dtype=torch.int32), pinned_empty((cap_h, 48))
    if hb["spec_rows_h"] is None or hb["spec_rows_h"].shape[0] < hb["cap_a"]:
        hb["spec_rows_h"], hb["spec_stage_h"] = pinned_empty((hb["cap_a"],), dtype=torch.int32), pinned_empty((hb["cap_a"], 48))
    return hb


def _tables(gaussians, dev, K=0):
    """Row-indexed scratch (independent of the batch): the slot every touched row lives in right now, two row masks.
    Rows of the HBM-resident prefix [0, K) live in slot r for good."""
    N = gaussians._xyz.shape[0]
    ht = getattr(gaussians, "_hwin_tabs", None)
    if ht is None or ht["N"] != N or ht.get("K", 0) != K:
        ht = gaussians._hwin_tabs = dict(N=N, K=K, cur_slot=torch.zeros((N,), dtype=torch.int32, device=dev),
                                         scratch_slot=torch.zeros((N,), dtype=torch.int32, device=dev),
                                         mark=torch.zeros((N,), dtype=torch.bool, device=dev),
                                         in_spec=torch.zeros((N,), dtype=torch.bool, device=dev),
                                         cam0=torch.zeros((N,), dtype=torch.bool, device=dev))
        if K:
            ht["cur_slot"][:K] = torch.arange(K, dtype=torch.int32, device=dev)
    return ht


def drop_speculation(gaussians):
    """Forget the rows staged for a hinted batch that is not coming (their host stamps expect a gradient that will not land)."""
    sp = getattr(gaussians, "_hwin_spec", None)
    gaussians._hwin_spec = None
    if sp is None:
        return
    sp["thread"].join()
    if sp.get("event") is not None:
        sp["event"].synchronize()
    if sp["n"]:
        gaussians._host_g_step[sp["rows_h"][:sp["n"]].long()] = 0


# ------------------------------------------------------------------------------------------------------------ stage 1
def _plan(w):
    """Visibility filters, the verdict on what was staged speculatively, the rows grouped by first / last camera, and the
    slot plan: arrival slots (A region) of every touched row, the survivors (rows used by more than one camera) with their
    P slots, the slot every row leaves from."""
    from .engine import _encode_bitmap, order_calculation
    g, args, dev, bsz, N, L = w.gaussians, w.args, w.dev, w.bsz, w.N, _lib.lib()
    with _lib.host_region("select_filters"):
        w.filters, touched = select_filters(w.cameras, g._xyz.detach(), g._scaling.detach(), g._rotation.detach())
    _lib.STATS.setdefault("touched_rows", []).append(int(touched.shape[0]))
    w.touched_all = touched
    K = w.K
    w.prefix_rows = None
    if K:
        # sh_hbm_budget_gb: rows [0, K) are resident (slot r = row r, stepped in HBM); only the rest is planned, staged and
        # stepped on the host below.  (touched is ascending: one comparison pass, one host read of the split point)
        n_lo = int((touched < K).sum())
        w.prefix_rows = touched[:n_lo].to(torch.int32)
        touched = touched[n_lo:]
    T = int(touched.shape[0])
    _lib.STATS.setdefault("host_touched_rows", []).append(T)
    w.sparsity = [len(f) / float(N) for f in w.filters]
    w.ordered_cams = list(range(bsz))
    if getattr(args, "reference_camera_order", False):  # the reference's TSP order (engine.py:135-298)
        _, w.cameras, w.filters, w.sparsity, w.ordered_cams = order_calculation(
            list(w.filters), list(w.cameras), N, bsz, w.perm_generator, args)[:5]
    # ---- what was staged for this batch's first camera while the previous batch rendered
    spec = getattr(g, "_hwin_spec", None)
    gen0 = (getattr(g, "_hwin_bufs", None) or {}).get("gen")
    if spec is not None and (spec["cam0"] != id(w.cameras[0]) or spec["N"] != N or spec["gen"] != gen0):
        drop_speculation(g)
        spec = None
    g._hwin_spec = None
    if spec is not None:
        with _lib.host_region("spec_join"):
            spec["thread"].join()
        if spec["err"]:
            raise spec["err"][0]
    ht = _tables(g, dev, K)
    mark, in_spec, cam0, cur_slot = ht["mark"], ht["in_spec"], ht["cam0"], ht["cur_slot"]
    with _lib.host_region("host_groups"):
        mark.zero_()
        utils.fill_rows(mark, touched, True)
        bitmap = _encode_bitmap(w.filters, N, bsz)  # bit bsz-1-i = camera i

        def group(sp):
            """-> (n_p, staged mask | None, wasted rows, late_all, rows_by_last32, cl)"""
            n_p, staged, wasted = 0, None, touched[:0]
            if sp is not None and sp["n"]:
                n_p = sp["n"]
                P = sp["rows"]
                in_spec.zero_()
                utils.fill_rows(in_spec, P, True)
                cam0.zero_()
                utils.fill_rows(cam0, w.filters[0], True)
                staged = in_spec & cam0                      # staged AND used by the camera they were staged for
                wasted = P[~staged[P]]                       # no gradient of this batch will land on them ...
                wasted = wasted[~mark[wasted]]               # ... unless a later camera stages them again (then re-stamped)
            late_all = torch.empty((T,), dtype=torch.int32, device=dev)
            rows_by_last32 = torch.empty((T,), dtype=torch.int32, device=dev)
            counts = torch.empty((2 * bsz + 1,), dtype=torch.int64, device=dev)
            if T == 0:  # (every touched row is resident: nothing to stage)
                return n_p, staged, wasted, late_all, rows_by_last32, [0] * (2 * bsz + 1)
            tb = L.clmgs_host_groups_temp_bytes(T)
            tmp = torch.empty((tb,), dtype=torch.uint8, device=dev)
            _lib.check(L.clmgs_host_groups(_lib.stream(), T, _lib.dptr(touched, torch.int64), _lib.dptr(bitmap),
                                           bitmap.element_size(), bsz,
                                           _lib.dptr(staged.view(torch.uint8), None, True) if staged is not None else None, 0,
                                           _lib.dptr(late_all), _lib.dptr(rows_by_last32), _lib.dptr(ht["scratch_slot"]),
                                           _lib.dptr(counts), _lib.dptr(tmp), tb))
            _t0 = time.perf_counter()
            cl = counts.tolist()                              # one host read: the group sizes
            _lib.STATS["host_wait_s"] += time.perf_counter() - _t0
            return n_p, staged, wasted, late_all, rows_by_last32, cl

        n_p, staged, wasted, late_all, rows_by_last32, cl = group(spec)
        n_first, n_last, n_late = [int(x) for x in cl[:bsz]], [int(x) for x in cl[bsz:2 * bsz]], int(cl[2 * bsz])
        # multi-use rows: more than one bit of the bitmap set  <=>  first camera < last camera
        multi = (bitmap & (bitmap - 1)) != 0

        def survivors(n_p_, staged_, late_all_, n_first_):
            """Arrival order = [speculative block | late rows by first camera]; -> (rows in arrival order, survivor flag,
            survivors up to the end of each camera's arrivals (device))."""
            late = late_all_[:n_late].long()
            if n_p_:
                arr = torch.cat((spec["rows"].long(), late))
                flag = torch.cat((multi[spec["rows"]] & staged_[spec["rows"]], multi[late]))
            else:
                arr, flag = late, multi[late]
            ends, e = [], n_p_
            for k in range(bsz):
                e += n_first_[k]
                ends.append(e)
            if arr.numel() == 0:
                return arr, flag, torch.zeros((bsz,), dtype=torch.int64, device=dev)
            cs = torch.cumsum(flag, 0, dtype=torch.int64)
            ends_d = torch.tensor(ends, device=dev)               # (tiny, blocking copy: the GPU is idle here anyway)
            cum = torch.where(ends_d > 0, cs[torch.clamp(ends_d - 1, min=0)], torch.zeros_like(ends_d))
            return arr, flag, cum

        arr, flag, cum = survivors(n_p, staged, late_all, n_first)
        _t0 = time.perf_counter()
        cum_h = [int(x) for x in cum.tolist()]               # second (and last) host read of the plan
        _lib.STATS["host_wait_s"] += time.perf_counter() - _t0
        n_surv = cum_h[-1] if cum_h else 0
        need_a = max([n_p + n_first[0]] + n_first[1:])
        hb = _buffers(g, need_a, n_surv, n_late, dev, K)
        if spec is not None and spec["gen"] != hb["gen"]:
            # the tables had to grow: what was staged went with the old ones -- everything is late (re-preparing a current
            # row is the identity; the stamps of the touched rows stay right, the untouched ones are un-stamped below)
            gone = spec["rows"]
            spec = None
            n_p, staged, _w, late_all, rows_by_last32, cl = group(None)
            wasted = gone[~mark[gone]]
            n_first, n_last, n_late = [int(x) for x in cl[:bsz]], [int(x) for x in cl[bsz:2 * bsz]], int(cl[2 * bsz])
            arr, flag, cum = survivors(0, None, late_all, n_first)
            cum_h = [int(x) for x in cum.tolist()]
            n_surv = cum_h[-1] if cum_h else 0
            hb = _buffers(g, max(n_first), n_surv, n_late, dev, K)
        # ---- slots (table rows: [prefix 0..K | P | A]).  A-region slot of every arrival: base_a + position inside its
        # camera's arrival block
        base_p, base_a = K, K + hb["cap_p"]
        w.base_a = base_a
        if n_p:
            cur_slot[spec["rows"].long()] = base_a + torch.arange(n_p, dtype=torch.int32, device=dev)
        g0 = 0
        for k in range(bsz):
            if n_first[k]:
                base = base_a + (n_p if k == 0 else 0)
                cur_slot[late_all[g0:g0 + n_first[k]].long()] = base + torch.arange(n_first[k], dtype=torch.int32, device=dev)
            g0 += n_first[k]
        surv_pos = torch.nonzero(flag).flatten()              # (size n_surv, known: no extra synchronisation in effect)
        w.surv_rows = arr[surv_pos]
        w.surv_src = cur_slot[w.surv_rows].long()             # their A slots ...
        w.surv_dst = base_p + torch.arange(n_surv, dtype=torch.int64, device=dev)  # ... and their P slots, in arrival order
        w.surv_cum = cum_h
        rbl = rows_by_last32.long()
        leave = cur_slot[rbl].long()
        if n_surv:  # a multi-use row leaves from its P slot
            pslot = ht["scratch_slot"]
            pslot[w.surv_rows] = w.surv_dst.to(torch.int32)
            leave = torch.where(multi[rbl], pslot[rbl].long(), leave)
        w.rows_by_last, w.leave_slots = rbl, leave
        w.n_p, w.n_first, w.n_last, w.n_late, w.n_surv, w.spec, w.hb = n_p, n_first, n_last, n_late, n_surv, spec, hb
        w.late_all, w.mark, w.touched = late_all, mark, touched
        _lib.STATS.setdefault("host_late_rows", []).append(n_late)
        if n_late:
            _lib.check(L.clmgs_memcpy_async(_lib.stream(), _ptr(hb["rows_h"][:n_late]), _ptr(late_all[:n_late]), n_late * 4, 2))
        # ---- the NEXT batch's first camera on the positions current now; what this batch touches cannot be staged early
        w.spec_rows = None
        if w.hint is not None:
            try:
                f_next, _ = select_filters(w.hint[:1], g._xyz.detach(), g._scaling.detach(), g._rotation.detach())
                sr = f_next[0]
                if K:
                    sr = sr[sr >= K]
                sr = sr[~mark[sr]]
                n_s = int(sr.shape[0])
                if 0 < n_s <= hb["cap_a"]:
                    w.spec_rows = sr
                    _lib.check(L.clmgs_memcpy_async(_lib.stream(), _ptr(hb["spec_rows_h"][:n_s]), _ptr(sr.to(torch.int32)), n_s * 4, 2))
            except AssertionError:  # the hinted camera sees nothing yet: that batch will complain itself
                w.spec_rows = None
        _t0 = time.perf_counter()
        wasted_h = wasted.cpu() if wasted.numel() else None
        _lib.STATS["host_wait_s"] += time.perf_counter() - _t0
    if wasted_h is not None:  # staged for this batch but not touched by it: they expect no gradient after all
        g._host_g_step[wasted_h] = 0


# ------------------------------------------------------------------------------------------------------------ stage 2
def _start_feeders(w):
    """Feeder thread: per first-use group, the host pool brings the rows up to date (deferred row optimizer) and copies them
    into pinned staging, chunk by chunk; each chunk goes to the bounce table with hipMemcpyAsync on the side stream -- group
    k+1 only once camera k has taken its rows out of the bounce table.  Speculation thread: the hinted next batch's first
    camera, after this batch's own rows."""
    g, hb, bsz, L = w.gaussians, w.hb, w.bsz, _lib.lib()
    row_adam = g.optimizer.cpu_adam
    w.step = step = row_adam.global_step + 1
    prev = g._host_grads_event
    w.prev_grads = prev
    g._host_grads_event = None
    w.comm_stream.wait_stream(w.default_stream)
    cs = ctypes.c_void_p(w.comm_stream.cuda_stream)
    w.ready = [threading.Event() for _ in range(bsz)]
    w.arrived = [None] * bsz
    w.bb_free_flag = [threading.Event() for _ in range(bsz)]
    w.bb_free_ev = [None] * bsz
    w.err = []
    rows_h, stage_h, bb = hb["rows_h"], hb["stage_h"], hb["bb"]
    n_p, n_first, skip_opt = w.n_p, w.n_first, w.skip_opt

    # Two helper threads: the PREPARER drives the host pool through the batch's late rows chunk by chunk (deferred row
    # optimizer + copy into pinned staging; it never waits for the GPU, so the pool -- the scarcer resource: 16 CPUs prepare
    # ~170 M rows/s = 33 GB/s against the link's 57 -- works without gaps), the COPIER sends finished chunks to the bounce
    # table in order, the first chunk of group k+1 only once camera k has emptied it.
    chunks = []  # (group, c0, c1, first_of_group, last_of_group)
    k0 = 0
    for i in range(bsz):
        k1 = k0 + n_first[i]
        cc = list(range(k0, k1, _CHUNK_ROWS))
        for j, c0 in enumerate(cc):
            chunks.append((i, c0, min(k1, c0 + _CHUNK_ROWS), j == 0, j == len(cc) - 1, k0))
        if not cc:
            chunks.append((i, k0, k0, True, True, k0))
        k0 = k1
    prepared = [threading.Event() for _ in chunks]

    def preparer():
        try:
            if prev is not None:  # gradients of the previous batch must have landed before any row is stepped
                prev.synchronize()
            for n_, (i, c0, c1, _f, _l, _g0) in enumerate(chunks):
                if c1 > c0:
                    _tp = time.perf_counter()
                    g.host_rows_prepare(rows_h[c0:c1], stage_h[c0:c1], to_step=step - 1,
                                        next_g_step=0 if skip_opt else step, sync_grads=False)
                    _lib.STATS["host_prepare_s"] = _lib.STATS.get("host_prepare_s", 0.0) + time.perf_counter() - _tp
                prepared[n_].set()
        except BaseException as e:  # surface in the main thread
            w.err.append(e)
            for ev_ in prepared:
                ev_.set()

    def feeder():
        try:
            for n_, (i, c0, c1, first, last, g0) in enumerate(chunks):
                prepared[n_].wait()
                if w.err:
                    break
                if first and i > 0 and c1 > c0:  # camera i-1 has emptied the bounce table (device order: its copy's event)
                    w.bb_free_flag[i - 1].wait()
                    if w.err:
                        break
                    w.comm_stream.wait_event(w.bb_free_ev[i - 1])
                if c1 > c0:
                    dst0 = (n_p if i == 0 else 0) + c0 - g0
                    _lib.check(L.clmgs_memcpy_async(cs, _ptr(bb[dst0:dst0 + c1 - c0]), _ptr(stage_h[c0:c1]), (c1 - c0) * 192, 1))
                if last:
                    ev = torch.cuda.Event()
                    ev.record(w.comm_stream)
                    w.arrived[i] = ev
                    w.ready[i].set()
        except BaseException as e:  # surface in the main thread
            w.err.append(e)
        finally:
            if w.err:
                for r in w.ready:
                    r.set()

    w.prep_thread = threading.Thread(target=preparer, name="clmgs-hostwin-prepare")
    w.prep_thread.start()
    w.worker = threading.Thread(target=feeder, name="clmgs-hostwin-feeder")
    w.worker.start()
    w.new_spec = None
    if w.spec_rows is not None:
        n_s = int(w.spec_rows.shape[0])
        s_rows_h, s_stage_h = hb["spec_rows_h"][:n_s], hb["spec_stage_h"][:n_s]
        ev_list = torch.cuda.Event()
        ev_list.record(w.default_stream)  # the row list has reached pinned memory once this has passed
        spec_err, spec_done = [], torch.cuda.Event()

        def speculate():
            try:
                w.prep_thread.join()      # after this batch's own rows: the host pool is free from here on
                ev_list.synchronize()
                for c0 in range(0, n_s, _CHUNK_ROWS):  # brought up to date + copied to pinned staging right away ...
                    c1 = min(n_s, c0 + _CHUNK_ROWS)
                    _tp = time.perf_counter()
                    g.host_rows_prepare(s_rows_h[c0:c1], s_stage_h[c0:c1], to_step=step, next_g_step=step + 1, sync_grads=False)
                    _lib.STATS["host_prepare_s"] = _lib.STATS.get("host_prepare_s", 0.0) + time.perf_counter() - _tp
                w.worker.join()           # ... sent once this batch's copies are queued (stream order on the side stream)
                w.bb_free_flag[bsz - 1].wait()  # and the last camera has taken its rows out of the bounce table
                if w.err:
                    return
                w.comm_stream.wait_event(w.bb_free_ev[bsz - 1])
                for c0 in range(0, n_s, _CHUNK_ROWS):
                    c1 = min(n_s, c0 + _CHUNK_ROWS)
                    _lib.check(L.clmgs_memcpy_async(cs, _ptr(bb[c0:c1]), _ptr(s_stage_h[c0:c1]), (c1 - c0) * 192, 1))
                spec_done.record(w.comm_stream)
            except BaseException as e:
                spec_err.append(e)

        th = threading.Thread(target=speculate, name="clmgs-hostwin-speculate")
        w.new_spec = dict(cam0=id(w.hint[0]), N=w.N, rows=w.spec_rows, n=n_s, thread=th, gen=hb["gen"], event=spec_done,
                          err=spec_err, rows_h=hb["spec_rows_h"])


# ------------------------------------------------------------------------------------------------------------ stage 3
def _cameras(w):
    """One camera after the other (the mode is bound by the host link, not by the GPU)."""
    from ...fused import train_one_camera
    from .engine import _zero_small_grads
    g, hb, bsz, N = w.gaussians, w.hb, w.bsz, w.N
    pt, gt, bb, cap_p = hb["pt"], hb["gt"], hb["bb"], w.base_a  # (cap_p below: first row of the arrival region)
    cur_slot = g._hwin_tabs["cur_slot"]
    ds, out_stream = w.default_stream, w.out_stream
    _zero_small_grads(g)
    if w.spec is not None:
        ds.wait_event(w.spec["event"])      # the speculative block has landed in the bounce table
    if w.new_spec is not None:
        w.new_spec["thread"].start()
    if w.prev_grads is not None:            # the previous batch's hand-back still reads the gradient table
        ds.wait_event(w.prev_grads)
    losses, l0, c_prev, ev_out = [], 0, 0, None
    try:
        return _camera_loop(w, losses, l0, c_prev, ev_out, pt, gt, bb, cap_p, cur_slot, ds, out_stream)
    finally:  # whatever happens, no helper thread may be left waiting for this thread
        for f in w.bb_free_flag:
            f.set()


def _camera_loop(w, losses, l0, c_prev, ev_out, pt, gt, bb, cap_p, cur_slot, ds, out_stream):
    from ...fused import train_one_camera
    g, bsz, N = w.gaussians, w.bsz, w.N
    for k in range(bsz):
        with _lib.host_region("wait_rows"):
            w.ready[k].wait()
        if w.err:
            for f in w.bb_free_flag:
                f.set()
            w.worker.join()
            w.prep_thread.join()
            raise w.err[0]
        if w.arrived[k] is not None:
            ds.wait_event(w.arrived[k])
        n_arr = (w.n_p if k == 0 else 0) + w.n_first[k]
        if n_arr:
            pt[cap_p:cap_p + n_arr].copy_(bb[:n_arr])            # bounce -> this camera's arrival block
        ev = torch.cuda.Event()
        ev.record(ds)
        w.bb_free_ev[k] = ev
        w.bb_free_flag[k].set()                                   # the feeder may send camera k+1's rows
        if ev_out is not None:
            ds.wait_event(ev_out)                                 # camera k-1's hand-back has read its gradient rows
        if n_arr:
            gt[cap_p:cap_p + n_arr].zero_()
        f = w.filters[k]
        sh_index = cur_slot[f]
        losses.append(train_one_camera(g, w.cameras[k], f, pt, 1, gt, w.background, w.cameras[k].original_image, sh_index=sh_index))
        # rows whose LAST camera this was: their gradient rows go home now (plain stores, side stream)
        l1 = l0 + w.n_last[k]
        if l1 > l0:
            out_stream.wait_stream(ds)
            with torch.cuda.stream(out_stream):
                clm_kernels._rows("clmgs_rows_gather", w.parameters_grad_buffer[:N, :], gt, w.rows_by_last[l0:l1],
                                  w.leave_slots[l0:l1], int(getattr(w.args, "host_scatter_grid", 0)))
                ev_out = torch.cuda.Event()
                ev_out.record(out_stream)
        l0 = l1
        # arrivals a later camera uses again: parameters and partial gradient move to their P slot
        c1 = w.surv_cum[k]
        if c1 > c_prev:
            src, dst = w.surv_src[c_prev:c1], w.surv_dst[c_prev:c1]
            clm_kernels._rows("clmgs_rows_gather", pt, pt, dst, src, 0)
            clm_kernels._rows("clmgs_rows_gather", gt, gt, dst, src, 0)
            cur_slot[w.surv_rows[c_prev:c1]] = dst.to(torch.int32)
        c_prev = c1
    w.worker.join()
    w.prep_thread.join()
    ev_g = torch.cuda.Event()
    ev_g.record(out_stream)
    g._host_grads_event = ev_g
    g._hwin_keep = (w.filters, w.rows_by_last, w.leave_slots, w.surv_rows, w.surv_src, w.surv_dst, w.late_all, w.touched)
    if w.new_spec is not None:
        g._hwin_spec = w.new_spec
    return losses


# ------------------------------------------------------------------------------------- the HBM-resident prefix (budget)
def _prefix_head(w):
    """Before the cameras: the resident rows this batch renders from are brought up to the previous step by the deferred
    row optimizer of the HBM engine (their waiting gradient at its own step, then the zero-gradient steps they skipped;
    consumed gradient rows are cleared, so the cameras accumulate into zeros).  Runs while the host pool prepares the
    first camera's rows."""
    px, g = w.px, w.gaussians
    if px is None:
        return
    K, hb = px["K"], w.hb
    if px["fill"]:  # first batch after a load: the rows themselves (the moments came with hbm_prefix_ensure)
        hb["pt"][:K].copy_(g._parameters.data[:K])
        hb["gt"][:K].zero_()
        px["fill"] = False
    if not w.args.sparse_adam:
        g.hbm_prefix_catch_up(w.prefix_rows, w.step - 1)


def _prefix_tail(w):
    """After the cameras: the resident rows' gradient of this batch waits in the table, stamped with this step (dense Adam:
    applied at the row's next touch / at a flush), or is applied now (sparse_adam: only touched rows ever step)."""
    px, g = w.px, w.gaussians
    if px is None or w.prefix_rows.numel() == 0:
        return
    K, hb, rows = px["K"], w.hb, w.prefix_rows
    if w.skip_opt:  # test hook: the unscaled sums go where the host rows' gradients go, and nothing is left behind
        idx = rows.long()
        clm_kernels._rows("clmgs_rows_gather", w.parameters_grad_buffer[:w.N, :], hb["gt"], idx, idx, 0)
        torch.cuda.synchronize()
        hb["gt"].index_fill_(0, idx, 0.0)
        return
    px["dirty"] = True
    if w.args.sparse_adam:
        opt = g.optimizer.cpu_adam
        gr = opt.param_groups[0]
        clm_kernels.adam_rows(hb["pt"][:K], hb["gt"][:K], px["m"], px["v"], rows, opt._col_lr(w.dev), gr["betas"][0],
                              gr["betas"][1], gr["eps"], w.step, gr["bias_correction"], 1.0 / float(w.bsz), True)
    else:
        utils.fill_rows(px["g_step"], rows.long(), w.step)


def train_one_batch_host_windowed(gaussians, scene, batched_cameras, parameters_grad_buffer, background, pipe_args,
                                  comm_stream, perm_generator, args):
    """-> (losses, ordered_cams, sparsity); see the module docstring."""
    from . import engine as E
    from ... import dp
    assert not dp.active(), "camera-DP is built for sh_residency='hbm' (every rank holds a full replica)"
    assert gaussians.deferred_host_rows
    assert getattr(args, "fused_front_end", True), "the host-resident mode runs the fused front end"
    assert args.lr_scale_mode == "sqrt", "Overlap CPUAdam only supports sqrt lr scaling"
    assert not args.stop_update_param, "Overlap CPUAdam does not support stop_update_param"
    w = _Win()
    w.gaussians, w.args, w.cameras, w.background = gaussians, args, list(batched_cameras), background
    w.parameters_grad_buffer, w.comm_stream, w.perm_generator = parameters_grad_buffer, comm_stream, perm_generator
    w.bsz, w.N, w.dev = len(batched_cameras), gaussians._xyz.shape[0], gaussians._xyz.device
    w.skip_opt = bool(getattr(args, "debug_skip_optimizer", False))  # test hook, see engine._train_one_batch_hbm
    w.default_stream = torch.cuda.current_stream()
    if getattr(gaussians, "_host_out_stream", None) is None:
        gaussians._host_out_stream = torch.cuda.Stream()
    w.out_stream = gaussians._host_out_stream
    w.hint = getattr(gaussians, "_next_batch_hint", None)
    gaussians._next_batch_hint = None
    if w.skip_opt or not getattr(args, "host_speculative_prefetch", True) or getattr(args, "reference_camera_order", False):
        w.hint = None
    w.px = gaussians.hbm_prefix_ensure()  # (sh_hbm_budget_gb; a first call flushes the host rows and loads the prefix)
    w.K = w.px["K"] if w.px is not None else 0
    with torch.no_grad():
        _plan(w)
        _start_feeders(w)
        _prefix_head(w)
        losses = _cameras(w)
        _prefix_tail(w)
    if w.skip_opt:
        torch.cuda.synchronize()
        return losses, w.ordered_cams, w.sparsity
    visibility_mask = None
    if args.sparse_adam:
        visibility_mask = torch.zeros((w.N,), dtype=torch.bool, device=w.dev)
        utils.fill_rows(visibility_mask, w.touched_all, True)
    E._gpu_adam_step(gaussians, args, visibility_mask)
    gaussians.invalidate_small_packed()
    E._mark_batch(gaussians)
    row_adam = gaussians.optimizer.cpu_adam
    row_adam.global_step = w.step
    row_adam.state[gaussians._parameters]["step"] = w.step
    return losses, w.ordered_cams, w.sparsity
