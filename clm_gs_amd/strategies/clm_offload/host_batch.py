"""Host-resident SH rows staged as the UNION of a batch's rows (sh_residency="host", host_staging="batch").

What the reference's retention pipeline and cpu-adam thread do (strategies/clm_offload/engine.py:494-508, 622-641, 789-825,
301-335), re-shaped around what this machine's host link and host cores reward; rounds 2-5's form of the mode, kept beside
the per-camera windows of host_window.py (the default: the same link traffic from half the staging memory).

* Every row the batch touches crosses the link ONCE per direction (the minimum the H / D / G retention sets aim at, reached
  without camera re-ordering): the union of the batch's filters is staged, grouped by the camera that uses a row FIRST
  (host -> GPU) and LAST (GPU -> host), so camera k renders as soon as its own new rows have arrived while the rows only
  later cameras need are still being prepared, and a row's gradient leaves right after the last camera that contributes.
* Host -> GPU: a feeder thread drives the host thread pool, which brings the touched rows up to date (DEFERRED row
  optimizer: the gradient a row received in an earlier batch is applied, and the zero-gradient Adam steps it has skipped
  since are replayed, only now -- one read and one write of p / m / v per touched row and batch, where the dense reference
  optimizer streams all N rows every batch) and copies them into a contiguous pinned staging buffer, chunk by chunk; each
  finished chunk goes to the GPU with a hipMemcpyAsync on the side stream (SDMA engine: no compute unit is taken from the
  renderer) while the pool works on the next chunk.
* SPECULATIVE PREFETCH (engine.hint_next_batch): the host pool is busy for the first ~60 % of a batch and the link for less;
  the rows of the NEXT batch that the present one does not touch (their state is final until then) are prepared and shipped
  in that idle time into the second staging table.  When the next batch arrives, its exact selection is compared with what
  was staged: only the LATE rows (touched by both batches: their gradient had to come home first -- plus the few a position
  update moved into view) go through the feeder at batch time, so the first camera waits for a third of its rows.
* The cameras render from / accumulate into GPU staging tables ([T,48] parameters and gradients, row -> slot through an
  index the fused front end follows).
* GPU -> host: zero-copy scatter STORES of the gradient rows into the pinned gradient table (plain stores, no
  read-modify-write over the link, no host pass), on a second side stream; they wait there, stamped with this batch's step,
  until the row is needed again.

Four stages, as in host_window.py: plan (filters, verdict on the speculation, row groups, slots) -> feeders (the two helper
threads) -> cameras -> the small attributes' optimizer step.
"""
import ctypes
import threading
import time

import torch

from ... import _lib, clm_kernels, utils
from ..base_engine import select_filters


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class _HostBatch:
    """State of one batch-staged host-resident batch, handed from stage to stage."""


# ------------------------------------------------------------------------------------------------------------ stage 1
def _plan(h):
    """Visibility filters, the verdict on what was staged speculatively, the rows grouped by first / last camera and
    their slots, the staging tables, the row list of the hinted next batch."""
    from . import engine as E
    g, args, dev, bsz, N, L = h.gaussians, h.args, h.dev, h.bsz, h.N, _lib.lib()
    with _lib.host_region("select_filters"):
        h.filters, touched_rows = select_filters(h.cameras, g._xyz.detach(), g._scaling.detach(), g._rotation.detach())
    h.touched_rows = touched_rows
    T = int(touched_rows.shape[0])
    _lib.STATS.setdefault("touched_rows", []).append(T)
    h.sparsity = [len(f) / float(N) for f in h.filters]
    h.ordered_cams = list(range(bsz))
    if getattr(args, "reference_camera_order", False):  # the reference's TSP order (engine.py:135-298)
        _, h.cameras, h.filters, h.sparsity, h.ordered_cams = E.order_calculation(
            list(h.filters), list(h.cameras), N, bsz, h.perm_generator, args)[:5]
    filters = h.filters
    # ---- what was staged for this batch while the previous one rendered
    spec = getattr(g, "_host_spec", None)
    gen0 = (getattr(g, "_host_bufs", None) or {}).get("gen")
    if spec is not None and (spec["key"] != tuple(sorted(id(c) for c in h.cameras)) or spec["N"] != N or spec["gen"] != gen0):
        E._drop_speculation(g)
        spec = None
    g._host_spec = None
    n_p = 0
    if spec is not None:
        with _lib.host_region("spec_join"):
            spec["thread"].join()
        if spec["err"]:
            raise spec["err"][0]
        n_p = spec["n"]
    ht = E._host_tables(g, dev)
    slot_of, mark, in_spec = ht["slot_of"], ht["mark"], ht["in_spec"]
    with _lib.host_region("host_groups"):
        mark.zero_()
        utils.fill_rows(mark, touched_rows, True)          # rows this batch touches
        wasted = touched_rows[:0]
        staged_mask = None
        if spec is not None and n_p:
            P = spec["rows"]                                 # staged rows, slot k = P[k]
            wasted = P[~mark[P]]                             # staged but not touched: no gradient will land
            in_spec.zero_()
            utils.fill_rows(in_spec, P, True)
            staged_mask = in_spec
        # ---- ONE library call (clmgs_host_groups) groups the rows: the rows still to be staged ("late": not in the
        # speculative block) by the camera that uses them FIRST (slot order), all touched rows by the camera that uses them
        # LAST (hand-back order), slot_of[] of the late rows, and the 2 bsz + 1 group sizes on the device -- two stable
        # one-digit radix sorts.  Rounds 2-3 did this with ~60 torch index ops (float64 log2 of the bitmap words, two 64-bit
        # sorts, bincounts, boolean selections): 24 ms of a 114 ms batch during which the GPU did little else.
        bitmap = E._encode_bitmap(filters, N, bsz)                     # MSB = camera 0
        late_all = torch.empty((T,), dtype=torch.int32, device=dev)
        rows_by_last32 = torch.empty((T,), dtype=torch.int32, device=dev)
        counts = torch.empty((2 * bsz + 1,), dtype=torch.int64, device=dev)
        tb = L.clmgs_host_groups_temp_bytes(T)
        tmp = torch.empty((tb,), dtype=torch.uint8, device=dev)
        _t0 = time.perf_counter()

        # (slot_of of the late rows needs slot0 = n_p, which is only final once the staging tables are known not to have
        #  been re-allocated: n_p is re-checked below and the call repeated in that rare case)
        def group(slot0, staged):
            _lib.check(L.clmgs_host_groups(_lib.stream(), T, _lib.dptr(touched_rows, torch.int64), _lib.dptr(bitmap),
                                           bitmap.element_size(), bsz, _lib.dptr(staged.view(torch.uint8), None, True)
                                           if staged is not None else None, int(slot0), _lib.dptr(late_all),
                                           _lib.dptr(rows_by_last32), _lib.dptr(slot_of), _lib.dptr(counts),
                                           _lib.dptr(tmp), tb))
        group(n_p, staged_mask)
        cl = counts.tolist()                                          # one host read: the group sizes
        _lib.STATS["host_wait_s"] += time.perf_counter() - _t0
        n_late = int(cl[2 * bsz])
        hb = E._host_buffers(g, n_p + n_late, dev)
        if spec is not None and spec["gen"] != hb["gen"]:
            # the staging tables had to grow: what was staged went with the old ones -- everything is late
            # (re-preparing a current row is the identity; the stamps of the touched rows stay right)
            n_p, spec = 0, None
            group(0, None)
            cl = counts.tolist()
            n_late = int(cl[2 * bsz])
        if spec is not None and n_p:
            slot_of[spec["rows"]] = torch.arange(n_p, dtype=torch.int32, device=dev)
        h.sh_stage = hb["sh_stage"][hb["cur"]]
        T_slots = n_p + n_late
        _lib.STATS.setdefault("host_late_rows", []).append(n_late)
        h.g_stage = hb["g_stage"][:T_slots]
        h.rows32 = late_all[:n_late]                                 # slot n_p + k holds row rows32[k]
        h.rows_h, h.stage_h = hb["rows_h"][:n_late], hb["stage_h"][:n_late]
        if n_late:
            _lib.check(L.clmgs_memcpy_async(_lib.stream(), _ptr(h.rows_h), _ptr(h.rows32), n_late * 4, 2))
        h.sh_index = [slot_of[f] for f in filters]
        h.rows_by_last = rows_by_last32.to(torch.int64)
        h.slots_by_last = slot_of[h.rows_by_last]
        # ---- the NEXT batch's rows on the positions current now; what this batch touches cannot be staged early
        h.spec_rows = None
        t_next = None
        if h.hint is not None:
            try:
                _, t_next = select_filters(h.hint, g._xyz.detach(), g._scaling.detach(), g._rotation.detach())
            except AssertionError:  # a hinted camera sees nothing yet: that batch will complain itself
                t_next = None
        if t_next is not None:
            spec_rows = t_next[~mark[t_next]]
            n_s = int(spec_rows.shape[0])
            del t_next
            if n_s and n_s <= hb["cap"]:
                E._host_spec_buffers(hb, dev)
                _lib.check(L.clmgs_memcpy_async(_lib.stream(), _ptr(hb["spec_rows_h"][:n_s]),
                                                _ptr(spec_rows.to(torch.int32)), n_s * 4, 2))
                h.spec_rows = spec_rows
        _t0 = time.perf_counter()
        wasted_h = wasted.cpu() if wasted.numel() else None
        _lib.STATS["host_wait_s"] += time.perf_counter() - _t0
    h.n_first, h.n_last = cl[:bsz], cl[bsz:2 * bsz]
    h.n_p, h.spec, h.hb = n_p, spec, hb
    if wasted_h is not None:  # staged for this batch but not touched by it: they expect no gradient after all
        g._host_g_step[wasted_h] = 0


# ------------------------------------------------------------------------------------------------------------ stage 2
def _start_feeders(h):
    """Feeder thread: prepare + stage + hipMemcpyAsync chunk by chunk, one event per first-use group.  Speculation thread
    (started by the camera stage): the hinted next batch's rows, after this batch's own, in the pool's idle time."""
    from .engine import _HOST_CHUNK_ROWS
    g, hb, bsz, L = h.gaussians, h.hb, h.bsz, _lib.lib()
    prev = g._host_grads_event
    if prev is not None:  # the previous batch's scatter still reads g_stage
        h.default_stream.wait_event(prev)
    h.g_stage.zero_()
    row_adam = g.optimizer.cpu_adam
    h.step = step = row_adam.global_step + 1
    h.comm_stream.wait_stream(h.default_stream)
    cs = ctypes.c_void_p(h.comm_stream.cuda_stream)
    h.ready = [threading.Event() for _ in range(bsz)]
    h.ev_group = [None] * bsz
    h.err = []
    n_p, n_first, skip_opt = h.n_p, h.n_first, h.skip_opt
    rows_h, stage_h, sh_stage = h.rows_h, h.stage_h, h.sh_stage

    def feeder():
        try:
            if prev is not None:  # gradients of the previous batch must have landed before any row is stepped
                prev.synchronize()
            k0 = 0
            for i in range(bsz):
                k1 = k0 + n_first[i]
                for c0 in range(k0, k1, _HOST_CHUNK_ROWS):
                    c1 = min(k1, c0 + _HOST_CHUNK_ROWS)
                    _tp = time.perf_counter()
                    g.host_rows_prepare(rows_h[c0:c1], stage_h[c0:c1], to_step=step - 1,
                                        next_g_step=0 if skip_opt else step, sync_grads=False)
                    _lib.STATS["host_prepare_s"] = _lib.STATS.get("host_prepare_s", 0.0) + time.perf_counter() - _tp
                    _lib.check(L.clmgs_memcpy_async(cs, _ptr(sh_stage[n_p + c0:n_p + c1]), _ptr(stage_h[c0:c1]),
                                                    (c1 - c0) * 192, 1))
                ev = torch.cuda.Event()
                ev.record(h.comm_stream)
                h.ev_group[i] = ev
                h.ready[i].set()
                k0 = k1
        except BaseException as e:  # surface in the main thread
            h.err.append(e)
            for r in h.ready:
                r.set()

    if prev is not None:
        g._host_grads_event = None  # consumed by the feeder above
    h.worker = threading.Thread(target=feeder, name="clmgs-host-feeder")
    h.worker.start()
    h.new_spec = None
    if h.spec_rows is not None:
        nb = 1 - hb["cur"]
        n_s = int(h.spec_rows.shape[0])
        s_rows_h, s_stage_h, s_dst = hb["spec_rows_h"][:n_s], hb["spec_stage_h"][:n_s], hb["sh_stage"][nb]
        ev_list = torch.cuda.Event()
        ev_list.record(h.default_stream)   # the row list has reached pinned memory once this has passed
        spec_err, spec_done = [], torch.cuda.Event()

        def speculate():
            try:
                h.worker.join()            # after this batch's own rows
                ev_list.synchronize()
                for c0 in range(0, n_s, _HOST_CHUNK_ROWS):
                    c1 = min(n_s, c0 + _HOST_CHUNK_ROWS)
                    _tp = time.perf_counter()
                    g.host_rows_prepare(s_rows_h[c0:c1], s_stage_h[c0:c1], to_step=step, next_g_step=step + 1,
                                        sync_grads=False)
                    _lib.STATS["host_prepare_s"] = _lib.STATS.get("host_prepare_s", 0.0) + time.perf_counter() - _tp
                    _lib.check(L.clmgs_memcpy_async(cs, _ptr(s_dst[c0:c1]), _ptr(s_stage_h[c0:c1]), (c1 - c0) * 192, 1))
                spec_done.record(h.comm_stream)
            except BaseException as e:
                spec_err.append(e)

        th = threading.Thread(target=speculate, name="clmgs-host-speculate")
        h.new_spec = dict(key=tuple(sorted(id(c) for c in h.hint)), N=h.N, rows=h.spec_rows, n=n_s, buf=nb, thread=th,
                          gen=hb["gen"], event=spec_done, err=spec_err, rows_h=hb["spec_rows_h"])


# ------------------------------------------------------------------------------------------------------------ stage 3
def _cameras(h):
    """One camera after the other (the mode is bound by the host side, not by the GPU: the camera pipeline of the HBM mode
    would only add its per-camera buffers to the peak)."""
    from ...fused import train_one_camera
    from .engine import _zero_small_grads
    g, bsz, N, ds, out_stream = h.gaussians, h.bsz, h.N, h.default_stream, h.out_stream
    _zero_small_grads(g)
    if h.spec is not None:
        ds.wait_event(h.spec["event"])  # the staged block has landed
    if h.new_spec is not None:
        h.new_spec["thread"].start()
    losses = []
    l0 = 0
    for i in range(bsz):
        with _lib.host_region("wait_rows"):
            h.ready[i].wait()
        if h.err:
            h.worker.join()
            raise h.err[0]
        ds.wait_event(h.ev_group[i])
        losses.append(train_one_camera(g, h.cameras[i], h.filters[i], h.sh_stage, 1, h.g_stage, h.background,
                                       h.cameras[i].original_image, sh_index=h.sh_index[i]))
        # rows whose LAST camera this was: their gradient rows go home now (plain stores, side stream)
        l1 = l0 + h.n_last[i]
        if l1 > l0:
            out_stream.wait_stream(ds)
            with torch.cuda.stream(out_stream):
                # (launch width: the kernel is bound by the link -- 51 GB/s of stores -- and its stalled waves take wave
                # slots from the render kernels next to it (preprocess_fwd 0.3 -> 4.7 ms in the trace), but narrowing it to
                # the reference's grid_size_H = 32 / 128 / 512 workgroups measured 101.3 / 100.6 / 101.1 ms per batch
                # against 99.1 at full width: the batch is bound by the link either way)
                clm_kernels._rows("clmgs_rows_gather", h.parameters_grad_buffer[:N, :], h.g_stage,
                                  h.rows_by_last[l0:l1], h.slots_by_last[l0:l1].to(torch.int64),
                                  int(getattr(h.args, "host_scatter_grid", 0)))
        l0 = l1
    h.worker.join()
    ev_g = torch.cuda.Event()
    ev_g.record(out_stream)
    g._host_grads_event = ev_g
    g._host_keep = (h.rows32, h.sh_index, h.touched_rows, h.filters, h.rows_by_last, h.slots_by_last)
    if h.new_spec is not None:
        h.hb["cur"] = h.new_spec["buf"]      # the next batch renders from the table its staged rows are landing in
        g._host_spec = h.new_spec
    return losses


def train_one_batch_host_batched(gaussians, scene, batched_cameras, parameters_grad_buffer, background, pipe_args,
                                 comm_stream, perm_generator, args):
    """-> (losses, ordered_cams, sparsity); see the module docstring."""
    from . import engine as E
    from ... import dp
    assert not float(getattr(args, "sh_hbm_budget_gb", 0.0) or 0.0), "sh_hbm_budget_gb is built on host_staging='window'"
    assert not dp.active(), "camera-DP is built for sh_residency='hbm' (every rank holds a full replica)"
    assert gaussians.deferred_host_rows
    assert getattr(args, "fused_front_end", True), "the host-resident mode runs the fused front end"
    assert args.lr_scale_mode == "sqrt", "Overlap CPUAdam only supports sqrt lr scaling"
    assert not args.stop_update_param, "Overlap CPUAdam does not support stop_update_param"
    h = _HostBatch()
    h.gaussians, h.args, h.cameras, h.background = gaussians, args, list(batched_cameras), background
    h.parameters_grad_buffer, h.comm_stream, h.perm_generator = parameters_grad_buffer, comm_stream, perm_generator
    h.bsz, h.N, h.dev = len(batched_cameras), gaussians._xyz.shape[0], gaussians._xyz.device
    h.skip_opt = bool(getattr(args, "debug_skip_optimizer", False))  # test hook, see engine._train_one_batch_hbm
    h.default_stream = torch.cuda.current_stream()
    if getattr(gaussians, "_host_out_stream", None) is None:
        gaussians._host_out_stream = torch.cuda.Stream()
    h.out_stream = gaussians._host_out_stream
    h.hint = getattr(gaussians, "_next_batch_hint", None)
    gaussians._next_batch_hint = None
    if h.skip_opt or not getattr(args, "host_speculative_prefetch", True):
        h.hint = None
    with torch.no_grad():
        _plan(h)
        _start_feeders(h)
        losses = _cameras(h)
    if h.skip_opt:
        torch.cuda.synchronize()
        return losses, h.ordered_cams, h.sparsity
    visibility_mask = None
    if args.sparse_adam:
        visibility_mask = torch.zeros((h.N,), dtype=torch.bool, device=h.dev)
        utils.fill_rows(visibility_mask, h.touched_rows, True)
    E._gpu_adam_step(gaussians, args, visibility_mask)
    gaussians.invalidate_small_packed()
    E._mark_batch(gaussians)
    row_adam = gaussians.optimizer.cpu_adam
    row_adam.global_step = h.step
    row_adam.state[gaussians._parameters]["step"] = h.step
    return losses, h.ordered_cams, h.sparsity
