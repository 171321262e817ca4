"""Independent-camera data parallelism (net-new: the reference is single GPU).

One process per GPU, every rank holds a full replica and renders its own cameras of the global
batch.  ONE exchange per batch, before the optimizer: sum of the dense [N,11] small gradients,
OR of the touched-row mask, and sum of the SH gradient rows restricted to the union of touched
rows.  Densification statistics accumulate locally and are reduced only right before
densify_and_prune (sum / sum / max).  Everything is plain torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; the CPU tests use gloo), device-agnostic tensors.

Three exchanges (engine option in brackets), same result after flush_lazy_rows():
  all-reduce      [default]            every rank steps every row: 240 B x globally touched rows, all-reduced
  owner-computes  [dp_owner_computes]  rows owned by index range; all-gather params / reduce-scatter grads
  locality        [dp_locality]        owner-computes + point-to-point: a rank fetches ONLY the rows its own
                                       cameras touch outside its range from their owners (all_to_all), sends
                                       their gradients back.  The small attributes are stepped by their
                                       owner too [dp_small_owner, default]: foreign copies go stale inside
                                       Adam's step bound, every batch starts by fetching the current lines of
                                       the rows that may be visible (small_fetch, step S); or [dp_small_owner
                                       off, round 3] the owners publish the summed small-attribute gradients of
                                       their touched rows (all-gather, step F) and every rank steps every row.
                                       With Z-ordered rows a rank's range is a spatial region; assign_cameras()
                                       deals every camera to the rank owning most of its rows, so most touched
                                       rows never travel.
WIRE counts the bytes this rank SENDS per exchange (ring model for all-reduce), see wire_bytes().
"""
import os
import time as _time

import torch
import torch.distributed as dist

from . import utils


def _take_into(out, table, idx):
    """out[:] = table[idx] without a temporary when the fast path applies (out: contiguous [len(idx), cols] view)."""
    if table.is_cuda and table.dtype == torch.float32 and table.dim() == 2 and table.shape[1] % 4 == 0 \
            and table.is_contiguous() and out.is_contiguous() and idx.numel():
        from .clm_kernels import _rows
        _rows("clmgs_rows_gather", out, table, None, idx.contiguous(), 0)
    elif idx.numel():
        out.copy_(utils.take_rows(table, idx))


def _take(table, idx):
    """table[idx] for the exchanges' pack steps: the library's own row mover (64-bit row arithmetic, one pass at HBM
    speed; clmgs_rows_gather) for fp32 device tables with 16-byte rows, the chunked torch form otherwise (CPU tensors
    of the gloo tests, stamps, odd widths)."""
    if table.is_cuda and table.dtype == torch.float32 and table.dim() == 2 and table.shape[1] % 4 == 0 \
            and table.is_contiguous() and idx.numel():
        from .clm_kernels import _rows
        out = table.new_empty((idx.numel(), table.shape[1]))
        _rows("clmgs_rows_gather", out, table, None, idx.contiguous(), 0)
        return out
    return utils.take_rows(table, idx)


def _put(table, idx, src):
    """table[idx] = src (unique ids), the unpack twin of _take."""
    if table.is_cuda and table.dtype == torch.float32 and table.dim() == 2 and table.shape[1] % 4 == 0 \
            and table.is_contiguous() and src.dtype == torch.float32 and idx.numel():
        from .clm_kernels import _rows
        _rows("clmgs_rows_gather", table, src.contiguous(), idx.contiguous(), None, 0)
        return
    utils.put_rows(table, idx, src)


WIRE = {}  # exchange kind -> bytes this rank has sent since reset_wire()

# Per-phase timing of the exchange (bench.py `dp.phase_ms`): PHASES = {} switches it on.  Every phase() block records an
# event pair on the stream it is ENQUEUED on (a collective of the nccl backend runs on the backend's own stream, which
# waits for the current stream and is waited for by it: the pair brackets it) and the host wall time of the block (the
# host reads of the plan, the host-blocking collectives of gloo).  Off (None): no event, no clock -- the product path.
PHASES = None


class phase:
    __slots__ = ("name", "t0", "e0")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if PHASES is not None:
            self.t0 = _time.perf_counter()
            self.e0 = None
            if torch.cuda.is_available():
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if PHASES is not None:
            e1 = None
            if self.e0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(torch.cuda.current_stream())
            PHASES.setdefault(self.name, []).append((_time.perf_counter() - self.t0, self.e0, e1))
        return False


def phase_summary(n_steps):
    """-> {phase: {"device_ms": per step, "host_ms": per step, "calls": n}}; call after a device synchronisation."""
    out = {}
    for name, recs in (PHASES or {}).items():
        dev = sum(e0.elapsed_time(e1) for _, e0, e1 in recs if e0 is not None)
        out[name] = {"device_ms": round(dev / max(1, n_steps), 3),
                     "host_ms": round(sum(r[0] for r in recs) * 1e3 / max(1, n_steps), 3), "calls": len(recs)}
    return out


def _count(kind, nbytes):
    WIRE[kind] = WIRE.get(kind, 0) + int(nbytes)


def reset_wire():
    WIRE.clear()


def wire_bytes():
    """-> {kind: bytes sent by this rank, ..., "total": sum}.  Model: an all-reduce of B bytes over G ranks
    sends 2 (G-1)/G B per rank (reduce-scatter + all-gather, ring or full mesh alike); an all-gather /
    reduce-scatter with a per-rank block of c bytes sends (G-1) c; an all_to_all sends what leaves the rank."""
    d = dict(WIRE)
    d["total"] = sum(WIRE.values())
    return d


def _allreduce_bytes(nbytes):
    g = max(1, world_size())
    return 2.0 * (g - 1) / g * nbytes


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active():
    """True when the engines must run the exchange: more than one rank, or CLMGS_DP_FORCE=1 with an
    initialised process group (a 1-rank group runs every collective as the identity -- this is how the
    RCCL path is exercised on a single-GPU box: tests/test_gpu_nccl.py)."""
    if world_size() > 1:
        return True
    return os.environ.get("CLMGS_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized()


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_small_grads(grads, average=True):
    """grads: list of [N,d] tensors -> summed (averaged) over ranks IN PLACE: four collectives
    issued back to back on the tensors themselves, no packing copies (the 44 B/Gaussian of dense
    small gradients are 1.2 GB at 28 M: cat + split would move them three more times).
    average=False leaves the SUM; the engine folds 1/ranks into the Adam gradient scale."""
    ws = world_size()
    if not active():
        return
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True) for g in grads]
    for w in works:
        w.wait()
    _count("all_reduce", sum(_allreduce_bytes(g.numel() * g.element_size()) for g in grads))
    if average:
        for g in grads:
            g /= ws


def allreduce_touched(touched):
    """bool[N] -> OR over ranks (as MAX over uint8)."""
    if not active():
        return touched
    t = touched.to(torch.uint8)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    _count("all_reduce_mask", _allreduce_bytes(t.numel()))
    return t.to(torch.bool)


def allreduce_rows(grad_rows, touched_global, average=True, rows=None):
    """Sum (average) grad_rows[N,48] over ranks, moving only rows in the (global) touched set.
    `rows`: the index list of touched_global if the caller already has it."""
    ws = world_size()
    if not active():
        return
    if rows is None:
        rows = torch.nonzero(touched_global).flatten()
    n_touched = rows.numel()
    if n_touched == 0:
        return
    if 2 * n_touched >= grad_rows.shape[0]:
        # most rows are in play: reduce the whole buffer in place, no pack / unpack copies
        dist.all_reduce(grad_rows, op=dist.ReduceOp.SUM)
        _count("all_reduce", _allreduce_bytes(grad_rows.numel() * grad_rows.element_size()))
        if average:
            grad_rows /= ws
        return
    rows = rows.long()
    buf = _take(grad_rows, rows)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    _count("all_reduce", _allreduce_bytes(buf.numel() * buf.element_size()))
    if average:
        buf /= ws
    _put(grad_rows, rows, buf)


def allreduce_tables_rows(tables, rows, n_total, average=False, dense_above=0.85):
    """Sum (average) several row tables [N, d_i] over ranks, moving ONLY `rows` (the globally
    touched set, identical on every rank): the rows of every table are packed into a contiguous message
    (236+ B per touched Gaussian in all), the collectives issued back to back, then unpacked.
    xGMI all-reduce bandwidth is ~20x below HBM bandwidth, so packing pays until almost every row
    is touched (pack + unpack cost 2 HBM passes; break-even at ~89 % touched): above `dense_above`
    the tables are reduced in place instead."""
    ws = world_size()
    if not active():
        return
    n = rows.numel()
    if n == 0:
        return
    if n >= dense_above * n_total:
        works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True) for t in tables]
        for w in works:
            w.wait()
        _count("all_reduce", sum(_allreduce_bytes(t.numel() * t.element_size()) for t in tables))
        if average:
            for t in tables:
                t /= ws
        return
    rows = rows.long()
    # one packed message per table (contiguous rows: packed and unpacked by the library's row mover in one pass each),
    # the collectives issued back to back
    bufs = [_take(t, rows) for t in tables]
    works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True) for b in bufs]
    for w in works:
        w.wait()
    _count("all_reduce", sum(_allreduce_bytes(b.numel() * b.element_size()) for b in bufs))
    for t, b in zip(tables, bufs):
        if average:
            b /= ws
        _put(t, rows, b)


# ------------------------------------------------------------------ owner-computes exchange
# Rank q OWNS the rows whose id lies in [q*N/G, (q+1)*N/G): only the owner keeps a row's Adam state
# current and runs its optimizer steps (1/G of the row-optimizer work per rank; the state of the other
# rows on a rank is dead weight that a later round can drop: ZeRO-1 for the [N,48] tables).  Per batch,
# for the globally touched rows R (identical, ascending, on every rank):
#   before rendering  all-gather   : owners send the up-to-date parameter rows of R, every rank fills R
#   after rendering   reduce-scatter: every rank sends its gradient rows of R, owners receive the sums
# -- the same bytes on the wire as the all-reduce of the gradient rows (a ring all-reduce IS a
# reduce-scatter + an all-gather), but the second half carries parameters instead of gradients, so no
# rank ever applies an optimizer step to a row it does not own.  Segments are padded to the longest one
# so that the tensor forms of the collectives apply (full-mesh xGMI: 7 peers in parallel).
class OwnerPlan:
    __slots__ = ("rows", "bounds", "chunk", "pos", "lo", "hi", "n_ranks")


def owner_range(n_total, q=None, n_ranks=None):
    n_ranks = world_size() if n_ranks is None else n_ranks
    q = rank() if q is None else q
    return (q * n_total) // n_ranks, ((q + 1) * n_total) // n_ranks


def owner_plan(rows, n_total):
    """rows: ascending int64 row ids (the same on every rank).  One host read (the G+1 segment bounds)."""
    G, r = world_size(), rank()
    cuts = torch.tensor([(q * n_total) // G for q in range(G + 1)], dtype=torch.int64).to(rows.device)
    b = torch.searchsorted(rows, cuts).tolist()
    b[0], b[-1] = 0, int(rows.numel())
    lens = [b[q + 1] - b[q] for q in range(G)]
    pl = OwnerPlan()
    pl.rows, pl.bounds, pl.n_ranks = rows, b, G
    pl.chunk = max(1, max(lens))
    shift = torch.repeat_interleave(torch.tensor([q * pl.chunk - b[q] for q in range(G)], dtype=torch.int64),
                                    torch.tensor(lens, dtype=torch.int64)).to(rows.device)
    pl.pos = torch.arange(rows.numel(), device=rows.device) + shift   # row k of `rows` -> row of the padded buffer
    pl.lo, pl.hi = b[r], b[r + 1]
    return pl


def owner_gather_rows(table, pl):
    """Every rank's table[pl.rows] <- the owners' rows (owners' own rows are unchanged)."""
    if not active():
        return
    W = table.shape[1]
    send = torch.zeros((pl.chunk, W), dtype=table.dtype, device=table.device)
    if pl.hi > pl.lo:
        _take_into(send[: pl.hi - pl.lo], table, pl.rows[pl.lo:pl.hi])
    recv = torch.empty((pl.n_ranks * pl.chunk, W), dtype=table.dtype, device=table.device)
    dist.all_gather_into_tensor(recv, send)
    _count("all_gather", send.numel() * send.element_size() * (pl.n_ranks - 1))
    _put(table, pl.rows, _take(recv, pl.pos))


def owner_reduce_rows(table, pl):
    """Sum of table[pl.rows] over the ranks, delivered to each row's OWNER; on the other ranks the rows
    are zeroed (the gradient has been handed over)."""
    if not active():
        return
    W = table.shape[1]
    buf = torch.zeros((pl.n_ranks * pl.chunk, W), dtype=table.dtype, device=table.device)
    _put(buf, pl.pos, _take(table, pl.rows))
    mine = torch.empty((pl.chunk, W), dtype=table.dtype, device=table.device)
    dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM)
    _count("reduce_scatter", mine.numel() * mine.element_size() * (pl.n_ranks - 1))
    utils.fill_rows(table, pl.rows, 0.0)
    if pl.hi > pl.lo:
        _put(table, pl.rows[pl.lo:pl.hi], mine[: pl.hi - pl.lo])


def owner_gather_dense(tables, n_total):
    """All rows of every table <- their owners' (flush: evaluation, saving, densification)."""
    if not active():
        return
    G = world_size()
    lo, hi = owner_range(n_total)
    chunk = max(((q + 1) * n_total) // G - (q * n_total) // G for q in range(G))
    for t in tables:
        t2 = t.reshape(t.shape[0], -1)
        send = torch.zeros((chunk, t2.shape[1]), dtype=t.dtype, device=t.device)
        send[: hi - lo] = t2[lo:hi]
        recv = torch.empty((G * chunk, t2.shape[1]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(recv, send)
        _count("flush_all_gather", send.numel() * send.element_size() * (G - 1))
        for q in range(G):
            a, b = owner_range(n_total, q, G)
            if q != rank():
                t2[a:b] = recv[q * chunk: q * chunk + (b - a)]


# ------------------------------------------------------------------ locality exchange (dp_locality)
# Owner-computes with point-to-point traffic only.  Rank r's cameras touch the ascending row list T_r.
#   A  border_plan        : T_r splits into `mine` (own range) and `border` (grouped by owner, ascending);
#                           one all_to_all of counts + one of row ids tells every owner which rows to serve
#   B  border_params_out  : owners send the CURRENT parameter rows of the requested ids (after their deferred
#                           optimizer caught them up); the requester writes them into its replica at the row ids
#   -- render: gradients accumulate by row id as on one GPU (first-touch stores, stamp = this step) --
#   D  border_grads_home  : the requester returns the gradient rows (SH row | packed small row, 240 B) of its
#                           border rows; the owner adds them, requester by requester in rank order (deterministic),
#                           storing instead of adding -- and stamping -- where the row had no gradient this step yet
#   F  publish_small      : (dp_small_owner off) every owner all-gathers (row id, summed packed small gradient) of
#                           ITS rows touched this step, so that the replicated small-attribute Adam (the next
#                           visibility pass needs every row's position on every rank) sees the same global sum
#                           everywhere: 52 B per peer for a row deep inside a rank's region
#   S  small_fetch        : (dp_small_owner, default; at the HEAD of the batch, before the visibility pass) the
#                           current packed small-attribute lines of the foreign rows that may be visible -- the
#                           candidates of clmgs_visibility_candidates -- from their owners, who alone step them;
#                           a row deep inside a rank's region then costs nothing but its share of the periodic
#                           all-gather of the owned ranges (gaussian_model.small_refresh)
# Nothing else travels.
class BorderPlan:
    __slots__ = ("n_ranks", "rank", "lo", "hi", "n_total", "mine", "border", "need", "serve", "serve_rows",
                 "own_rows", "own_counts", "parts")


class _Part:
    """One slice of the border rows for an exchange issued on its own: split sizes per peer + the positions of the
    slice inside `border` (requester side) and inside `serve_rows` (owner side), both grouped by peer."""
    __slots__ = ("need", "serve", "border_idx", "serve_idx")


def _segments(starts, lens, dev):
    """Concatenation of arange(starts[q], starts[q] + lens[q]) over the peers (host lists -> one device tensor)."""
    segs = [torch.arange(a, a + n, device=dev) for a, n in zip(starts, lens) if n]
    return torch.cat(segs) if segs else torch.empty((0,), dtype=torch.int64, device=dev)


def _isin_sorted(values, sorted_set):
    """values[i] in sorted_set (ascending, unique) -> bool, by binary search (no set materialised)."""
    if sorted_set is None or sorted_set.numel() == 0 or values.numel() == 0:
        return torch.zeros(values.shape, dtype=torch.bool, device=values.device)
    pos = torch.searchsorted(sorted_set, values).clamp_(max=sorted_set.numel() - 1)
    return sorted_set[pos] == values


def border_plan(touched_rows, n_total, first_rows=None, last_rows=None, publish_counts=True):
    """touched_rows: ascending int64 ids this rank's cameras touch.  Two small collectives + the count all-gather; two
    host reads (the owner boundaries, before any collective; the split sizes).

    first_rows / last_rows (ascending ids of the FIRST / LAST camera's filter, optional) split the exchange so that
    it can hide behind rendering (engine: clm_offload/engine.py):
      * B is issued in two parts: `params0` = the border rows the first camera renders from, `params1` = the rest --
        camera 0 starts as soon as its own part has landed, the larger rest travels while camera 0 renders;
      * D likewise: `grads0` = the border rows the LAST camera does not touch (their gradient lines are final once the
        second-to-last backward has run: they travel under the last camera's backward), `grads1` = the rest.
    Within every peer's segment the border list is ordered (first-camera rows, then the others), ascending inside each
    group; the in-last flag travels in bit 62 of the ids, so both sides derive the same D lists.  Without first_rows /
    last_rows there is one part each (`params0` / `grads0` empty), the exchange is what it was.
    publish_counts=False (dp_small_owner: nothing is published, step F does not run) skips the all-gather of the
    own-row counts and its host read."""
    G, r = world_size(), rank()
    dev = touched_rows.device
    cuts = torch.tensor([(q * n_total) // G for q in range(G + 1)], dtype=torch.int64).to(dev)
    # host read 1: the owner boundaries inside the sorted touched list (no collective behind it: the caller has just
    # read the filter sizes of the same selection, the device has nothing queued)
    bl = torch.searchsorted(touched_rows, cuts).tolist()
    n_t = int(touched_rows.numel())
    bl[0], bl[-1] = 0, n_t
    need_l = [bl[q + 1] - bl[q] for q in range(G)]
    need_l[r] = 0
    nb_start = [sum(need_l[:q]) for q in range(G)]
    n_b = sum(need_l)
    pl = BorderPlan()
    pl.n_ranks, pl.rank, pl.n_total = G, r, n_total
    pl.lo, pl.hi = owner_range(n_total, r, G)
    pl.mine = touched_rows[bl[r]:bl[r + 1]]
    border = torch.cat((touched_rows[:bl[r]], touched_rows[bl[r + 1]:]))     # ascending = grouped by owner
    i64 = dict(dtype=torch.int64, device=dev)
    starts_d = torch.tensor(nb_start, **i64)
    ends_d = torch.tensor([a + n for a, n in zip(nb_start, need_l)], **i64)
    if n_b:
        # everything below works on the BORDER rows only (none on one rank, a fifth of the touched rows on two) with
        # scans and scatters -- no sort, no atomics
        in_first = _isin_sorted(border, first_rows)
        in_last = _isin_sorted(border, last_rows) if last_rows is not None else torch.ones_like(in_first)
        cf = torch.cat((torch.zeros(1, **i64), torch.cumsum(in_first.to(torch.int64), 0)))      # exclusive counts
        cl = torch.cat((torch.zeros(1, **i64), torch.cumsum((~in_last).to(torch.int64), 0)))
        need0 = cf[ends_d] - cf[starts_d]
        need_d0 = cl[ends_d] - cl[starts_d]
    else:
        in_first = in_last = torch.zeros((0,), dtype=torch.bool, device=dev)
        need0 = need_d0 = torch.zeros((G,), **i64)
    need3 = torch.stack((torch.tensor(need_l, **i64), need0, need_d0), dim=1).contiguous()      # [G,3]
    serve3 = torch.empty_like(need3)
    dist.all_to_all_single(serve3.view(-1), need3.view(-1))
    host = torch.cat((need3.view(-1), serve3.view(-1))).tolist()              # host read 2: the split sizes
    n3, s3 = host[:3 * G], host[3 * G:]
    need0_l, need_d0_l = n3[1::3], n3[2::3]
    serve_l, serve0_l, serve_d0_l = s3[0::3], s3[1::3], s3[2::3]
    if n_b:
        # stable partition of every owner's segment into (first-camera rows, the others): destination by rank
        seg = torch.repeat_interleave(torch.arange(G, device=dev), torch.tensor(need_l, **i64), output_size=n_b)
        pos = torch.arange(n_b, device=dev)
        rank_first = cf[:-1] - cf[starts_d][seg]                 # first-camera rows before me in my segment
        rank_rest = (pos - starts_d[seg]) - rank_first
        dest = starts_d[seg] + torch.where(in_first, rank_first, need0[seg] + rank_rest)
        pl.border = torch.empty_like(border)
        pl.border[dest] = border
        border_last = torch.empty_like(in_last)
        border_last[dest] = in_last
    else:
        pl.border, border_last = border, in_last
    pl.need, pl.serve = need_l, serve_l
    pl.serve_rows = torch.empty((sum(serve_l),), dtype=torch.int64, device=dev)
    tagged = pl.border | (border_last.to(torch.int64) << 62)
    dist.all_to_all_single(pl.serve_rows, tagged, output_split_sizes=serve_l, input_split_sizes=need_l)
    serve_last = ((pl.serve_rows >> 62) & 1).to(torch.bool)
    pl.serve_rows = pl.serve_rows & ((1 << 62) - 1)
    _count("all_to_all_ids", 8 * (3 * G + pl.border.numel()))
    # ---- the parts
    sv_start = [sum(serve_l[:q]) for q in range(G)]
    parts = {}
    p0, p1 = _Part(), _Part()
    p0.need, p0.serve = need0_l, serve0_l
    p1.need, p1.serve = [a - c for a, c in zip(need_l, need0_l)], [a - c for a, c in zip(serve_l, serve0_l)]
    p0.border_idx = _segments(nb_start, need0_l, dev)
    p1.border_idx = _segments([a + c for a, c in zip(nb_start, need0_l)], p1.need, dev)
    p0.serve_idx = _segments(sv_start, serve0_l, dev)
    p1.serve_idx = _segments([a + c for a, c in zip(sv_start, serve0_l)], p1.serve, dev)
    parts["params0"], parts["params1"] = p0, p1
    # D parts: the rows with in_last == 0 (part 0) / == 1 (part 1); ascending positions ARE grouped by peer, and their
    # counts are known on the host (nonzero_static: no readback)
    def by_flag(flags, n_zero):
        n = int(flags.numel())
        if n == 0:
            e = torch.empty((0,), **i64)
            return e, e
        return (torch.nonzero_static(~flags, size=n_zero).flatten(), torch.nonzero_static(flags, size=n - n_zero).flatten())
    g0, g1 = _Part(), _Part()
    g0.need, g0.serve = need_d0_l, serve_d0_l
    g1.need, g1.serve = [a - c for a, c in zip(need_l, need_d0_l)], [a - c for a, c in zip(serve_l, serve_d0_l)]
    g0.border_idx, g1.border_idx = by_flag(border_last, sum(need_d0_l))
    g0.serve_idx, g1.serve_idx = by_flag(serve_last, sum(serve_d0_l))
    parts["grads0"], parts["grads1"] = g0, g1
    pl.parts = parts
    # the rows of this rank's range anybody touches this batch, and how many every rank has: known NOW, so the
    # end-of-batch publication of the small-gradient sums (step F) needs no size readback of its own
    pl.own_rows = border_own_rows(pl)
    pl.own_counts = None
    if publish_counts:
        k = torch.tensor([pl.own_rows.numel()], dtype=torch.int64, device=dev)
        counts = torch.empty((G,), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(counts, k)
        pl.own_counts = counts.tolist()
        _count("all_gather_small", 8 * (G - 1))
    return pl


def border_own_rows(pl):
    """Rows of this rank's range that anybody touches this batch (own cameras or served), ascending."""
    dev = pl.mine.device
    if pl.serve_rows.numel() == 0:
        return pl.mine  # nobody asked for anything (always so on one rank): the own touched rows, already ascending
    mark = torch.zeros((max(1, pl.hi - pl.lo),), dtype=torch.bool, device=dev)
    if pl.mine.numel():
        utils.fill_rows(mark, pl.mine - pl.lo, True)
    if pl.serve_rows.numel():
        utils.fill_rows(mark, pl.serve_rows - pl.lo, True)
    return torch.nonzero(mark).flatten() + pl.lo


def border_params_out(table, pl, part=None):
    """B: table[border] <- the owners' rows.  part: None = all border rows in one exchange; "params0" / "params1" = the
    two slices of border_plan (the first camera's rows, the rest), each an exchange of its own."""
    W = table.shape[1]
    if part is None:
        need, serve, b_rows, s_rows = pl.need, pl.serve, pl.border, pl.serve_rows
    else:
        pt = pl.parts[part]
        need, serve = pt.need, pt.serve
        b_rows = pl.border[pt.border_idx] if pt.border_idx.numel() else pl.border[:0]
        s_rows = pl.serve_rows[pt.serve_idx] if pt.serve_idx.numel() else pl.serve_rows[:0]
    send = _take(table, s_rows) if s_rows.numel() else table.new_empty((0, W))
    recv = table.new_empty((b_rows.numel(), W))
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=need, input_split_sizes=serve)
    _count("all_to_all_params", send.numel() * send.element_size())
    if b_rows.numel():
        _put(table, b_rows, recv)


def border_grads_send(tables, stamp, step, pl, part=None):
    """D, first half: the gradient lines of this rank's border rows (all, or the slice "grads0" / "grads1" of
    border_plan) travel to their owners.  -> the received lines [n_served, sum of widths] (+ the served row ids), to be
    handed to border_grads_apply once nothing on this rank writes the gradient tables any more.  Only READS the
    tables (rows of this slice), so it may run on a side stream under the last camera's backward when the slice holds
    no row that camera touches ("grads0")."""
    widths = [t.shape[1] for t in tables]
    Wt = sum(widths)
    t0 = tables[0]
    if part is None:
        need, serve, b_rows, s_rows = pl.need, pl.serve, pl.border, pl.serve_rows
    else:
        pt = pl.parts[part]
        need, serve = pt.need, pt.serve
        b_rows = pl.border[pt.border_idx] if pt.border_idx.numel() else pl.border[:0]
        s_rows = pl.serve_rows[pt.serve_idx] if pt.serve_idx.numel() else pl.serve_rows[:0]
    if b_rows.numel():
        send = torch.cat([_take(t, b_rows) for t in tables], dim=1)
        if stamp is not None:
            # The plan lists every border row of the FILTERS; the backward stores (and stamps) only rows that were drawn
            # (radius > 0).  A filtered row the exact projection did not draw (fast-accept vs exact tie, a camera redone
            # over capacity) keeps the gradient lines of an EARLIER step under first-touch stores: it travels as zeros,
            # never as that stale content (the owner then adds / stores nothing for it).
            live = (utils.take_rows(stamp, b_rows) == step)[:, None]
            send = torch.where(live, send, torch.zeros((), dtype=send.dtype, device=send.device))
    else:
        send = t0.new_empty((0, Wt))
    recv = t0.new_empty((s_rows.numel(), Wt))
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=serve, input_split_sizes=need)
    _count("all_to_all_grads", send.numel() * send.element_size())
    return recv, s_rows, serve, b_rows


def border_grads_apply(tables, stamp, step, pl, received):
    """D, second half (owner side): add the received lines, requester by requester in rank order (ids are unique
    within a requester's segment -> no atomics, a fixed summation order), storing instead of adding -- and stamping
    -- where the row had no gradient this step yet (first-touch policy; stamp None: clearing policy, plain add)."""
    recv, s_rows, serve, b_rows = received
    widths = [t.shape[1] for t in tables]
    off = 0
    for q in range(pl.n_ranks):
        k = serve[q]
        if not k:
            continue
        ids, seg = s_rows[off:off + k], recv[off:off + k]
        c = 0
        if stamp is None:
            for t, w in zip(tables, widths):
                _put(t, ids, _take(t, ids) + seg[:, c:c + w])
                c += w
        else:
            fresh = (utils.take_rows(stamp, ids) != step)[:, None]  # no gradient at the owner yet this step: store
            for t, w in zip(tables, widths):
                _put(t, ids, torch.where(fresh, seg[:, c:c + w], _take(t, ids) + seg[:, c:c + w]))
                c += w
            utils.fill_rows(stamp, ids, step)
        off += k
    if stamp is None and b_rows.numel():
        for t in tables:
            utils.fill_rows(t, b_rows, 0.0)


def border_grads_home(tables, stamp, step, pl):
    """D in one go (send + apply of all border rows): the gradient rows of this rank's border rows go to their owners,
    which accumulate them.  tables: gradient tables [N, w_i] sharing `stamp` (int32 [N]: the step a row's gradient
    lines belong to; first-touch policy).  stamp None: clearing policy (rows without a gradient hold zeros) -- the
    owner simply adds, and the sender's border rows are zeroed once handed over."""
    border_grads_apply(tables, stamp, step, pl, border_grads_send(tables, stamp, step, pl))


def publish_rows(tables, own_rows, n_total, counts=None, live=None, stamp=None, step=0):
    """F, general form: every owner all-gathers (row id, its summed rows of `tables`, side by side) for `own_rows`
    (ascending absolute ids inside its range); the receivers store the rows.  `counts`: len(own_rows) of every rank
    if the caller exchanged them already (border_plan does) -- no host read here then.
    `stamp` + `step` (or `live`, a bool per own row): rows whose line is not of this step are sent as zeros (the sizes
    stay the plan's).
    -> (counts per rank, list of the absolute id tensors received from every other rank)."""
    G, r = world_size(), rank()
    lo, _ = owner_range(n_total, r, G)
    t0 = tables[0]
    widths = [t.shape[1] if t.dim() > 1 else 1 for t in tables]
    W = sum(widths)
    if counts is None:
        k = torch.tensor([own_rows.numel()], dtype=torch.int64, device=t0.device)
        counts = torch.empty((G,), dtype=torch.int64, device=t0.device)
        dist.all_gather_into_tensor(counts, k)
        counts = counts.tolist()  # host read
        _count("all_gather_small", 8 * (G - 1))
    chunk = max(1, max(counts))
    # one message per rank: [chunk, W] summed rows, then chunk row ids (int32, relative to the owner's range, carried
    # as raw bits in a float lane) -- two contiguous blocks, so the pack is one gather per table and one copy
    send = t0.new_empty((chunk * (W + 1),))
    rows_blk, ids_blk = send[:chunk * W].view(chunk, W), send[chunk * W:]
    n_own = own_rows.numel()
    t_ = tables[0]
    if n_own and len(tables) == 1 and t_.dim() == 2 and t_.is_cuda and t_.dtype == torch.float32 and W % 4 == 0 \
            and t_.is_contiguous() and live is None:
        # one library pass: rows (zeros for lines that are not of `step`, if a stamp table is given) + the id block
        from . import _lib
        from ._lib import check, dptr, stream
        check(_lib.lib().clmgs_publish_pack(stream(), dptr(send), dptr(t_), dptr(own_rows.contiguous(), torch.int64),
                                            dptr(stamp, torch.int32, True), int(step), int(lo), int(n_own), int(chunk), int(W)))
    elif n_own:
        if stamp is not None and live is None:
            live = utils.take_rows(stamp, own_rows) == step
        if len(tables) == 1 and tables[0].dim() == 2:
            _take_into(rows_blk[:n_own], tables[0], own_rows)
        else:
            c = 0
            for t, w in zip(tables, widths):
                rows_blk[:n_own, c:c + w] = utils.take_rows(t.reshape(t.shape[0], -1), own_rows)
                c += w
        if live is not None:  # bool[n_own]: rows whose line does not belong to this step travel as zeros
            rows_blk[:n_own].masked_fill_(~live[:, None], 0.0)
        ids_blk[:n_own] = (own_rows - lo).to(torch.int32).view(torch.float32)
    recv = t0.new_empty((G * chunk * (W + 1),))
    dist.all_gather_into_tensor(recv, send)
    _count("all_gather_small", send.numel() * send.element_size() * (G - 1))
    got = []
    for q in range(G):
        if q == r or not counts[q]:
            continue
        blk = recv[q * chunk * (W + 1):(q + 1) * chunk * (W + 1)]
        seg = blk[:chunk * W].view(chunk, W)[:counts[q]]
        ids = blk[chunk * W:chunk * W + counts[q]].contiguous().view(torch.int32).to(torch.int64) + owner_range(n_total, q, G)[0]
        c = 0
        for t, w in zip(tables, widths):
            if len(tables) == 1 and t.dim() == 2:
                _put(t, ids, seg)
            else:
                utils.put_rows(t.reshape(t.shape[0], -1), ids, seg[:, c:c + w])
            c += w
        got.append(ids)
    return counts, got


def publish_small(small_g, stamp, step, n_total, pl=None):
    """F (first-touch policy): the rows of this rank's range anybody touched (all stamped `step` by now: by this
    rank's backward kernels or by border_grads_home) are published with their summed packed small-gradient row, the
    receivers store and stamp them, so the replicated small-attribute Adam consumes identical sums everywhere.
    With the batch's BorderPlan the row list and every rank's count are at hand: no scan, no host read."""
    if pl is not None:
        # own rows no camera drew after all (see border_grads_home) hold an earlier step's line: published as zeros
        # -- the sizes of the exchange stay the plan's (known before rendering, no readback), its CONTENT follows the
        # stamps; a zero line stamped `step` is what the replicated small-attribute Adam reads for an unstamped row
        counts, got = publish_rows([small_g], pl.own_rows, n_total, counts=pl.own_counts, stamp=stamp, step=step)
    else:
        lo, hi = owner_range(n_total)
        own = torch.nonzero(stamp[lo:hi] == step).flatten() + lo
        counts, got = publish_rows([small_g], own, n_total)
    for ids in got:
        utils.fill_rows(stamp, ids, step)
    return counts


def small_fetch(rows, n_total, packed):
    """S (dp_small_owner): the CURRENT packed small-attribute lines ([., 12]: xyz 3 | opacity 1 | scaling 3 |
    rotation 4 | pad) of the rows `rows` (ascending int64 ids outside this rank's range: the candidates of this
    batch's visibility pass) from their owners, whose mirror `packed` is current for their own range.
    -> recv [len(rows), 12] in the order of `rows`.  Two host reads (owner boundaries, split sizes), three
    all_to_alls (counts, ids, lines).  Every rank calls it every batch, also with nothing to ask for."""
    G, r = world_size(), rank()
    dev = rows.device
    cuts = torch.tensor([(q * n_total) // G for q in range(G + 1)], dtype=torch.int64).to(dev)
    bl = torch.searchsorted(rows, cuts).tolist()
    n_r = int(rows.numel())
    bl[0], bl[-1] = 0, n_r
    need = [bl[q + 1] - bl[q] for q in range(G)]
    assert need[r] == 0, "small_fetch: own rows are current, only foreign rows are fetched"
    need_t = torch.tensor(need, dtype=torch.int64).to(dev)
    serve_t = torch.empty_like(need_t)
    dist.all_to_all_single(serve_t, need_t)
    serve = serve_t.tolist()
    serve_rows = torch.empty((sum(serve),), dtype=torch.int64, device=dev)
    dist.all_to_all_single(serve_rows, rows.contiguous(), output_split_sizes=serve, input_split_sizes=need)
    send = _take(packed, serve_rows) if serve_rows.numel() else packed.new_empty((0, packed.shape[1]))
    recv = packed.new_empty((n_r, packed.shape[1]))
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=need, input_split_sizes=serve)
    _count("all_to_all_small_ids", 8 * (G + n_r))
    _count("all_to_all_small", send.numel() * send.element_size())
    return recv


def small_scatter(rows, lines, packed, tensors):
    """rows' packed lines -> the mirror and the four parameter tensors (xyz, opacity, scaling, rotation)."""
    if rows.numel() == 0:
        return
    xyz, opa, sca, rot = tensors
    if packed.is_cuda:
        from . import _lib
        _lib.check(_lib.lib().clmgs_small_rows_scatter(
            _lib.stream(), int(rows.numel()), _lib.dptr(rows.contiguous(), torch.int64), _lib.dptr(lines.contiguous()),
            _lib.dptr(xyz), _lib.dptr(opa), _lib.dptr(sca), _lib.dptr(rot), _lib.dptr(packed)))
        return
    packed[rows] = lines  # CPU tensors of the gloo tests
    xyz[rows], opa[rows], sca[rows], rot[rows] = lines[:, 0:3], lines[:, 3:4], lines[:, 4:7], lines[:, 7:11]


def camera_shares(filters, n_total, n_ranks):
    """[len(filters), n_ranks] int64 (host): how many of each camera's rows lie in each rank's range."""
    dev = filters[0].device
    cuts = torch.tensor([(q * n_total) // n_ranks for q in range(n_ranks + 1)], dtype=torch.int64).to(dev)
    b = torch.stack([torch.searchsorted(f, cuts) for f in filters])
    return (b[:, 1:] - b[:, :-1]).cpu()


def assign_cameras(shares, per_rank=None):
    """Deal cameras to ranks by locality: camera c prefers the rank owning most of its rows (shares[c, q]);
    every rank gets at most `per_rank` (default ceil(len / ranks)) and at least floor(len / ranks) cameras: cameras are
    taken in order of how much they lose by not getting their first choice, each goes to its best rank with room left.
    -> list of rank per camera (deterministic: every rank computes the same deal)."""
    shares = torch.as_tensor(shares).to(torch.float64)
    n, G = shares.shape
    cap = per_rank if per_rank is not None else (n + G - 1) // G
    top2 = torch.topk(shares, k=min(2, G), dim=1).values
    regret = (top2[:, 0] - (top2[:, 1] if G > 1 else 0)).tolist()
    order = sorted(range(n), key=lambda c: (-regret[c], c))
    room = [cap] * G
    # every rank is guaranteed floor(n / G) cameras (a cap alone lets the last ranks run short or empty: 30 cameras over
    # 8 ranks dealt 4,4,4,4,4,4,4,2): while the cameras left only just cover the outstanding minimums, a camera may
    # only go to a rank that is still below its minimum
    owed = [min(cap, n // G)] * G
    out = [0] * n
    pref = torch.argsort(shares, dim=1, descending=True, stable=True).tolist()
    left = n
    for c in order:
        must = left <= sum(owed)
        for q in pref[c]:
            if room[q] > 0 and (not must or owed[q] > 0):
                out[c] = q
                room[q] -= 1
                owed[q] = max(0, owed[q] - 1)
                break
        left -= 1
    return out


def deal_cameras(cameras, gaussians, n_ranks=None, chunk=8, per_rank=None):
    """Locality deal of a camera list over the ranks for the CURRENT row order of `gaussians` (rows in Z-order:
    utils.morton_order / model.spatial_sort): visibility of every camera (GPU selection, `chunk` cameras per pass)
    -> camera_shares -> assign_cameras.  Deterministic; every rank computes the same deal.
    -> (rank of every camera, shares [n_cameras, n_ranks] on the host)."""
    from .strategies.base_engine import select_filters
    n_ranks = world_size() if n_ranks is None else n_ranks
    n = gaussians._xyz.shape[0]
    rows = []
    with torch.no_grad():
        for a in range(0, len(cameras), chunk):
            filters, _ = select_filters(cameras[a:a + chunk], gaussians._xyz.detach(), gaussians._scaling.detach(),
                                        gaussians._rotation.detach())
            rows.append(camera_shares(filters, n, n_ranks))
    shares = torch.cat(rows)
    return assign_cameras(shares, per_rank), shares


def exchange_bytes(touched_per_rank, n_total, small_refresh=8):
    """Wire bytes per rank and batch of the three exchanges for given per-rank touched sets (ascending int64 id
    tensors, one per rank) -- pure index arithmetic, no communication: used to account a partition of the bench
    scenes into virtual ranks.  -> {"allreduce": [...], "owner": [...], "locality": [...], "union": U, ...}
    "locality" is the exchange with step F (dp_small_owner off); "locality_small_owner" the default: no F, step S
    instead (ids + 48 B lines of the candidates -- counted as the border rows, which the candidates exceed by the rim
    the drift margins add) and the all-gather of the owned small-attribute ranges every `small_refresh` batches."""
    G = len(touched_per_rank)
    dev = touched_per_rank[0].device
    cuts = torch.tensor([(q * n_total) // G for q in range(G + 1)], dtype=torch.int64).to(dev)
    mark = torch.zeros((n_total,), dtype=torch.bool, device=dev)
    for t in touched_per_rank:
        utils.fill_rows(mark, t, True)
    U = int(mark.sum())
    cs = torch.cumsum(mark.to(torch.int64), 0)
    ends = torch.cat((torch.zeros(1, dtype=torch.int64, device=dev), cs[cuts[1:] - 1])).tolist()
    own_touched = [ends[q + 1] - ends[q] for q in range(G)]           # |U in range q|
    per = torch.stack([torch.searchsorted(t, cuts) for t in touched_per_rank])
    per = (per[:, 1:] - per[:, :-1]).tolist()                          # per[p][q] = |T_p in range q|
    f = 2.0 * (G - 1) / G
    allreduce = [f * (240.0 * U + n_total)] * G                        # packed rows + the uint8 touched mask
    chunk = max(own_touched)
    owner = [(G - 1) * 192.0 * chunk * 2 + f * (48.0 * U + n_total)] * G
    locality, locality_so = [], []
    own_max = max(((q + 1) * n_total) // G - (q * n_total) // G for q in range(G))
    for r in range(G):
        border = sum(per[r][q] for q in range(G) if q != r)
        serve = sum(per[p][r] for p in range(G) if p != r)
        core = 8.0 * (3 * G + border) + 192.0 * serve + 240.0 * border + 8.0 * (G - 1)
        locality.append(core + (G - 1) * 52.0 * chunk)
        locality_so.append(core - 8.0 * (G - 1) + 8.0 * (G + border) + 48.0 * serve + (G - 1) * 44.0 * own_max / float(small_refresh))
    return {"allreduce": allreduce, "owner": owner, "locality": locality, "locality_small_owner": locality_so,
            "union": U, "n_ranks": G,
            "touched": [int(t.numel()) for t in touched_per_rank],
            "border": [sum(per[r][q] for q in range(G) if q != r) for r in range(G)],
            "own_touched": own_touched, "reference_240B_x_union": 240.0 * U}


def allreduce_densify_stats(gaussians):
    if not active():
        return
    dist.all_reduce(gaussians.xyz_gradient_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(gaussians.denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(gaussians.max_radii2D, op=dist.ReduceOp.MAX)


def seed_split_generator(gaussians, seed=1234):
    """densify_and_split draws torch.normal: replicas must draw the same numbers."""
    dev = gaussians._xyz.device
    gaussians.split_generator = torch.Generator(device=dev).manual_seed(seed)
