"""Independent-camera data parallelism (net-new: the reference is single GPU).

One process per GPU, every rank holds a full replica and renders its own cameras of the global
batch.  ONE exchange per batch, before the optimizer: sum of the dense [N,11] small gradients,
OR of the touched-row mask, and sum of the SH gradient rows restricted to the union of touched
rows.  Densification statistics accumulate locally and are reduced only right before
densify_and_prune (sum / sum / max).  Everything is plain torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; the CPU tests use gloo), device-agnostic tensors.
"""
import os

import torch
import torch.distributed as dist

from . import utils


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
    if average:
        for g in grads:
            g /= ws


def allreduce_touched(touched):
    """bool[N] -> OR over ranks (as MAX over uint8)."""
    if not active():
        return touched
    t = touched.to(torch.uint8)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
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
        if average:
            grad_rows /= ws
        return
    rows = rows.long()
    buf = utils.take_rows(grad_rows, rows)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if average:
        buf /= ws
    utils.put_rows(grad_rows, rows, buf)


def allreduce_tables_rows(tables, rows, n_total, average=False, dense_above=0.85):
    """Sum (average) several row tables [N, d_i] over ranks, moving ONLY `rows` (the globally
    touched set, identical on every rank): the rows of all tables are packed side by side into one
    [n, sum d_i] message -> ONE collective per batch (236+ B per touched Gaussian) -> unpacked.
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
        if average:
            for t in tables:
                t /= ws
        return
    rows = rows.long()
    widths = [t.shape[1] for t in tables]
    buf = torch.cat([utils.take_rows(t, rows) for t in tables], dim=1)  # chunked: see utils.gather_rows
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if average:
        buf /= ws
    for t, piece in zip(tables, torch.split(buf, widths, dim=1)):
        utils.put_rows(t, rows, piece)


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
        send[: pl.hi - pl.lo] = utils.take_rows(table, pl.rows[pl.lo:pl.hi])
    recv = torch.empty((pl.n_ranks * pl.chunk, W), dtype=table.dtype, device=table.device)
    dist.all_gather_into_tensor(recv, send)
    _count("all_gather", send.numel() * send.element_size() * (pl.n_ranks - 1))
    utils.put_rows(table, pl.rows, utils.take_rows(recv, pl.pos))


def owner_reduce_rows(table, pl):
    """Sum of table[pl.rows] over the ranks, delivered to each row's OWNER; on the other ranks the rows
    are zeroed (the gradient has been handed over)."""
    if not active():
        return
    W = table.shape[1]
    buf = torch.zeros((pl.n_ranks * pl.chunk, W), dtype=table.dtype, device=table.device)
    utils.put_rows(buf, pl.pos, utils.take_rows(table, pl.rows))
    mine = torch.empty((pl.chunk, W), dtype=table.dtype, device=table.device)
    dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM)
    _count("reduce_scatter", mine.numel() * mine.element_size() * (pl.n_ranks - 1))
    utils.fill_rows(table, pl.rows, 0.0)
    if pl.hi > pl.lo:
        utils.put_rows(table, pl.rows[pl.lo:pl.hi], mine[: pl.hi - pl.lo])


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
        for q in range(G):
            a, b = owner_range(n_total, q, G)
            if q != rank():
                t2[a:b] = recv[q * chunk: q * chunk + (b - a)]


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
