"""Independent-camera data parallelism (net-new: the reference is single GPU).

One process per GPU, every rank holds a full replica and renders its own cameras of the global
batch.  ONE exchange per batch, before the optimizer: sum of the dense [N,11] small gradients,
OR of the touched-row mask, and sum of the SH gradient rows restricted to the union of touched
rows.  Densification statistics accumulate locally and are reduced only right before
densify_and_prune (sum / sum / max).  Everything is plain torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; the CPU tests use gloo), device-agnostic tensors.
"""
import torch
import torch.distributed as dist


import os


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
    buf = grad_rows[rows]
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if average:
        buf /= ws
    grad_rows[rows] = buf


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
    buf = torch.cat([t[rows] for t in tables], dim=1)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if average:
        buf /= ws
    for t, piece in zip(tables, torch.split(buf, widths, dim=1)):
        t[rows] = piece


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
