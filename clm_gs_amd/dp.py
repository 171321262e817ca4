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


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_small_grads(grads):
    """grads: list of [N,d] tensors -> averaged over ranks in one collective."""
    ws = world_size()
    if ws == 1:
        return
    widths = [g.shape[1] for g in grads]
    flat = torch.cat(grads, dim=1)  # [N, 11]: one 44 B/Gaussian message instead of four
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= ws
    for g, piece in zip(grads, torch.split(flat, widths, dim=1)):
        g.copy_(piece)


def allreduce_touched(touched):
    """bool[N] -> OR over ranks (as MAX over uint8)."""
    if world_size() == 1:
        return touched
    t = touched.to(torch.uint8)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.to(torch.bool)


def allreduce_rows(grad_rows, touched_global):
    """Average grad_rows[N,48] over ranks, moving only rows in the (global) touched set."""
    ws = world_size()
    if ws == 1:
        return
    n_touched = int(touched_global.sum())
    if n_touched == 0:
        return
    if 2 * n_touched >= grad_rows.shape[0]:
        # most rows are in play: reduce the whole buffer in place, no pack / unpack copies
        dist.all_reduce(grad_rows, op=dist.ReduceOp.SUM)
        grad_rows /= ws
        return
    rows = torch.nonzero(touched_global).flatten()
    buf = grad_rows[rows]
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    buf /= ws
    grad_rows[rows] = buf


def allreduce_densify_stats(gaussians):
    if world_size() == 1:
        return
    dist.all_reduce(gaussians.xyz_gradient_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(gaussians.denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(gaussians.max_radii2D, op=dist.ReduceOp.MAX)


def seed_split_generator(gaussians, seed=1234):
    """densify_and_split draws torch.normal: replicas must draw the same numbers."""
    dev = gaussians._xyz.device
    gaussians.split_generator = torch.Generator(device=dev).manual_seed(seed)
