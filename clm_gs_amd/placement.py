"""Row tables placed by measurement (round 6).

The four [capacity, 48] row tables (parameters, gradients, two moments) are read and written BY ROW INDEX every batch
(`clmgs_adam_catch_up`, `clmgs_preprocess_fwd/bwd`): 192 B rows, ascending gappy lists.  How fast such a gather runs depends
on the PHYSICAL placement the driver hands out for an allocation -- identical virtual addresses, clocks and power measured
4.3 ... 5.4 TB/s between processes, and between three allocate / free / allocate rounds inside one process
(profiles/gather_probe.hip, `alloc_free_realloc`) -- and nothing in user space can ask for a placement.
What user space can do is LOOK: allocate a candidate, time a gather of a synthetic ascending run list from it, keep the
fastest of a few candidates (the losers are held until the choice is made, so the driver cannot hand their pages back, then
released to it).  A few hipMallocs and ~10 ms of probing per table at set-up time, nothing per batch.

`probe_log()` returns what was measured (bench.py puts it into `measured.alloc`).
"""
import torch

_LOG = []
_MIN_BYTES = 1 << 30      # smaller tables are not worth a probe
_PROBE_ROWS = 4_000_000   # rows gathered per probe (768 MB read + 768 MB written)
_RUNS = 2000


def probe_log():
    return list(_LOG)


def _probe_index(capacity, dev):
    """~_PROBE_ROWS rows of [0, capacity) as ascending contiguous runs spread over the whole table (what a batch of nadir
    cameras over Z-ordered rows touches)."""
    n = int(min(_PROBE_ROWS, max(1, capacity // 4)))
    run = max(1, n // _RUNS)
    runs = max(1, n // run)
    starts = (torch.arange(runs, device=dev, dtype=torch.int64) * ((capacity - run) // max(1, runs - 1) if runs > 1 else 0))
    idx = (starts[:, None] + torch.arange(run, device=dev, dtype=torch.int64)[None, :]).reshape(-1)
    return idx.clamp_(max=capacity - 1).contiguous()


def gather_rate(table, idx, scratch, reps=5, in_place=False):
    """GB/s at which `table`'s rows idx[] are read by the library's row mover (min of `reps` launches, HIP events).
    in_place: every listed row is read and written back where it is (the deferred row pass's pattern: read-modify-write of
    gathered rows); the figure is then rows x row bytes x 2 over the time."""
    from . import clm_kernels
    times = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if in_place:
            clm_kernels._rows("clmgs_rows_gather", table, table, idx, idx, 0)
        else:
            clm_kernels._rows("clmgs_rows_gather", scratch, table, None, idx, 0)
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
    best = min(times[1:])  # (the first launch is the warm-up)
    return idx.numel() * table.shape[1] * 4 * (2 if in_place else 1) / (best * 1e-3) / 1e9


def alloc_rows_placed(capacity, cols=48, candidates=3, name=None):
    """torch.empty((capacity, cols)) on the GPU, the best-placed of `candidates` allocations by the gather probe."""
    dev = torch.device("cuda")
    nbytes = int(capacity) * cols * 4
    if candidates <= 1 or nbytes < _MIN_BYTES:
        return torch.empty((capacity, cols), dtype=torch.float32, device=dev)
    idx = _probe_index(int(capacity), dev)
    scratch = torch.empty((idx.numel(), cols), dtype=torch.float32, device=dev)
    best, best_r, held, rates = None, -1.0, [], []
    for _ in range(int(candidates)):
        try:
            t = torch.empty((capacity, cols), dtype=torch.float32, device=dev)
        except torch.cuda.OutOfMemoryError:  # no room for another candidate: keep what we have
            if best is None:
                raise
            break
        r = gather_rate(t, idx, scratch)
        rates.append(round(r, 1))
        if r > best_r:
            if best is not None:
                held.append(best)
            best, best_r = t, r
        else:
            held.append(t)
    _LOG.append({"table": name, "bytes": nbytes, "candidate_gather_GBps": rates, "chosen": rates.index(round(best_r, 1))})
    del held, scratch, idx, t
    torch.cuda.empty_cache()  # the losers go back to the driver, not into the caching allocator's pool
    return best
