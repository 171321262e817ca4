"""Throughput of the deferred host row optimizer (clmgs_host_rows_prepare) on the GPU box's host:
pool size sweep, pinned (NUMA-interleaved or not) vs pageable tables.  One JSON line per setting.
usage: python profiles/probe_host_prepare.py [pinned|pageable]"""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clm_gs_amd import _lib  # noqa: E402
from clm_gs_amd.host import pinned_empty  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "pinned"
N, V = 12_000_000, 3_300_000
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
alloc = (lambda shape, dt=torch.float32: pinned_empty(shape, dtype=dt)) if kind == "pinned" else (lambda shape, dt=torch.float32: torch.empty(shape, dtype=dt))
L.clmgs_host_pool_start(64)
t0 = time.perf_counter()
p, g, m, v = alloc((N, 48)), alloc((N, 48)), alloc((N, 48)), alloc((N, 48))
stage = alloc((V, 48))
last, gs = alloc((N,), torch.int32), alloc((N,), torch.int32)
t_alloc = time.perf_counter() - t0
for t in (p, g, m, v):
    t.fill_(1e-3)
last.zero_()
gs.zero_()
rows = torch.randperm(N)[:V].sort().values.to(torch.int32).contiguous()
col_lr = torch.full((48,), 1e-3)
print(json.dumps({"kind": kind, "interleave": os.environ.get("CLMGS_PINNED_NO_INTERLEAVE") is None, "alloc_s": round(t_alloc, 2)}))
step = 0
for nt in (8, 16):
    L.clmgs_host_pool_start(nt)
    best = 1e9
    for rep in range(3):
        step += 2
        gs[rows.long()] = step - 1   # a gradient waits at step-1, one zero-gradient replay after it
        t0 = time.perf_counter()
        _lib.check(L.clmgs_host_rows_prepare(P(p), P(g), P(m), P(v), P(last), P(gs), P(rows), V, 48, P(col_lr), 0.9, 0.999,
                                             1e-15, step, 0, 1, 0.25, 256, P(stage), 0))
        best = min(best, time.perf_counter() - t0)
    print(json.dumps({"threads": nt, "ms": round(best * 1e3, 1), "Mrows_per_s": round(V / best / 1e6, 1),
                      "GBps": round(V * (768 + 576 + 192) / best / 1e9, 1)}))
