"""The deferred SH-row optimizer pass alone (python profiles/catch_up_microbench.py; CLMGS_LIB_PATH selects the library):
28 M rows of [N,48] p / m / v / g tables, 10.7 M touched rows in ~2 000 contiguous runs (what a batch of four nadir
cameras touches in Z-ordered tables), a third of them with a gradient line waiting -- clmgs_adam_catch_up called
directly (the stamps are not advanced, so every call does the same work); int32 and int64 row lists."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clm_gs_amd import _lib
from clm_gs_amd._lib import dptr

N, T = 28_000_000, 10_700_000
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
p = torch.randn((N, 48), device=dev, generator=g)
m = torch.randn((N, 48), device=dev, generator=g) * 1e-3
v = torch.rand((N, 48), device=dev, generator=g) * 1e-6
gr = torch.randn((N, 48), device=dev, generator=g) * 1e-3
runs = 2000
starts = torch.sort(torch.randint(0, N - 2 * T // runs, (runs,), device=dev, generator=g)).values
lens = torch.randint(T // runs // 2, 3 * T // runs // 2, (runs,), device=dev, generator=g)
mask = torch.zeros((N + 1,), dtype=torch.int32, device=dev)
mask.index_add_(0, starts, torch.ones_like(starts, dtype=torch.int32))
mask.index_add_(0, torch.clamp(starts + lens, max=N), -torch.ones_like(starts, dtype=torch.int32))
rows64 = torch.nonzero(torch.cumsum(mask[:N], 0) > 0).flatten()
rows32 = rows64.to(torch.int32)
last = torch.zeros((N,), dtype=torch.int32, device=dev)
g_step = (torch.rand((N,), device=dev, generator=g) < 0.33).to(torch.int32) * 3   # a gradient of step 3 waits on a third
col_lr = torch.full((48,), 1e-3, device=dev)
L = _lib.lib()
out = {"lib": os.environ.get("CLMGS_LIB_PATH", "default"), "rows": int(rows64.numel())}
for name, rows, is64 in (("int32", rows32, 0), ("int64", rows64, 1)):
    def call():
        _lib.check(L.clmgs_adam_catch_up(_lib.stream(), dptr(p), dptr(m), dptr(v), dptr(last), dptr(rows, None), is64,
                                         int(rows.numel()), 48, dptr(col_lr), 0.9, 0.999, 1e-15, 4, 1, 256, dptr(gr),
                                         dptr(g_step), 0.25, 1))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    out[name + "_ms_min_med"] = [round(ts[0], 4), round(ts[5], 4)]
# a list whose rows are all current (a later camera's duplicates): stamps only
last.fill_(4)
for _ in range(2):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
out["all_rows_current_ms"] = round(e0.elapsed_time(e1), 4)
print(json.dumps(out))
