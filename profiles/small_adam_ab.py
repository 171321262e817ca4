"""A/B of the packed small-attribute Adam kernel: this build's clmgs_adam_small_packed against another build's (path of
its libclmgs_hip.so as argv[1]) in ONE process on one box, 28 M rows, first-touch stamps on 38 % of the rows."""
import ctypes
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from clm_gs_amd import _lib

n = 28_000_000
other = ctypes.CDLL(sys.argv[1]) if len(sys.argv) > 1 else None
g = torch.Generator(device="cuda").manual_seed(0)
widths = (3, 1, 3, 4)
ps = [torch.randn(n, w, device="cuda", generator=g) for w in widths]
ms = [torch.zeros(n, w, device="cuda") for w in widths]
vs = [torch.zeros(n, w, device="cuda") for w in widths]
pk = torch.zeros(n, 12, device="cuda")
gk = torch.randn(n, 12, device="cuda", generator=g)
stamp = torch.where(torch.rand(n, device="cuda", generator=g) < 0.38, 7, 3).to(torch.int32)
arr = lambda xs: (ctypes.c_void_p * 4)(*[x.data_ptr() for x in xs])
lrs = (ctypes.c_double * 4)(1e-4, 5e-2, 5e-3, 1e-3)
VP, I64, D, I, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_float


def call(lib, step):
    f = lib.clmgs_adam_small_packed
    f.restype = I
    f.argtypes = [VP, I64, VP, VP, VP, VP, VP, VP, D, D, D, I, I, F, VP, I]
    rc = f(_lib.stream(), n, arr(ps), arr(ms), arr(vs), lrs, pk.data_ptr(), gk.data_ptr(), 0.9, 0.999, 1e-15, step, 1,
           0.25, stamp.data_ptr(), 7)
    assert rc == 0


def timed(lib, reps=20):
    for s in range(3):
        call(lib, 1 + s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(reps):
        call(lib, 4 + s)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e3, 4)


res = {}
for rnd in range(3):
    res.setdefault("this_build_ms", []).append(timed(ctypes.CDLL(_lib.LIB_PATH)))
    if other is not None:
        res.setdefault("other_build_ms", []).append(timed(other))
print("SMALLADAM " + json.dumps(res))
