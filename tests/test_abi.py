"""CPU-only: the C-ABI library loads and exports every symbol include/clmgs.h declares, the
ctypes table matches the header, and the product refuses to run without a GPU / library."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "clmgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clmgs_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_table_agree():
    from clm_gs_amd import _lib
    assert _header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_symbol():
    from clm_gs_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    l = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(l, name), name
    l.clmgs_version.restype = ctypes.c_int
    assert l.clmgs_version() >= 100


def test_no_cpu_fallback():
    """Operators must raise on CPU tensors instead of silently computing on the host."""
    from clm_gs_amd import _lib, gsplat
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        gsplat.spherical_harmonics(0, torch.zeros(1, 4, 3), torch.zeros(1, 4, 16, 3))


def test_product_never_imports_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "clm_gs_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dp, f)


def test_tsp_and_host_adam_run_on_cpu():
    """Host-side entry points of the library (no device needed)."""
    from clm_gs_amd import _lib, fast_tsp
    from oracle import gs_oracle as O
    dist = [[0, 5, 9, 1], [5, 0, 2, 8], [9, 2, 0, 7], [1, 8, 7, 0]]
    tour = fast_tsp.find_tour(dist)
    assert sorted(tour) == [0, 1, 2, 3]
    cost = sum(dist[tour[i]][tour[i + 1]] for i in range(3))
    assert cost == 8  # 3-0-1-2 or reverse is optimal (1+5+2)
    g = torch.Generator().manual_seed(0)
    N, cols = 5000, 48
    p, gr = torch.randn(N, cols, generator=g), torch.randn(N, cols, generator=g)
    m, v = torch.rand(N, cols, generator=g) * 0.1, torch.rand(N, cols, generator=g) * 0.01
    rows = torch.randperm(N, generator=g)[:2500].to(torch.int32).contiguous()
    col_lr = torch.cat([torch.full((3,), 2.5e-3), torch.full((45,), 1.25e-4)]).contiguous()
    ref = [t.clone().double() for t in (p, gr, m, v)]
    O.adam_rows(*ref, rows, col_lr.double(), 0.9 ** 4, 0.999 ** 4, 5e-16, step=3, scale=0.25, zero_grad=True)
    sig = torch.ones(1, dtype=torch.int32)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().clmgs_host_adam_rows(P(p), P(gr), P(m), P(v), P(rows), rows.numel(), cols, P(col_lr),
                                               0.9 ** 4, 0.999 ** 4, 5e-16, 3, 1, 0.25, 1, P(sig), 2))
    for a, b in zip((p, gr, m, v), ref):
        assert ((a.double() - b).norm() / (b.norm() + 1e-30)) < 1e-6


@pytest.mark.parametrize("sparse", [0, 1])
def test_deferred_host_row_optimizer_equals_step_by_step_adam(sparse):
    """clmgs_host_rows_prepare (host-resident mode): rows at different staleness, some with a gradient
    waiting at its own step, brought up to date in ONE pass == the eager optimizer applied step by step
    (float64 oracle: zero-gradient steps for rows without a gradient; with sparse Adam such steps do
    not exist); stamps advance; the staged copy is the up-to-date row."""
    from clm_gs_amd import _lib
    from oracle import gs_oracle as O
    L = _lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    g0 = torch.Generator().manual_seed(0)
    N, cols, to_step = 6000, 48, 7
    p, g = torch.randn(N, cols, generator=g0), torch.randn(N, cols, generator=g0)
    m, v = torch.rand(N, cols, generator=g0) * 0.1, torch.rand(N, cols, generator=g0) * 0.01
    m[:100] = 0
    v[:100] = 0
    last = torch.randint(0, 4, (N,), generator=g0, dtype=torch.int32)
    gstep = torch.zeros(N, dtype=torch.int32)
    pend = torch.rand(N, generator=g0) < 0.5
    gstep[pend] = last[pend] + 1 + torch.randint(0, 2, (int(pend.sum()),), generator=g0, dtype=torch.int32)
    col_lr = torch.cat([torch.full((3,), 2.5e-3), torch.full((45,), 1.25e-4)]).contiguous()
    b1, b2, eps = 0.9 ** 4, 0.999 ** 4, 5e-16
    rp, rm, rv = [t.clone().double() for t in (p, m, v)]
    p0 = p.clone()
    for s in range(1, to_step + 1):
        active = last.long() < s
        has = (gstep.long() == s) & active
        rows = torch.nonzero(has if sparse else active).flatten()
        gg = torch.zeros(N, cols, dtype=torch.float64)
        gg[has] = g[has].double()
        O.adam_rows(rp, gg, rm, rv, rows, col_lr.double(), b1, b2, eps, step=s, scale=0.25)
    rows = torch.randperm(N, generator=g0).to(torch.int32).contiguous()
    stage = torch.empty(N, cols)
    assert L.clmgs_host_pool_start(3) == 3
    _lib.check(L.clmgs_host_rows_prepare(P(p), P(g), P(m), P(v), P(last), P(gstep), P(rows), N, cols, P(col_lr),
                                         b1, b2, eps, to_step, to_step + 1, 1, 0.25, 256, P(stage), sparse))
    rl = lambda a, b: ((a.double() - b).norm() / (b.norm() + 1e-30)).item()
    assert rl(p, rp) < 1e-6 and rl(m, rm) < 1e-6 and rl(v, rv) < 1e-6
    assert torch.equal(stage, p[rows.long()])
    assert int(last.min()) == to_step == int(last.max()) and int(gstep.min()) == to_step + 1
    untouched = (~pend)[:100]  # all-zero moments, no gradient: bit-identical
    assert torch.equal(p[:100][untouched], p0[:100][untouched])
