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
