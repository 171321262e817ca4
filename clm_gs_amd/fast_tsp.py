"""fast_tsp.find_tour stand-in (strategies/clm_offload/engine.py:179).  The upstream solver is
a 1 ms time-budgeted local search and therefore non-deterministic; this one is a deterministic
nearest-neighbour + 2-opt heuristic in the C ABI.  The order only changes transfer volume and
float summation order, never the batch gradient in exact arithmetic."""
import ctypes

import torch

from . import _lib


def find_tour(dist, duration_seconds=0.001):
    n = len(dist)
    d = torch.tensor(dist, dtype=torch.int64).contiguous()
    assert d.shape == (n, n)
    tour = torch.empty(n, dtype=torch.int32)
    _lib.check(_lib.lib().clmgs_tsp_tour(n, ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(tour.data_ptr())))
    return tour.tolist()
