"""Pinned (page-locked, device-mapped) host tensors.

Replaces numba.cuda.pinned_array at strategies/clm_offload/gaussian_model.py:34-44:
exact-size hipHostMalloc through the C ABI, wrapped as a torch tensor without a copy.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_NP = {torch.float32: np.float32, torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8}


class _PinnedBlock:
    """Owns one hipHostMalloc allocation; freed when the last tensor view dies."""

    def __init__(self, nbytes):
        self.ptr = _lib.lib().clmgs_pinned_alloc(max(int(nbytes), 16))
        if not self.ptr:
            msg = _lib.lib().clmgs_last_error()
            raise _lib.ClmgsError(f"pinned allocation of {nbytes} bytes failed: {msg.decode() if msg else ''}")
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().clmgs_pinned_free(ctypes.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=torch.float32):
    """Uninitialised pinned host tensor of exactly prod(shape) elements."""
    if isinstance(shape, int):
        shape = (shape,)
    n = 1
    for s in shape:
        n *= int(s)
    itemsize = torch.empty((), dtype=dtype).element_size()
    block = _PinnedBlock(n * itemsize)
    buf = (ctypes.c_char * max(n * itemsize, 16)).from_address(block.ptr)
    # ownership rides on the STORAGE, not on one tensor object: torch keeps `arr` alive for as long
    # as any tensor / view / Parameter shares the storage, `arr.base` is `buf`, `buf` owns the block
    buf._clmgs_block = block
    arr = np.frombuffer(buf, dtype=_NP[dtype], count=n).reshape(tuple(int(s) for s in shape))
    t = torch.from_numpy(arr)
    t._clmgs_pinned = True
    return t


def is_pinned(t):
    """True for pinned_empty() tensors (and views made through this module's users) or torch-pinned."""
    return bool(getattr(t, "_clmgs_pinned", False)) or t.is_pinned()
