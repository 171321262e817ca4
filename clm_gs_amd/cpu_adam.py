"""cpu_adam.FusedCPUAdam / CPUAdam surface (optimizer.py:130-144,
strategies/clm_offload/gaussian_model.py:161-211, strategies/clm_offload/engine.py:316-328)
on top of the C ABI.

One [N,48] parameter tensor with PER-COLUMN learning rates, DeepSpeed-style bias-corrected
Adam.  The tensor (and its grad / exp_avg / exp_avg_sq) may live
  * in pinned host memory  -> clmgs_host_adam_rows (threads on the host cores, GIL released,
    busy-waits on the pinned signal flags written by clm_kernels.set_signal), or
  * in HBM                 -> clmgs_adam_rows on whatever stream is current.
Step counter: ONE global step per optimizer call (step()/batched_sparse_step()), shared by
all rows -- with every row touched each call this equals dense torch.optim.Adam.
"""
import ctypes
import os

import torch

from . import _lib
from .host import pinned_empty


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class FusedCPUAdam(torch.optim.Optimizer):
    def __init__(self, params, columns_sizes, columns_lr, lr=1e-3, bias_correction=True,
                 betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, adamw_mode=False,
                 fp32_optimizer_states=True, n_threads=0, state_tensors=None):
        assert weight_decay == 0 and not amsgrad, "not used by CLM-GS"
        defaults = dict(lr=lr, bias_correction=bias_correction, betas=betas, eps=eps,
                        weight_decay=weight_decay, amsgrad=amsgrad)
        super().__init__(params, defaults)
        assert len(self.param_groups) == 1 and len(self.param_groups[0]["params"]) == 1
        self.columns_sizes = list(columns_sizes)
        self.columns_lr = torch.tensor(list(columns_lr), dtype=torch.float32)  # mutable, host
        self.n_threads = n_threads or int(os.environ.get("CLMGS_HOST_ADAM_THREADS", "0"))
        self.global_step = 0
        self._init_state(state_tensors)

    # -- state is eager (the reference asserts it exists right after construction)
    def _init_state(self, state_tensors=None):
        p = self.param_groups[0]["params"][0]
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            if state_tensors is not None:  # caller-owned (capacity-sized) buffers, viewed [:N]
                st["exp_avg"], st["exp_avg_sq"] = state_tensors
            elif p.is_cuda:
                st["exp_avg"] = torch.zeros_like(p.data)
                st["exp_avg_sq"] = torch.zeros_like(p.data)
            else:
                st["exp_avg"] = pinned_empty(tuple(p.shape)).zero_()
                st["exp_avg_sq"] = pinned_empty(tuple(p.shape)).zero_()

    def _col_lr(self, device):
        """Per-column learning rates as a tensor on `device`.  The device copy is cached per value
        set: a host->device copy from pageable memory blocks the host until the stream has
        drained, i.e. it would be a hidden end-of-batch synchronisation."""
        lrs = tuple(float(l) for l in self.columns_lr.tolist())
        key = (str(device), lrs)
        cache = self.__dict__.setdefault("_col_lr_cache", {})
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            v = torch.cat([torch.full((n,), l) for n, l in zip(self.columns_sizes, lrs)])
            cache[key] = v.to(device) if device.type == "cuda" else v
        return cache[key]

    def _update(self, rows, signal, grad_scale, zero_grad, step):
        g = self.param_groups[0]
        p = g["params"][0]
        st = self.state[p]
        b1, b2 = g["betas"]
        cols = p.shape[1]
        grad = p.grad
        assert grad is not None, "parameters.grad must be set (clm_offload/engine.py:317)"
        L = _lib.lib()
        n_rows = rows.numel() if rows is not None else p.shape[0]
        if p.is_cuda:
            col_lr = self._col_lr(p.device)
            _lib.check(L.clmgs_adam_rows(
                _lib.stream(), _p(p.data), _p(grad), _p(st["exp_avg"]), _p(st["exp_avg_sq"]),
                _p(rows), 1 if (rows is not None and rows.dtype == torch.int64) else 0, None,
                n_rows, cols, _p(col_lr), b1, b2, g["eps"], step, int(g["bias_correction"]),
                grad_scale, int(zero_grad)))
            self._keep = col_lr  # keep alive until the stream consumed it
        else:
            col_lr = self._col_lr(torch.device("cpu")).contiguous()
            if rows is not None:
                assert rows.dtype == torch.int32 and not rows.is_cuda
            _lib.check(L.clmgs_host_adam_rows(
                _p(p.data), _p(grad), _p(st["exp_avg"]), _p(st["exp_avg_sq"]), _p(rows), n_rows,
                cols, _p(col_lr), b1, b2, g["eps"], step, int(g["bias_correction"]), grad_scale,
                int(zero_grad), signal, self.n_threads))

    @torch.no_grad()
    def step(self, closure=None):
        self.global_step += 1
        self._update(None, None, 1.0, False, self.global_step)
        self.state[self.param_groups[0]["params"][0]]["step"] = self.global_step

    @torch.no_grad()
    def batched_sparse_step(self, batch_size, batched_sparse_indices, signal_tensor_pinned,
                            version=3, scale=1.0, sparse_adam=False):
        """Row groups [untouched, finished-after-mb0, ..., finished-after-mb(bsz-1)]
        (clm_offload/engine.py:203-213).  Group 0 needs no signal and is skipped when
        sparse_adam; group i+1 waits for signal[i].  version 3 zeroes consumed grad rows."""
        assert len(batched_sparse_indices) == batch_size + 1
        self.global_step += 1
        step = self.global_step
        zero = version == 3
        if not sparse_adam and batched_sparse_indices[0].numel():
            self._update(batched_sparse_indices[0], None, scale, zero, step)
        for i in range(batch_size):
            sig = None
            if signal_tensor_pinned is not None:
                sig = ctypes.c_void_p(signal_tensor_pinned.data_ptr() + 4 * i)
            self._update(batched_sparse_indices[i + 1], sig, scale, zero, step)
        self.state[self.param_groups[0]["params"][0]]["step"] = step

    def zero_grad(self, set_to_none=False):
        p = self.param_groups[0]["params"][0]
        if p.grad is not None:
            if set_to_none:
                p.grad = None
            else:
                p.grad.zero_()


class CPUAdam(FusedCPUAdam):
    """cpu_adam.CPUAdam (naive_offload only; signature-compatible): dense .step() and
    .sparse_step(indices i32 cpu) over host tensors with one learning rate."""

    def __init__(self, params, lr=1e-3, eps=1e-8, betas=(0.9, 0.999)):
        cols = params[0]["params"][0].shape[1]
        super().__init__(params, [cols], [params[0].get("lr", lr)], lr=lr, betas=betas, eps=eps)

    @torch.no_grad()
    def sparse_step(self, sparse_indices):
        self.global_step += 1
        self._update(sparse_indices, None, 1.0, False, self.global_step)
