"""SelectiveAdam and UnifiedAdam with the attribute surface the model and engine touch
(optimizer.py:6-184 of the reference: .gpu_adam, .cpu_adam, .columns_lr, .param_groups,
.state, step/zero_grad)."""
import torch

from . import cpu_adam
from .clm_kernels import selective_adam_update
from .host import is_pinned


class SelectiveAdam(torch.optim.Adam):
    """Visibility-masked fused Adam (Taming-3DGS flavour: no bias correction)."""

    def __init__(self, params, eps, betas):
        super().__init__(params=params, eps=eps, betas=betas)

    @torch.no_grad()
    def step(self, visibility):
        N = visibility.numel()
        for group in self.param_groups:
            assert len(group["params"]) == 1, "more than one tensor in group"
            param = group["params"][0]
            if param.grad is None:
                continue
            state = self.state[param]
            if len(state) == 0:
                state["step"] = torch.tensor(0.0, dtype=torch.float32)
                state["exp_avg"] = torch.zeros_like(param, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(param, memory_format=torch.preserve_format)
            beta1, beta2 = group["betas"]
            selective_adam_update(param, param.grad, state["exp_avg"], state["exp_avg_sq"],
                                  visibility, group["lr"], beta1, beta2, group["eps"], N,
                                  param.numel() // N)


class UnifiedAdam(torch.optim.Optimizer):
    """Two independent Adams behind one interface: torch fused Adam (or SelectiveAdam) for the
    GPU-resident groups and FusedCPUAdam for the group named "parameters" (the [N,48] SH rows,
    pinned host or HBM-resident)."""

    def __init__(self, params, columns_sizes, columns_lr, lr=1e-3, bias_correction=True,
                 betas=(0.9, 0.999), eps=1e-15, weight_decay=0, amsgrad=False, adamw_mode=False,
                 fp32_optimizer_states=True, fused=False, sparse=False, state_tensors=None):
        params_device, params_rows = [], []
        for p in params:
            if p["name"] == "parameters":
                t = p["params"][0]
                assert t.is_cuda or is_pinned(t), "SH rows must be pinned host or device memory"
                params_rows.append(p)
            else:
                assert p["params"][0].is_cuda
                params_device.append(p)
        if sparse:
            self.gpu_adam = SelectiveAdam(params_device, eps=eps, betas=betas)
        else:
            self.gpu_adam = torch.optim.Adam(params_device, lr=0.0, eps=eps, fused=fused)
        self.cpu_adam = cpu_adam.FusedCPUAdam(
            params_rows, columns_sizes=columns_sizes, columns_lr=columns_lr, lr=0.0,
            bias_correction=bias_correction, betas=betas, eps=eps, weight_decay=weight_decay,
            amsgrad=amsgrad, adamw_mode=adamw_mode, fp32_optimizer_states=fp32_optimizer_states,
            state_tensors=state_tensors)
        self.columns_lr = self.cpu_adam.columns_lr
        self.param_groups = self.gpu_adam.param_groups + self.cpu_adam.param_groups
        self.state = self.gpu_adam.state | self.cpu_adam.state

    @torch.no_grad()
    def gpu_step_scaled(self, grad_scale=1.0):
        """Dense Adam of the GPU-resident groups through clmgs_adam_rows: gradient scale (the
        engine's `grad /= bsz`), update and gradient zeroing in ONE pass per tensor instead of
        torch's three (div, fused Adam, fresh zeros next batch).  State stays in
        torch.optim.Adam's own layout (step / exp_avg / exp_avg_sq per parameter), so the
        densification surgery, capture / restore and a later torch .step() keep working; same
        bias-corrected formula as torch (step_size = lr / bc1, denom = sqrt(v) / sqrt(bc2) + eps)."""
        from .clm_kernels import adam_rows
        assert not isinstance(self.gpu_adam, SelectiveAdam)
        for group in self.gpu_adam.param_groups:
            p = group["params"][0]
            if p.grad is None:
                continue
            st = self.gpu_adam.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            cache = self.__dict__.setdefault("_gpu_steps", {})
            if id(st["step"]) not in cache:  # one host read per (re)created state, then counted here
                cache.clear() if len(cache) > 64 else None
                cache[id(st["step"])] = int(st["step"].item())
            cache[id(st["step"])] += 1
            step = cache[id(st["step"])]
            st["step"] += 1
            # one learning rate per tensor -> the [N,d] layout is irrelevant: stream it as float4 rows
            cols = 4 if p.numel() % 4 == 0 else 1
            key = (cols, float(group["lr"]))
            lrs = self.__dict__.setdefault("_col_lr_cache", {})
            if key not in lrs:
                if len(lrs) > 256:
                    lrs.clear()
                lrs[key] = torch.full((cols,), float(group["lr"]), dtype=torch.float32, device=p.device)
            b1, b2 = group["betas"]
            adam_rows(p.data.view(-1, cols), p.grad.view(-1, cols), st["exp_avg"].view(-1, cols),
                      st["exp_avg_sq"].view(-1, cols), None, lrs[key], b1, b2, group["eps"], step, True,
                      float(grad_scale), True)
        self.state = self.gpu_adam.state | self.cpu_adam.state

    @torch.no_grad()
    def gpu_step_packed(self, packed_p, packed_g, grad_scale=1.0, g_stamp=None, cur_step=0, row_range=None):
        """Dense Adam of the four GPU-resident tensors from the packed [N,12] gradient table
        (clmgs_adam_small_packed): updates p / exp_avg / exp_avg_sq of every group in place,
        refreshes the packed parameter mirror and zeroes the gradient table, all in one pass.
        g_stamp / cur_step: first-touch gradient table (only rows stamped cur_step carry a gradient;
        nothing is zeroed).  row_range = (lo, hi): only these rows are stepped (camera-DP: the small attributes are
        computed by the owner of a row range, clm_offload/gaussian_model.py small_owner); the step counters advance
        as usual -- they are per tensor, and every rank steps its range at every batch."""
        import ctypes
        from . import _lib
        assert not isinstance(self.gpu_adam, SelectiveAdam)
        groups = {g["name"]: g for g in self.gpu_adam.param_groups}
        order = [groups[n] for n in ("xyz", "opacity", "scaling", "rotation")]
        ps, ms, vs, lrs, steps = [], [], [], [], []
        step = None
        cache = self.__dict__.setdefault("_gpu_steps", {})
        for group in order:
            p = group["params"][0]
            st = self.gpu_adam.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if id(st["step"]) not in cache:
                cache.clear() if len(cache) > 64 else None
                cache[id(st["step"])] = int(st["step"].item())
            cache[id(st["step"])] += 1
            steps.append(st["step"])
            step = cache[id(st["step"])] if step is None else step
            assert cache[id(st["step"])] == step, "the four groups step together"
            assert p.is_contiguous() and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()
            ps.append(p.data_ptr()); ms.append(st["exp_avg"].data_ptr()); vs.append(st["exp_avg_sq"].data_ptr())
            lrs.append(float(group["lr"]))
        torch._foreach_add_(steps, 1)  # the four torch-Adam step counters: one launch instead of four
        g0 = order[0]
        arr = lambda xs: (ctypes.c_void_p * 4)(*xs)
        L = _lib.lib()
        lo, hi = (0, -1) if row_range is None else (int(row_range[0]), int(row_range[1]))
        _lib.check(L.clmgs_adam_small_packed_range(
            _lib.stream(), int(packed_p.shape[0]), lo, hi, arr(ps), arr(ms), arr(vs), (ctypes.c_double * 4)(*lrs),
            _lib.dptr(packed_p), _lib.dptr(packed_g), float(g0["betas"][0]), float(g0["betas"][1]),
            float(g0["eps"]), int(step), 1, float(grad_scale),
            None if g_stamp is None else _lib.dptr(g_stamp, torch.int32), int(cur_step)))
        self.state = self.gpu_adam.state | self.cpu_adam.state

    def get_all_states(self):
        return [self.gpu_adam.state, self.cpu_adam.state]

    def zero_grad(self, set_to_none=False):
        self.gpu_adam.zero_grad(set_to_none)
        self.cpu_adam.zero_grad(set_to_none)

    def step(self, closure=None):
        self.gpu_adam.step()
        self.cpu_adam.step()
        self.state = self.gpu_adam.state | self.cpu_adam.state
