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

    def get_all_states(self):
        return [self.gpu_adam.state, self.cpu_adam.state]

    def zero_grad(self, set_to_none=False):
        self.gpu_adam.zero_grad(set_to_none)
        self.cpu_adam.zero_grad(set_to_none)

    def step(self, closure=None):
        self.gpu_adam.step()
        self.cpu_adam.step()
        self.state = self.gpu_adam.state | self.cpu_adam.state
