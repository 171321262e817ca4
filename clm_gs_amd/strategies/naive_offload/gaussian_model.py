"""GaussianModelNaiveOffload: EVERY parameter and all optimizer state live in pinned host memory
(reference: strategies/naive_offload/gaussian_model.py:27-680); the GPU only ever holds a
per-batch copy.  Scope row f4 -- the comparison baseline of the three-strategy table, not a hot
path.  Densification runs the shared base-class logic against temporary DEVICE copies of the 11
small attributes (the statistics live on the device anyway) and resizes the pinned host tables.

Layout (MI355X-side choice): two pinned buffers, `small[N,12]` = xyz 3 | opacity 1 | scaling 3 |
rotation 4 | pad, and `parameters[N,48]` (SH rows), each with twin grad / exp_avg / exp_avg_sq
buffers, so a batch moves two contiguous blocks per direction and the host Adam
(clmgs_host_adam_rows, per-column learning rates) walks two dense row tables.
`_xyz/_opacity/_scaling/_rotation/_features_dc/_features_rest` are host VIEWS with the
reference's shapes.
"""
import torch
from torch import nn

from ...cpu_adam import FusedCPUAdam
from ...host import pinned_empty
from ..base_gaussian_model import BaseGaussianModel

_SMALL = {"xyz": (0, 3), "opacity": (3, 4), "scaling": (4, 7), "rotation": (7, 11)}


class GaussianModelNaiveOffload(BaseGaussianModel):
    def __init__(self, sh_degree, only_for_rendering=False):
        super().__init__(sh_degree, only_for_rendering)
        self.device = "cpu"

    # ------------------------------------------------------------------ storage
    def create_from_tensors(self, xyz, shs48, scaling, rotation, opacity, spatial_lr_scale=1.0):
        self.spatial_lr_scale = spatial_lr_scale
        n = xyz.shape[0]
        self._small = nn.Parameter(pinned_empty((n, 12)).zero_().requires_grad_(True))
        self._parameters = nn.Parameter(pinned_empty((n, 48)).requires_grad_(True))
        with torch.no_grad():
            for name, src in (("xyz", xyz), ("opacity", opacity), ("scaling", scaling), ("rotation", rotation)):
                a, b = _SMALL[name]
                self._small[:, a:b] = src.detach().float().cpu().reshape(n, b - a)
            self._parameters.copy_(shs48.detach().float().cpu().reshape(n, 48))
        self._bind_views()
        self.max_radii2D = torch.zeros((n,), device="cuda")

    def _bind_views(self):
        d = self._small.detach()
        self._xyz, self._opacity = d[:, 0:3], d[:, 3:4]
        self._scaling, self._rotation = d[:, 4:7], d[:, 7:11]
        f = self._parameters.detach().view(-1, 16, 3)
        self._features_dc, self._features_rest = f[:, :1, :], f[:, 1:, :]

    @property
    def get_features(self):
        return self._parameters.detach().view(-1, 16, 3)

    def all_parameters(self):
        return [self._small, self._parameters]

    # ---------------------------------------------------------------- optimiser
    def training_setup(self, training_args):
        self.percent_dense = training_args.percent_dense
        n = self._small.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device="cuda")
        self.denom = torch.zeros((n, 1), device="cuda")
        a = self.args
        self._small.grad = pinned_empty((n, 12)).zero_()
        self._parameters.grad = pinned_empty((n, 48)).zero_()
        xyz_lr = training_args.position_lr_init * self.spatial_lr_scale * a.lr_scale_pos_and_scale
        self.small_adam = FusedCPUAdam(
            [{"params": [self._small], "lr": 0.0, "name": "small"}], [3, 1, 3, 4, 1],
            [xyz_lr, training_args.opacity_lr, training_args.scaling_lr * a.lr_scale_pos_and_scale,
             training_args.rotation_lr, 0.0], lr=0.0, betas=(0.9, 0.999), eps=1e-15)
        self.row_adam = FusedCPUAdam(
            [{"params": [self._parameters], "lr": 0.0, "name": "parameters"}], [3, 45],
            [training_args.feature_lr, training_args.feature_lr / 20.0], lr=0.0, betas=(0.9, 0.999),
            eps=1e-15)
        self.optimizer = _NaiveOptimizer(self.small_adam, self.row_adam)
        lr_scale = self._scale_groups_for_bsz(training_args)
        if training_args.lr_scale_mode in ("linear", "sqrt"):
            self.small_adam.columns_lr *= lr_scale
            self.row_adam.columns_lr *= lr_scale
        self._xyz_lr_scale = lr_scale

    def update_learning_rate(self, iteration):
        lr = self.xyz_scheduler_args(iteration)
        self.small_adam.columns_lr[0] = lr
        return lr

    # ------------------------------------------------------------ densification
    def _device_views(self, on):
        """While densify_and_prune runs, _xyz/_opacity/_scaling/_rotation are device copies (the
        shared logic indexes them with device masks); afterwards they are host views again."""
        if on:
            d = self._small.detach().to("cuda")
            self._xyz, self._opacity = d[:, 0:3].contiguous(), d[:, 3:4].contiguous()
            self._scaling, self._rotation = d[:, 4:7].contiguous(), d[:, 7:11].contiguous()
        else:
            self._bind_views()
        self._dev_views = on

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        self._device_views(True)
        try:
            super().densify_and_prune(max_grad, min_opacity, extent, max_screen_size)
        finally:
            self._device_views(False)

    def _shs48_rows(self, mask):
        p = self._parameters.detach()
        if mask is None:
            return p
        return p[mask.cpu()].to(mask.device)

    def _resize(self, new_small, new_rows, state_fn):
        """Swap in resized pinned tables (+ gradients) and carry the host Adam state along."""
        for opt, name, new in ((getattr(self, "small_adam", None), "_small", new_small),
                               (getattr(self, "row_adam", None), "_parameters", new_rows)):
            old = getattr(self, name)
            buf = pinned_empty(tuple(new.shape))
            buf.copy_(new)
            p = nn.Parameter(buf.requires_grad_(True))
            if opt is not None:
                st = opt.state.pop(old, None)
                p.grad = pinned_empty(tuple(new.shape)).zero_()
                opt.param_groups[0]["params"][0] = p
                if st is not None:
                    for k in ("exp_avg", "exp_avg_sq"):
                        v = state_fn(st[k])
                        st[k] = pinned_empty(tuple(v.shape))
                        st[k].copy_(v)
                    opt.state[p] = st
            setattr(self, name, p)
        if getattr(self, "optimizer", None) is not None:
            self.optimizer.param_groups = self.small_adam.param_groups + self.row_adam.param_groups
        n = new_small.shape[0]
        if getattr(self, "_dev_views", False):
            self._device_views(True)
        else:
            self._bind_views()
        return n

    def _append_rows(self, new):
        k = new["xyz"].shape[0]
        add_small = torch.zeros((k, 12))
        for name in ("xyz", "opacity", "scaling", "rotation"):
            a, b = _SMALL[name]
            add_small[:, a:b] = new[name].detach().float().cpu().reshape(k, b - a)
        add_rows = new["shs48"].detach().float().cpu().reshape(k, 48)
        self._resize(torch.cat((self._small.detach(), add_small), 0), torch.cat((self._parameters.detach(), add_rows), 0),
                     lambda s: torch.cat((s, torch.zeros((k, s.shape[1]))), 0))

    def prune_points(self, mask):
        keep = (~mask).cpu()
        self._resize(self._small.detach()[keep], self._parameters.detach()[keep], lambda s: s[keep])
        kd = keep.to(self.max_radii2D.device)
        self.xyz_gradient_accum = self.xyz_gradient_accum[kd]
        self.denom = self.denom[kd]
        self.max_radii2D = self.max_radii2D[kd]

    def reset_opacity(self):
        from ... import utils
        with torch.no_grad():
            op = torch.sigmoid(self._small.detach()[:, 3:4])
            self._small.detach()[:, 3:4] = utils.inverse_sigmoid(torch.min(op, torch.ones_like(op) * 0.01))
            for k in ("exp_avg", "exp_avg_sq"):  # the reference zeroes the opacity moments
                self.small_adam.state[self._small][k][:, 3:4] = 0.0
        self._bind_views()


class _NaiveOptimizer:
    """The two host Adams behind one optimizer-shaped object (param_groups for the shared bsz
    scaling, step / sparse_step / zero_grad as naive_offload/engine.py:318-334 calls them)."""

    def __init__(self, small, rows):
        self.small, self.rows = small, rows
        self.param_groups = small.param_groups + rows.param_groups

    @property
    def state(self):
        s = dict(self.small.state)
        s.update(self.rows.state)
        return s

    def step(self, grad_scale=1.0):
        for o in (self.small, self.rows):
            o.global_step += 1
            o._update(None, None, grad_scale, True, o.global_step)

    def sparse_step(self, sparse_indices, grad_scale=1.0):
        for o in (self.small, self.rows):
            o.global_step += 1
            o._update(sparse_indices, None, grad_scale, True, o.global_step)

    def zero_grad(self, set_to_none=False):
        pass  # the host Adam zeroes every gradient row it consumes
