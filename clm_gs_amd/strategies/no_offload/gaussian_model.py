"""GaussianModelNoOffload: all six parameter tensors live on the GPU
(reference: strategies/no_offload/gaussian_model.py:27-813)."""
import torch
from torch import nn

from ... import utils
from ...optimizer import SelectiveAdam
from ..base_gaussian_model import BaseGaussianModel


class GaussianModelNoOffload(BaseGaussianModel):
    _GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def _attr(self, name):
        return {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest",
                "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}[name]

    def create_from_tensors(self, xyz, shs48, scaling, rotation, opacity, spatial_lr_scale=1.0):
        self.spatial_lr_scale = spatial_lr_scale
        n = xyz.shape[0]
        f = shs48.reshape(n, 16, 3).float().cuda()
        self._xyz = nn.Parameter(xyz.float().cuda().contiguous().requires_grad_(True))
        self._features_dc = nn.Parameter(f[:, :1, :].contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(f[:, 1:, :].contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scaling.float().cuda().contiguous().requires_grad_(True))
        self._rotation = nn.Parameter(rotation.float().cuda().contiguous().requires_grad_(True))
        self._opacity = nn.Parameter(opacity.float().cuda().contiguous().requires_grad_(True))
        self.max_radii2D = torch.zeros((n,), device="cuda")

    def all_parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity]

    def training_setup(self, training_args):
        self.percent_dense = training_args.percent_dense
        n = self.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device="cuda")
        self.denom = torch.zeros((n, 1), device="cuda")
        a = self.args
        l = [
            {"params": [self._xyz], "lr": training_args.position_lr_init * self.spatial_lr_scale * a.lr_scale_pos_and_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": training_args.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": training_args.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": training_args.scaling_lr * a.lr_scale_pos_and_scale, "name": "scaling"},
            {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
        ]
        if a.sparse_adam:
            self.optimizer = SelectiveAdam(l, eps=1e-15, betas=(0.9, 0.999))
        else:
            self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15, fused=True)
        self._scale_groups_for_bsz(training_args)

    # ------------------------------------------------ optimizer-state surgery
    def _replace(self, name, new_tensor, state_fn):
        for g in self.optimizer.param_groups:
            if g["name"] != name:
                continue
            old = g["params"][0]
            st = self.optimizer.state.get(old, None)
            if st is not None:
                for k in ("exp_avg", "exp_avg_sq"):
                    if k in st:
                        st[k] = state_fn(st[k])
                del self.optimizer.state[old]
            g["params"][0] = nn.Parameter(new_tensor.requires_grad_(True))
            if st is not None:
                self.optimizer.state[g["params"][0]] = st
            setattr(self, self._attr(name), g["params"][0])
            return g["params"][0]
        raise KeyError(name)

    def _append_rows(self, new):
        k = new["xyz"].shape[0]
        f = new["shs48"].reshape(k, 16, 3)
        ext = {"xyz": new["xyz"], "f_dc": f[:, :1, :].contiguous(), "f_rest": f[:, 1:, :].contiguous(),
               "opacity": new["opacity"], "scaling": new["scaling"], "rotation": new["rotation"]}
        for name in self._GROUPS:
            cur = getattr(self, self._attr(name)).detach()
            e = ext[name]
            self._replace(name, torch.cat((cur, e), dim=0),
                          lambda s, e=e: torch.cat((s, torch.zeros_like(e)), dim=0))

    def prune_points(self, mask):
        keep = ~mask
        for name in self._GROUPS:
            cur = getattr(self, self._attr(name)).detach()
            self._replace(name, utils.select_rows(cur, keep).contiguous(), lambda s: utils.select_rows(s, keep).contiguous())
        self.xyz_gradient_accum = utils.select_rows(self.xyz_gradient_accum, keep)
        self.denom = utils.select_rows(self.denom, keep)
        self.max_radii2D = utils.select_rows(self.max_radii2D, keep)

    def permute_rows(self, order):
        order = order.to(self._xyz.device)
        assert order.numel() == self._xyz.shape[0]
        for name in self._GROUPS:
            cur = getattr(self, self._attr(name)).detach()
            self._replace(name, utils.gather_rows(cur, order), lambda s: utils.gather_rows(s, order))
        self.xyz_gradient_accum = utils.gather_rows(self.xyz_gradient_accum, order)
        self.denom = utils.gather_rows(self.denom, order)
        self.max_radii2D = utils.gather_rows(self.max_radii2D, order)

    def _shs48_rows(self, mask):
        f = self.get_features.detach()
        f = f if mask is None else utils.select_rows(f, mask)
        return f.reshape(f.shape[0], 48)

    def reset_opacity(self):
        new = utils.inverse_sigmoid(torch.min(self.get_opacity.detach(), torch.ones_like(self._opacity) * 0.01))
        self._replace("opacity", new, lambda s: torch.zeros_like(s))
