"""Shared Gaussian-model logic (reference: strategies/base_gaussian_model.py:32-399).

Storage is described once, by name: every optimisable attribute is a row-major [N, d]
tensor registered in ``self._store``; densify / clone / split / prune are written against
that table, so the optimizer-state surgery exists exactly once for both strategies.
"""
from abc import ABC, abstractmethod

import numpy as np
import torch

from .. import utils
from ..clm_kernels import densify_stats
from ..utils import RGB2SH, build_rotation, get_expon_lr_func, inverse_sigmoid


class BaseGaussianModel(ABC):
    def setup_functions(self):
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    def __init__(self, sh_degree: int, only_for_rendering: bool = False):
        self.args = utils.get_args()
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self._xyz = torch.empty(0)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self._scaling = torch.empty(0)
        self._rotation = torch.empty(0)
        self._opacity = torch.empty(0)
        self._parameters = torch.empty(0)
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.parameters_buffer = torch.empty(0)
        self.parameters_grad_buffer = torch.empty(0)
        self.only_for_rendering = only_for_rendering
        self.setup_functions()
        self.device = "cuda"
        # densify_and_split draws from this generator; camera-DP replicas seed it identically
        self.split_generator = None

    # ------------------------------------------------------------------ getters
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    @property
    @abstractmethod
    def get_features(self):
        ...

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ------------------------------------------------------------ initialisation
    def _init_values_from_points(self, points, colors, dist2=None):
        """xyz, SH dc from RGB, log-scale from mean 3-NN distance, identity quats,
        opacity 0.1 (clm_offload/gaussian_model.py:24-111)."""
        points = torch.as_tensor(np.asarray(points)).float()
        colors = torch.as_tensor(np.asarray(colors)).float()
        n = points.shape[0]
        feats = torch.zeros((n, 16, 3))
        feats[:, 0, :] = RGB2SH(colors)
        if dist2 is None:
            from ..simple_knn import distCUDA2
            dist2 = distCUDA2(points.cuda()).cpu()
        dist2 = torch.clamp_min(dist2, 1e-7)
        scales = torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3)
        rots = torch.zeros((n, 4))
        rots[:, 0] = 1
        opac = inverse_sigmoid(0.1 * torch.ones((n, 1)))
        return points, feats.reshape(n, 48), scales, rots, opac

    @abstractmethod
    def create_from_tensors(self, xyz, shs48, scaling, rotation, opacity, spatial_lr_scale=1.0):
        ...

    def create_from_pcd(self, pcd, spatial_lr_scale: float, subsample_ratio: float = 1.0):
        pts, shs, sc, rot, op = self._init_values_from_points(pcd.points, pcd.colors)
        if subsample_ratio != 1.0:
            g = torch.Generator().manual_seed(1)
            keep = torch.randperm(pts.shape[0], generator=g)[: int(pts.shape[0] * subsample_ratio)].sort().values
            pts, shs, sc, rot, op = pts[keep], shs[keep], sc[keep], rot[keep], op[keep]
        self.create_from_tensors(pts, shs, sc, rot, op, spatial_lr_scale)

    # ---------------------------------------------------------------- optimiser
    @abstractmethod
    def all_parameters(self):
        ...

    @abstractmethod
    def training_setup(self, training_args):
        ...

    def _scale_groups_for_bsz(self, training_args):
        """bsz scaling of lr / eps / betas (no_offload/gaussian_model.py:223-246).  Under camera-DP the
        optimizer sees the GLOBAL batch (bsz per rank x ranks), so that is what scales."""
        from .. import dp
        bsz = self.args.bsz * dp.world_size()
        mode = training_args.lr_scale_mode
        if mode == "linear":
            lr_scale = bsz
        elif mode == "sqrt":
            lr_scale = float(np.sqrt(bsz))
        elif mode == "accumu":
            lr_scale = 1
        else:
            raise AssertionError(f"lr_scale_mode {mode} not supported.")
        for g in self.optimizer.param_groups:
            if mode == "accumu":
                continue
            g["lr"] *= lr_scale
            if mode == "sqrt" and "eps" in g:
                g["eps"] /= lr_scale
                g["betas"] = [b ** bsz for b in g["betas"]]
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=training_args.position_lr_init * self.spatial_lr_scale * lr_scale * self.args.lr_scale_pos_and_scale,
            lr_final=training_args.position_lr_final * self.spatial_lr_scale * lr_scale * self.args.lr_scale_pos_and_scale,
            lr_delay_mult=training_args.position_lr_delay_mult,
            max_steps=training_args.position_lr_max_steps)
        return lr_scale

    def update_learning_rate(self, iteration):
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                g["lr"] = self.xyz_scheduler_args(iteration)
                return g["lr"]

    # ---------------------------------------------------------- densification
    def _reset_stats(self):
        n = self.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device="cuda")
        self.denom = torch.zeros((n, 1), device="cuda")
        self.max_radii2D = torch.zeros((n,), device="cuda")

    @abstractmethod
    def _append_rows(self, new):
        """new: dict name -> [k,d] device tensors for xyz/opacity/scaling/rotation/shs48."""

    @abstractmethod
    def prune_points(self, mask):
        ...

    @abstractmethod
    def _shs48_rows(self, mask):
        """[k,48] device copy of the SH rows selected by a device bool mask."""

    @abstractmethod
    def reset_opacity(self):
        ...

    def densification_postfix(self, new_xyz, new_shs48, new_opacities, new_scaling, new_rotation):
        self._append_rows(dict(xyz=new_xyz, shs48=new_shs48, opacity=new_opacities,
                               scaling=new_scaling, rotation=new_rotation))
        self._reset_stats()

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = torch.norm(grads, dim=-1) >= grad_threshold
        sel &= torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent
        utils.get_log_file().write(f"Number of cloned gaussians: {int(sel.sum())}\n")
        self.densification_postfix(utils.select_rows(self._xyz.detach(), sel), self._shs48_rows(sel),
                                   utils.select_rows(self._opacity.detach(), sel), utils.select_rows(self._scaling.detach(), sel),
                                   utils.select_rows(self._rotation.detach(), sel))

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, defer_prune=False):
        """defer_prune=True: the split originals are NOT pruned here; their mask (over the rows after the append) is
        returned so that densify_and_prune can remove them together with the low-opacity rows in ONE compaction of the
        row tables (the reference prunes twice, clm/gaussian_model.py:687-692 and base_gaussian_model.py:364-388; both
        prunes keep the row order and the second mask only looks at values of rows the first one keeps, so one prune
        with the OR of the two masks leaves exactly the same rows in the same order)."""
        n_init = self.get_xyz.shape[0]
        padded = torch.zeros((n_init,), device="cuda")
        padded[: grads.shape[0]] = grads.squeeze()
        sel = padded >= grad_threshold
        sel &= torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent
        stds = utils.select_rows(self.get_scaling.detach(), sel).repeat(N, 1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=self.split_generator)
        utils.get_log_file().write(f"Number of split gaussians: {int(sel.sum())}\n")
        rots = build_rotation(utils.select_rows(self._rotation.detach(), sel)).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + utils.select_rows(self._xyz.detach(), sel).repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(utils.select_rows(self.get_scaling.detach(), sel).repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz, self._shs48_rows(sel).repeat(N, 1),
                                   utils.select_rows(self._opacity.detach(), sel).repeat(N, 1), new_scaling,
                                   utils.select_rows(self._rotation.detach(), sel).repeat(N, 1))
        prune = torch.cat((sel, torch.zeros(N * int(sel.sum()), device="cuda", dtype=torch.bool)))
        if defer_prune:
            return prune
        self.prune_points(prune)
        return None

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        """base_gaussian_model.py:364-388.  The screen-size prune is provably inert upstream
        (max_radii2D has just been zeroed by densification_postfix; asserts at :376-381), so
        only opacity and world-size pruning act."""
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        split_mask = self.densify_and_split(grads, max_grad, extent, defer_prune=True)
        prune_mask = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            assert torch.all(self.max_radii2D == 0)
            big_ws = self.get_scaling.max(dim=1).values > 0.1 * extent
            prune_mask = torch.logical_or(prune_mask, big_ws)
        mask = torch.logical_or(prune_mask, split_mask)  # one compaction instead of two
        if getattr(self, "fuse_sort_into_prune", False):
            # the trainer keeps the rows in Z-order (spatial_row_order): the compaction of this prune also re-sorts
            # (clm_offload model, HBM rows) -- the spatial_sort() the trainer calls next finds nothing left to do
            self.prune_points(mask, resort=True)
        else:
            self.prune_points(mask)

    # ----------------------------------------------------------- storage order
    def permute_rows(self, order):
        """Re-order the Gaussians: row i of every per-row tensor (parameters, optimizer moments,
        densification statistics, deferred-optimizer stamps) becomes old row order[i]."""
        raise NotImplementedError

    def spatial_sort(self):
        """Store the rows along a Z-order curve of (x, y) (utils.morton_order): the rows a camera sees and
        the rows a batch touches become contiguous runs of the row tables (coalesced gathers, TLB reach,
        streaming host walks) -- +6 % img/s HBM-resident, +24 % host-resident at 28 M.  Row order is not part
        of the model; call it after loading a scene and after densification (trainer.training does)."""
        self.permute_rows(utils.morton_order(self._xyz.detach()))

    # ------------------------------------------------------------- statistics
    def gsplat_add_densification_stats_exact_filter(self, viewspace_point_tensor_grad, radii,
                                                    send2gpu_final_filter_indices, width, height):
        """max_radii2D / |grad * (W/2,H/2)| / count on the filter's rows, every row
        (clm_offload/gaussian_model.py:833-851), as one fused kernel."""
        densify_stats(send2gpu_final_filter_indices, viewspace_point_tensor_grad, radii, width,
                      height, self.max_radii2D, self.xyz_gradient_accum, self.denom,
                      only_visible=False)

    def gsplat_add_densification_stats(self, viewspace_point_tensor_grad, send2gpu_visibility_filter,
                                       update_filter, width, height):
        """Boolean-mask form (no_offload/gaussian_model.py:767-783); does not touch max_radii2D."""
        grad = viewspace_point_tensor_grad
        g = grad[update_filter] * torch.tensor([width * 0.5, height * 0.5], device=grad.device)
        self.xyz_gradient_accum[send2gpu_visibility_filter] += torch.norm(g, dim=-1, keepdim=True)
        self.denom[send2gpu_visibility_filter] += 1

    # ------------------------------------------------------------- checkpoint
    def _opt_state(self, p):
        """State dict of parameter p in the optimizer that OWNS it (UnifiedAdam.state is only a
        merged snapshot of its two sub-optimizers)."""
        o = self.optimizer
        for sub in (getattr(o, "gpu_adam", None), getattr(o, "cpu_adam", None)):
            if sub is not None and any(p is q for g in sub.param_groups for q in g["params"]):
                return sub.state[p]
        return o.state[p]

    def capture(self):
        """Everything needed to resume training bit-for-bit in exact arithmetic: parameters,
        optimizer moments and step counters, densification statistics (scope row f3; the
        reference's capture/restore start with `assert False`, no_offload/gaussian_model.py:38-56)."""
        if hasattr(self, "flush_lazy_rows"):  # deferred row optimizers (HBM lazy rows / host rows)
            self.flush_lazy_rows()
        opt = {}
        # camera-DP with sharded row moments (clm_offload/gaussian_model.py): the full tables are assembled by a
        # collective, so capture() is then called on ALL ranks (as flush_lazy_rows above already requires)
        full = self.row_moments_full() if getattr(self, "moments_sharded", False) else None
        for g in self.optimizer.param_groups:
            st = self._opt_state(g["params"][0])
            if full is not None and g["name"] == "parameters":
                st = dict(st, exp_avg=full[0], exp_avg_sq=full[1])
            opt[g["name"]] = {
                "lr": g["lr"],
                "exp_avg": st["exp_avg"].detach().cpu().clone() if "exp_avg" in st else None,
                "exp_avg_sq": st["exp_avg_sq"].detach().cpu().clone() if "exp_avg_sq" in st else None,
                "step": float(st["step"]) if "step" in st else 0.0,
            }
        row_adam = getattr(self.optimizer, "cpu_adam", None)
        return {
            "xyz": self._xyz.detach().cpu().clone(), "opacity": self._opacity.detach().cpu().clone(),
            "scaling": self._scaling.detach().cpu().clone(), "rotation": self._rotation.detach().cpu().clone(),
            "shs48": self._shs48_rows(None).cpu().clone(), "active_sh_degree": self.active_sh_degree,
            "spatial_lr_scale": self.spatial_lr_scale, "optimizer": opt,
            "row_global_step": row_adam.global_step if row_adam is not None else None,
            "xyz_gradient_accum": self.xyz_gradient_accum.cpu().clone(), "denom": self.denom.cpu().clone(),
            "max_radii2D": self.max_radii2D.cpu().clone(),
        }

    def restore(self, state, training_args):
        self.create_from_tensors(state["xyz"], state["shs48"], state["scaling"], state["rotation"],
                                 state["opacity"], state["spatial_lr_scale"])
        self.active_sh_degree = state["active_sh_degree"]
        self.training_setup(training_args)
        for g in self.optimizer.param_groups:
            saved = state["optimizer"][g["name"]]
            p = g["params"][0]
            g["lr"] = saved["lr"]
            if saved["exp_avg"] is None:
                continue
            st = self._opt_state(p)
            if "exp_avg" in st:  # row optimizer: state lives in the model's capacity buffers
                a, b = 0, saved["exp_avg"].shape[0]
                if getattr(self, "moments_sharded", False) and g["name"] == "parameters":
                    a, b = self._mom_lo, self._mom_lo + st["exp_avg"].shape[0]  # this rank's shard of the saved tables
                st["exp_avg"].copy_(saved["exp_avg"][a:b])
                st["exp_avg_sq"].copy_(saved["exp_avg_sq"][a:b])
                st["step"] = int(saved["step"])
            else:              # torch Adam creates its state lazily
                st["step"] = torch.tensor(saved["step"], dtype=torch.float32, device=p.device)
                st["exp_avg"] = saved["exp_avg"].to(p.device)
                st["exp_avg_sq"] = saved["exp_avg_sq"].to(p.device)
        row_adam = getattr(self.optimizer, "cpu_adam", None)
        if row_adam is not None and state["row_global_step"] is not None:
            row_adam.global_step = state["row_global_step"]
            if getattr(self, "lazy_rows", False):
                self._row_last_step.fill_(row_adam.global_step)
                self._row_g_step.zero_()
            if getattr(self, "deferred_host_rows", False):
                self._host_last_step.fill_(row_adam.global_step)
                self._host_g_step.zero_()
        self.xyz_gradient_accum = state["xyz_gradient_accum"].cuda()
        self.denom = state["denom"].cuda()
        self.max_radii2D = state["max_radii2D"].cuda()

    # -------------------------------------------------------------------- I/O
    def _flush_before_read(self):
        """Deferred optimizer steps applied and (camera-DP owner modes: a collective, already done by the caller on all
        ranks, then a no-op here) the replicas completed BEFORE any parameter tensor is read out."""
        if hasattr(self, "flush_lazy_rows"):
            self.flush_lazy_rows()

    def save_tensors(self, folder):
        """Five-file .pt layout (clm_offload/gaussian_model.py:236-243)."""
        import os
        os.makedirs(folder, exist_ok=True)
        self._flush_before_read()
        torch.save(self._xyz.detach().cpu(), os.path.join(folder, "xyz.pt"))
        torch.save(self._opacity.detach().cpu(), os.path.join(folder, "opacity.pt"))
        torch.save(self._scaling.detach().cpu(), os.path.join(folder, "scaling.pt"))
        torch.save(self._rotation.detach().cpu(), os.path.join(folder, "rotation.pt"))
        torch.save(self._shs48_rows(None).cpu(), os.path.join(folder, "parameters.pt"))

    def save_ply(self, path):
        """3DGS .ply (base_gaussian_model.py:189-248 layout), written without plyfile."""
        import os
        from ..io_ply import save_ply
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self._flush_before_read()
        save_ply(path, self._xyz, self._shs48_rows(None), self._opacity, self._scaling, self._rotation)

    def save_sub_plys(self, path, n_split, split_size):
        """Split PLY output for models that do not fit host RAM in one piece
        (clm_offload/gaussian_model.py:292-360; scene/__init__.py:262-277): files
        <stem>_rk{i}_ws{n_split}.ply holding rows [i*split_size, (i+1)*split_size)."""
        import os
        from ..io_ply import save_ply
        assert path.endswith(".ply")
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        n = self._xyz.shape[0]
        self._flush_before_read()
        shs = self._shs48_rows(None)
        written = []
        for i in range(n_split):
            a, b = i * split_size, min((i + 1) * split_size, n)
            this_path = path[:-4] + "_rk" + str(i) + "_ws" + str(n_split) + ".ply"
            save_ply(this_path, self._xyz[a:b], shs[a:b], self._opacity[a:b], self._scaling[a:b],
                     self._rotation[a:b])
            written.append(this_path)
        return written

    def load_sub_plys(self, path, n_split, spatial_lr_scale=1.0):
        """Inverse of save_sub_plys: reassemble <stem>_rk{i}_ws{n_split}.ply in rank order."""
        import torch as _t
        from ..io_ply import load_ply
        parts = [load_ply(path[:-4] + "_rk" + str(i) + "_ws" + str(n_split) + ".ply") for i in range(n_split)]
        cat = {k: _t.cat([p[k] for p in parts], dim=0) for k in ("xyz", "shs48", "scaling", "rotation", "opacity")}
        self.create_from_tensors(cat["xyz"], cat["shs48"], cat["scaling"], cat["rotation"], cat["opacity"],
                                 spatial_lr_scale)
        self.active_sh_degree = self.max_sh_degree

    def load_ply(self, path, spatial_lr_scale=1.0):
        from ..io_ply import load_ply
        d = load_ply(path)
        self.create_from_tensors(d["xyz"], d["shs48"], d["scaling"], d["rotation"], d["opacity"],
                                 spatial_lr_scale)
        self.active_sh_degree = self.max_sh_degree

    def load_tensors(self, folder, spatial_lr_scale=1.0):
        import os
        ld = lambda n: torch.load(os.path.join(folder, n + ".pt"))
        self.create_from_tensors(ld("xyz"), ld("parameters"), ld("scaling"), ld("rotation"),
                                 ld("opacity"), spatial_lr_scale)
        self.active_sh_degree = self.max_sh_degree
