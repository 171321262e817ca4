"""GaussianModelCLMOffload (reference: strategies/clm_offload/gaussian_model.py:18-851).

xyz / opacity / scaling / rotation are GPU parameters; the 48 SH floats per Gaussian are ONE
row of a pre-allocated [capacity,48] buffer with a twin gradient buffer and twin Adam-state
buffers.  Where those buffers live is the MI355X re-sizing of the CLM split:

  sh_residency == "hbm"  (default): all four buffers in HBM -- 28 M Gaussians need 21.5 GB,
      102 M need 78 GB, both fit a 288 GB MI355X; the row-sparse Adam runs on the GPU.
  sh_residency == "host": pinned (hipHostMalloc, device-mapped) host buffers exactly like the
      reference; zero-copy gather/scatter kernels on the comm stream + host Adam thread.
The attribute surface (parameters_buffer, parameters_grad_buffer, _parameters, _features_dc/
_rest as split views, all_parameters() order, optimizer.gpu_adam/.cpu_adam/.columns_lr) is
the reference's, so the engine and train loop read the same in both modes.
"""
import os

import torch
from torch import nn

from ... import utils
from ...host import pinned_empty
from ...optimizer import UnifiedAdam
from ..base_gaussian_model import BaseGaussianModel

_ROW_BUFFERS = ("parameters_buffer", "parameters_grad_buffer", "_exp_avg_buffer", "_exp_avg_sq_buffer")


def _gather_f32(t, idx):
    """t[idx] (int64 row ids on the GPU) for a per-row float32 tensor [N] / [N,c], by the library's row mover: one launch
    per tensor (utils.take_rows goes through torch's advanced indexing in chunks of 2^23 indices -- its defect beyond 2^26
    -- with a temporary per chunk: ~8 launches per tensor, and a structural change moves fifteen such tensors)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and idx.dtype == torch.int64):
        return utils.take_rows(t, idx).contiguous()
    from ... import clm_kernels
    m = idx.numel()
    out = torch.empty((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if m:
        clm_kernels._rows("clmgs_rows_gather", out.view(m, -1), t.view(t.shape[0], -1), None, idx.contiguous(), 0)
    return out


def dp_range(n):
    from ... import dp
    return dp.owner_range(n)


class GaussianModelCLMOffload(BaseGaussianModel):
    _GPU_GROUPS = (("xyz", "_xyz"), ("opacity", "_opacity"), ("scaling", "_scaling"),
                   ("rotation", "_rotation"))

    # ------------------------------------------------------------- storage
    @property
    def sh_on_host(self):
        return getattr(self.args, "sh_residency", "hbm") == "host"

    def _alloc_rows(self, capacity, name=None):
        if self.sh_on_host:
            return pinned_empty((capacity, 48))
        n_cand = int(getattr(self.args, "placement_candidates", 1) or 1)
        if n_cand > 1:  # tables of >= 1 GB: the best-placed of a few allocations by a gather probe (placement.py)
            from ...placement import alloc_rows_placed
            return alloc_rows_placed(capacity, 48, n_cand, name)
        return torch.empty((capacity, 48), dtype=torch.float32, device="cuda")

    def _capacity_for(self, n):
        cap = int(getattr(self.args, "prealloc_capacity", -1))
        if cap <= 0:
            cap = n
        assert cap >= n, f"prealloc_capacity {cap} < number of gaussians {n}"
        return cap

    def _bind_rows(self, n):
        """(Re)create the [:n] views after the row count changed."""
        self._parameters = nn.Parameter(self.parameters_buffer[:n].requires_grad_(True))
        self._features_dc, self._features_rest = torch.split(self._parameters, [3, 45], dim=1)

    @property
    def get_features(self):
        return self._parameters.detach().reshape(-1, 16, 3)

    def create_from_tensors(self, xyz, shs48, scaling, rotation, opacity, spatial_lr_scale=1.0):
        # new tensors replace the model (first creation, load, restore): whatever the previous optimizer still had waiting
        # belonged to the tensors that are going away -- dropped with them, never replayed onto the new ones
        self._hbm_prefix = None  # (sh_hbm_budget_gb: the device copy of the old rows goes with them)
        if getattr(self, "_hwin_bufs", None) is not None and self._hwin_bufs.get("K", 0):
            torch.cuda.synchronize()
            self._hwin_bufs = None
        self.optimizer = None
        self._small_def, self._small_def_dirty = None, False
        self._lazy_dirty, self._sorted_tag = True, None
        self._mutations = getattr(self, "_mutations", 0) + 1
        self._small_key = None  # (the packed mirror of the small attributes is rebuilt from the new tensors)
        self.spatial_lr_scale = spatial_lr_scale
        n = xyz.shape[0]
        cap = self._capacity_for(n)
        self.parameters_buffer = self._alloc_rows(cap, "parameters")
        if not self.only_for_rendering:
            self.parameters_grad_buffer = self._alloc_rows(cap, "gradients")
            self.parameters_grad_buffer.zero_()
        self.parameters_buffer[:n].copy_(shs48.reshape(n, 48).float())
        self._xyz = nn.Parameter(xyz.float().cuda().contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scaling.float().cuda().contiguous().requires_grad_(True))
        self._rotation = nn.Parameter(rotation.float().cuda().contiguous().requires_grad_(True))
        self._opacity = nn.Parameter(opacity.float().cuda().contiguous().requires_grad_(True))
        self._bind_rows(n)
        self.max_radii2D = torch.zeros((n,), device="cuda")

    def all_parameters(self):
        # the engine relies on "first 4 are GPU" (clm_offload/engine.py:870)
        return [self._xyz, self._opacity, self._scaling, self._rotation, self._parameters]

    # ----------------------------------------------------------- optimiser
    def training_setup(self, training_args):
        # a model that has trained since its last flush (restore() / a second training_setup()): the waiting steps belong
        # to the OLD optimizer -- apply them before it is discarded, then start from clean deferral flags
        if getattr(self, "optimizer", None) is not None:
            if getattr(self, "_small_def", None) and not self._small_def_clean():
                self.flush_small()
            if self.lazy_rows and getattr(self, "_lazy_dirty", False):
                self.flush_lazy_rows()
            if getattr(self, "_hbm_prefix", None) is not None:  # its moments belong to the optimizer that is going away
                self.hbm_prefix_drop(writeback=True)
        self._small_def_dirty = False
        self._lazy_dirty = True
        self._mutations = getattr(self, "_mutations", 0) + 1
        self._sorted_tag = None
        self.percent_dense = training_args.percent_dense
        n = self.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device="cuda")
        self.denom = torch.zeros((n, 1), device="cuda")
        a = self.args
        l = [
            {"params": [self._xyz], "lr": training_args.position_lr_init * self.spatial_lr_scale * a.lr_scale_pos_and_scale, "name": "xyz"},
            {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": training_args.scaling_lr * a.lr_scale_pos_and_scale, "name": "scaling"},
            {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
            {"params": [self._parameters], "lr": training_args.feature_lr, "name": "parameters"},
        ]
        cap = self.parameters_buffer.shape[0]
        # camera-DP, locality exchange: only the OWNER of a row ever steps it (engine.py: catch_up_rows(own rows)), so
        # the two moment tables are held for the owned row range only -- 2 x 192 B x N / ranks instead of 2 x 192 B x N
        from ... import dp
        self._mom_sharded = bool((not self.sh_on_host) and (not a.sparse_adam) and getattr(a, "lazy_dense_adam", True)
                                 and dp.active() and getattr(a, "dp_locality", False)
                                 and getattr(a, "dp_shard_moments", True))
        self._mom_lo, self._mom_n = 0, n
        # ... and the small attributes (xyz / opacity / scaling / rotation): their dense Adam runs on the owned row range
        # only; a rank's copies of foreign rows go stale by a BOUNDED amount and are refreshed on demand (small_prepare)
        self._small_owner = bool((not self.sh_on_host) and (not a.sparse_adam) and getattr(a, "lazy_dense_adam", True)
                                 and dp.active() and getattr(a, "dp_locality", False)
                                 and getattr(a, "dp_small_owner", True) and getattr(a, "fused_front_end", True)
                                 and getattr(a, "packed_small", True))
        self._small_since, self._small_drift = 0, [0.0, 0.0]  # batches since all copies were current; drift bounds
        # single GPU: the dense Adam of the small attributes deferred per block of 256 Z-ordered rows (small_deferred)
        self._small_def = None
        if ((not self.sh_on_host) and (not a.sparse_adam) and getattr(a, "lazy_dense_adam", True) and not dp.active()
                and getattr(a, "deferred_small_adam", True) and getattr(a, "fused_front_end", True)
                and getattr(a, "packed_small", True) and getattr(a, "first_touch_grads", True)
                and not a.stop_update_param):
            self._small_def = {"hist": [], "blk_last": None, "n": -1}
        m_cap, m_n = cap, n
        if self._mom_sharded:
            lo, hi = dp.owner_range(n)
            self._mom_lo, m_n, m_cap = lo, hi - lo, self._moment_capacity(cap)
        self._exp_avg_buffer = self._alloc_rows(m_cap, "exp_avg").zero_()
        self._exp_avg_sq_buffer = self._alloc_rows(m_cap, "exp_avg_sq").zero_()
        self.optimizer = UnifiedAdam(
            l, [3, 45], [training_args.feature_lr, training_args.feature_lr / 20.0], lr=0.0,
            bias_correction=True, betas=(0.9, 0.999), eps=1e-15, fused=True, sparse=a.sparse_adam,
            state_tensors=(self._exp_avg_buffer[:m_n], self._exp_avg_sq_buffer[:m_n]))
        lr_scale = self._scale_groups_for_bsz(training_args)
        if training_args.lr_scale_mode in ("linear", "sqrt"):
            self.optimizer.columns_lr *= lr_scale
        # deferred dense Adam (HBM rows only): optimizer step each row is current as of
        self._row_last_step = self._row_g_step = None
        if (not self.sh_on_host) and (not a.sparse_adam) and getattr(a, "lazy_dense_adam", True):
            self._row_last_step = torch.zeros((cap,), dtype=torch.int32, device="cuda")
            # step whose gradient waits in parameters_grad_buffer[row] (applied at the row's next touch)
            self._row_g_step = torch.zeros((cap,), dtype=torch.int32, device="cuda")
        # host-resident rows: DEFERRED row optimizer (clmgs_host_rows_prepare).  Two host stamps per row:
        # step the row is current as of, and step whose gradient waits in parameters_grad_buffer (0: none)
        self._host_last_step = self._host_g_step = None
        self._host_grads_event = None  # the last batch's gradient rows have landed in host memory
        if self.sh_on_host:
            self._host_last_step = pinned_empty((cap,), dtype=torch.int32).zero_()
            self._host_g_step = pinned_empty((cap,), dtype=torch.int32).zero_()

    # ------------------------------------------- packed mirror of the small attributes
    def _small_tensors(self):
        return (self._xyz, self._opacity, self._scaling, self._rotation)

    def small_packed(self):
        """[N,12] mirror  xyz 3 | opacity 1 | scaling 3 | rotation 4 | pad  of the four GPU-resident
        parameter tensors for the per-camera gathers (one 48 B row instead of four scattered
        pieces).  The parameter tensors stay the source of truth: the mirror is rebuilt whenever
        one of them was replaced or modified by torch (data_ptr / version), and refreshed in place
        by the packed dense Adam (UnifiedAdam.gpu_step_packed)."""
        from ... import _lib
        key = tuple((p.data_ptr(), p._version) for p in self._small_tensors())
        n = self._xyz.shape[0]
        pk = getattr(self, "_small_pk", None)
        if pk is None or pk.shape[0] != n or getattr(self, "_small_key", None) != key:
            pk = self._row_mirror("pk", 12, n)
            t = [p.detach().contiguous() for p in self._small_tensors()]
            _lib.check(_lib.lib().clmgs_pack_small(_lib.stream(), n, *[_lib.dptr(x) for x in t], _lib.dptr(pk)))
            self._small_pk, self._small_key = pk, key
        return pk

    def small_grad(self):
        """Persistent zero-initialised [N,12] gradient table (the packed Adam zeroes what it eats)."""
        n = self._xyz.shape[0]
        g = getattr(self, "_small_gk", None)
        if g is None or g.shape[0] != n:
            g = self._small_gk = self._row_mirror("gk", 12, n)
            g.zero_()
        return g

    def _row_mirror(self, name, cols, n):
        """[:n] of a persistent [capacity, cols] table: the per-row side tables (packed mirror, packed gradients,
        statistics deltas) are sized like the row tables -- prealloc_capacity head room -- so that a densification, which
        changes n by a fraction of a percent, re-uses them.  (Allocated at exactly n rows they were re-allocated after
        every densification and the old blocks, a few rows too small for their successors, stayed cached: 2.7 GB of
        reserved memory more per densification at 28 M rows, bench.py --trainer-trace.)"""
        bufs = self.__dict__.setdefault("_mirrors", {})
        buf = bufs.get(name)
        if buf is None or buf.shape[0] < n or buf.device != self._xyz.device:
            cap = max(n, int(getattr(self, "parameters_buffer", torch.empty(0)).shape[0]))
            bufs[name] = buf = None  # (the old table is released before its successor is requested)
            buf = bufs[name] = torch.empty((cap, cols), dtype=torch.float32, device=self._xyz.device)
        return buf[:n]

    def invalidate_small_packed(self):
        self._small_key = None

    # ---- densification statistics: the fused path accumulates them in ONE [N,4] delta table
    # (max radius | grad accum | count | pad); the three model tensors are brought up to date
    # lazily, whenever anybody reads them (every 100 images for densification, capture, DP).
    def stats_delta(self):
        n = self._xyz.shape[0]
        d = getattr(self, "_stats_d", None)
        if d is None or d.shape[0] != n:
            self.merge_stats()
            d = self._stats_d = self._row_mirror("stats", 4, n)
            d.zero_()
        self._stats_dirty = True
        return d

    def merge_stats(self):
        d = getattr(self, "_stats_d", None)
        if d is None or not getattr(self, "_stats_dirty", False):
            return
        self._stats_dirty = False
        if self._max_radii2D.shape[0] == d.shape[0]:
            self._max_radii2D = torch.maximum(self._max_radii2D, d[:, 0])
            self._xyz_gradient_accum = self._xyz_gradient_accum + d[:, 1:2]
            self._denom = self._denom + d[:, 2:3]
        d.zero_()

    def _stat_get(self, name):
        self.merge_stats()
        return getattr(self, name)

    max_radii2D = property(lambda self: self._stat_get("_max_radii2D"),
                           lambda self, v: (self.merge_stats(), setattr(self, "_max_radii2D", v))[0])
    xyz_gradient_accum = property(lambda self: self._stat_get("_xyz_gradient_accum"),
                                  lambda self, v: (self.merge_stats(), setattr(self, "_xyz_gradient_accum", v))[0])
    denom = property(lambda self: self._stat_get("_denom"),
                     lambda self, v: (self.merge_stats(), setattr(self, "_denom", v))[0])

    # ---------------------------------------------------- camera-DP: small attributes at their owners
    @property
    def small_owner(self):
        """Camera-DP locality exchange, dp_small_owner (default): xyz / opacity / scaling / rotation of a row are
        stepped by the rank that owns the row (the packed dense Adam on the owned range, 1/ranks of the rows), and
        step F of the exchange (the owners' summed gradients all-gathered to every rank, 52 B x (ranks-1) per touched
        row) is gone.  What replaces it:
          * a rank's copies of foreign rows are STALE between refreshes, by at most Adam's step bound per batch
            (small_after_step) -- enough to cull conservatively (clmgs_visibility_candidates);
          * step S at the head of a batch (small_prepare): the candidates' current lines are fetched from their owners
            (48 B per candidate row, point to point), after which the exact visibility pass selects exactly the rows it
            selects on one rank and every row a camera renders from is current;
          * every `dp_small_refresh` batches (and at every flush_lazy_rows()) the owned ranges are all-gathered, which
            resets the drift bounds."""
        return bool(getattr(self, "_small_owner", False))

    def _small_params(self):
        return [self._xyz.data, self._opacity.data, self._scaling.data, self._rotation.data]

    def small_refresh(self, moments=False):
        """All ranks' copies of the small attributes <- their owners' (a collective); moments too for flush_lazy_rows
        (structural changes carry every row's moments along)."""
        from ... import dp
        n = self._xyz.shape[0]
        tables = self._small_params()
        if moments:
            for p in self._small_tensors():
                st = self.optimizer.gpu_adam.state.get(p, {})
                if "exp_avg" in st:
                    tables += [st["exp_avg"], st["exp_avg_sq"]]
        dp.owner_gather_dense(tables, n)
        self.invalidate_small_packed()  # the mirror is rebuilt from the tensors at its next use
        self._small_since, self._small_drift = 0, [0.0, 0.0]

    def small_prepare(self, cameras):
        """Step S, before the batch's visibility pass (all ranks, every batch).  See small_owner."""
        if self._small_since == 0:
            return  # every copy is current (start, refresh, flush): nothing can be stale
        from ... import dp
        from ...gsplat import visibility_candidates
        # refresh (the same decision on every rank: counters and learning rates are replicated) after dp_small_refresh
        # batches, or earlier once the scale bound has grown past dp_small_max_log_gain -- large global batches scale
        # the learning rates up, and a candidate set inflated by exp(1.6) costs more than the all-gather it avoids
        if (self._small_since >= int(getattr(self.args, "dp_small_refresh", 8))
                or self._small_drift[1] > float(getattr(self.args, "dp_small_max_log_gain", 0.7))):
            self.small_refresh()
            return
        n = self._xyz.shape[0]
        lo, hi = dp.owner_range(n)
        Ks = torch.stack([c.create_k_on_gpu() if getattr(c, "K", None) is None else c.K for c in cameras])
        viewmats = torch.stack([c.world_view_transform.transpose(0, 1) for c in cameras])
        d_xyz, d_ls = self._small_drift
        import math
        cand = visibility_candidates(self._xyz.data, self._scaling.data, viewmats, Ks, int(utils.get_img_width()),
                                     int(utils.get_img_height()), pos_margin=math.sqrt(3.0) * d_xyz * 1.001 + 1e-12,
                                     scale_gain=math.exp(d_ls) * 1.001, own_lo=lo, own_hi=hi)
        pk = self.small_packed()
        lines = dp.small_fetch(cand, n, pk)
        dp.small_scatter(cand, lines, pk, self._small_params())
        if os.environ.get("CLMGS_DP_DEBUG"):
            self._dbg_cand = cand
            print("DPDEBUG small_prepare rank %d since %d drift %s candidates %d of %d foreign rows" % (
                dp.rank(), self._small_since, self._small_drift, cand.numel(), n - (hi - lo)), flush=True)

    def small_after_step(self):
        """After the owners' Adam step of a batch: how far a foreign copy may now be off.  |m_hat| / sqrt(v_hat) of
        Adam is bounded by (1 - b1) / sqrt(1 - b2) / sqrt(1 - b1^2 / b2) (Cauchy-Schwarz on the two moment sums; the
        bias-correction ratio sqrt(1 - b2^t) / (1 - b1^t) is <= 1), i.e. 7.28 for (0.9, 0.999): no element moves by
        more than that times its learning rate in one step, whatever the gradients."""
        import math
        groups = {g["name"]: g for g in self.optimizer.gpu_adam.param_groups}
        b1, b2 = groups["xyz"]["betas"]
        c = (1.0 - b1) / math.sqrt(1.0 - b2) / math.sqrt(1.0 - b1 * b1 / b2) * 1.001
        self._small_drift[0] += c * float(groups["xyz"]["lr"])
        self._small_drift[1] += c * float(groups["scaling"]["lr"])
        self._small_since += 1

    # ---------------------------------------------------- single GPU: small attributes stepped per block, when needed
    @property
    def small_deferred(self):
        """Single GPU, dense optimizer (default): xyz / opacity / scaling / rotation are not stepped at the end of every
        batch.  A batch only RECORDS its step (learning rates, bias-correction index: small_defer_record); the gradient
        lines stay in the packed gradient table under their first-touch stamps.  At the head of the next batch
        (small_catch_up) every block of 256 consecutive rows that may hold a row visible in one of the batch's cameras
        -- by the cull's own conservative test on its stale values, dilated by what Adam can have moved them in the
        waiting steps -- replays its waiting steps exactly (clmgs_adam_small_deferred); blocks nobody looks at wait, at
        most clmgs_small_deferred_kmax() steps.  The exact visibility pass, the render and every gradient are the eager
        run's, bit for bit; what changes is that a batch streams p / m / v of the blocks near its cameras instead of
        all N rows.  Readers of the four tensors outside a batch go through flush_lazy_rows() (densification, opacity
        reset, evaluation, saving, capture: all do)."""
        return getattr(self, "_small_def", None) is not None

    def _small_def_tables(self, current_as_of=None):
        """blk_last: the step every block of 256 rows is current as of.  (Re)made -- all blocks current as of the newest
        recorded step, or of `current_as_of` when nothing is recorded yet (a fresh or restored model) -- whenever the
        row count has changed; every structural change flushes first, so nothing can be waiting then."""
        sd = self._small_def
        n = self._xyz.shape[0]
        if sd["blk_last"] is None or sd["n"] != n:
            assert self._small_def_clean(), "row count changed with small-attribute steps waiting"
            to = sd["hist"][-1][0] if sd["hist"] else int(current_as_of or 0)
            sd["blk_last"] = torch.full(((n + 255) // 256,), int(to), dtype=torch.int32, device=self._xyz.device)
            sd["n"] = n
        return sd["blk_last"]

    def _small_def_clean(self):
        return not getattr(self, "_small_def_dirty", False)

    def small_defer_record(self, step):
        """End of a batch: optimizer step `step` (the row optimizer's numbering = the stamps of the gradient lines) of the
        small attributes is RECORDED with the constants it must be replayed with."""
        import math
        sd = self._small_def
        opt = self.optimizer
        groups = {g["name"]: g for g in opt.gpu_adam.param_groups}
        order = [groups[n_] for n_ in ("xyz", "opacity", "scaling", "rotation")]
        cache = opt.__dict__.setdefault("_gpu_steps", {})
        steps, idx = [], None
        for g in order:
            p = g["params"][0]
            st = opt.gpu_adam.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if id(st["step"]) not in cache:
                cache.clear() if len(cache) > 64 else None
                cache[id(st["step"])] = int(st["step"].item())
            cache[id(st["step"])] += 1
            idx = cache[id(st["step"])] if idx is None else idx
            assert cache[id(st["step"])] == idx, "the four groups step together"
            steps.append(st["step"])
        torch._foreach_add_(steps, 1)  # torch-Adam's own step counters stay what an eager run leaves (capture / restore)
        opt.state = opt.gpu_adam.state | opt.cpu_adam.state
        if sd["hist"] and sd["hist"][-1][0] != step - 1:
            self.flush_small()  # a gap in the step numbering (a batch ran in another mode): nothing may span it
            sd["hist"] = []
            sd["blk_last"] = None
        self._small_def_tables(current_as_of=step - 1)  # (nothing recorded yet: every row is current up to the step before)
        b1, b2 = order[0]["betas"]
        c = (1.0 - b1) / math.sqrt(1.0 - b2) / math.sqrt(1.0 - b1 * b1 / b2) * 1.001  # Adam's step bound / lr (small_after_step)
        sd["hist"].append((int(step), [float(g["lr"]) for g in order], int(idx), c * float(order[0]["lr"]),
                           c * float(order[2]["lr"])))
        kmax = self._small_kmax()
        if len(sd["hist"]) > kmax:
            del sd["hist"][:len(sd["hist"]) - kmax]
        self._small_def_dirty = True

    def _small_kmax(self):
        k = getattr(self, "_small_kmax_v", None)
        if k is None:
            from ... import _lib
            k = self._small_kmax_v = int(_lib.lib().clmgs_small_deferred_kmax())
        return k

    def small_catch_up(self, cameras=None):
        """Head of a batch (cameras given) or a flush (None): see small_deferred."""
        import ctypes
        import math
        from ... import _lib
        sd = self._small_def
        if not sd["hist"] or (cameras is None and self._small_def_clean()):
            return None
        opt = self.optimizer
        groups = {g["name"]: g for g in opt.gpu_adam.param_groups}
        order = [groups[n_] for n_ in ("xyz", "opacity", "scaling", "rotation")]
        ps, ms, vs = [], [], []
        for g in order:
            p = g["params"][0]
            st = opt.gpu_adam.state[p]
            ps.append(p.data_ptr()); ms.append(st["exp_avg"].data_ptr()); vs.append(st["exp_avg_sq"].data_ptr())
        hist = sd["hist"][::-1]  # newest first
        nh = len(hist)
        lr = (ctypes.c_double * (4 * nh))(*[x for h in hist for x in h[1]])
        sidx = (ctypes.c_int32 * nh)(*[h[2] for h in hist])
        pm, sg_, dx, dl = [1e-12], [1.001], 0.0, 0.0
        for h in hist:
            dx += h[3]
            dl += h[4]
            pm.append(math.sqrt(3.0) * dx * 1.001 + 1e-12)
            sg_.append(math.exp(dl) * 1.001)
        pos_margin = (ctypes.c_float * (nh + 1))(*pm)
        scale_gain = (ctypes.c_float * (nh + 1))(*sg_)
        arr = lambda xs: (ctypes.c_void_p * 4)(*xs)
        pk, gk, blk = self.small_packed(), self.small_grad(), self._small_def_tables()
        b1, b2 = order[0]["betas"]
        C, vm, Ks, flags = 0, None, None, None
        if cameras is not None:
            C = len(cameras)
            flags = sd.get("blk_flag")
            if flags is None or flags.shape[0] != blk.shape[0]:
                flags = sd["blk_flag"] = torch.empty((blk.shape[0],), dtype=torch.uint8, device=blk.device)
            Ks = torch.stack([c.create_k_on_gpu() if getattr(c, "K", None) is None else c.K for c in cameras]).contiguous()
            vm = torch.stack([c.world_view_transform.transpose(0, 1) for c in cameras]).contiguous()
        _lib.check(_lib.lib().clmgs_adam_small_deferred(
            _lib.stream(), int(self._xyz.shape[0]), arr(ps), arr(ms), arr(vs), _lib.dptr(pk), _lib.dptr(gk),
            _lib.dptr(self._row_g_step, torch.int32), _lib.dptr(blk, torch.int32), int(hist[0][0]), nh, lr, sidx,
            pos_margin, scale_gain, float(b1), float(b2), float(order[0]["eps"]), 1.0 / float(self.args.bsz), C,
            _lib.dptr(vm, None, True), _lib.dptr(Ks, None, True), int(utils.get_img_width()), int(utils.get_img_height()),
            0.3, 0.01, 1e10, 0 if cameras is not None else 1, _lib.dptr(flags, torch.uint8, True)))
        if cameras is None:
            self._small_def_dirty = False
        return flags

    def flush_small(self):
        """Every block brought to the newest recorded step: the four tensors (and the mirror) are what an eager run holds."""
        if self.small_deferred:
            self.small_catch_up(None)

    # ---------------------------------------------------- deferred dense Adam
    @property
    def lazy_rows(self):
        return getattr(self, "_row_last_step", None) is not None

    @property
    def first_touch_grads(self):
        """The fused HBM engine STORES a row's SH gradient on its first touch of a step (stamping
        `_row_g_step`) instead of accumulating into a row its consumer cleared; the deferred row optimizer
        then leaves consumed rows as they are.  One policy per model: every producer of the gradient
        table must follow it, so it is a function of the (constant) engine options only."""
        from ... import dp
        a = self.args
        # camera-DP: the all-reduce / owner-computes exchanges sum whole row sets over the ranks (a rank's untouched
        # rows must be zeros), so they keep the clearing policy; the locality exchange moves stamped rows only
        dp_ok = (not dp.active()) or bool(getattr(a, "dp_locality", False))
        return bool(self.lazy_rows and getattr(a, "fused_front_end", True) and getattr(a, "first_touch_grads", True)
                    and dp_ok and not self.deferred_host_rows)

    @property
    def moments_sharded(self):
        """Row moments held for the owned row range only (camera-DP locality exchange, training_setup)."""
        return bool(getattr(self, "_mom_sharded", False))

    @staticmethod
    def _moment_capacity(cap):
        from ... import dp
        return -(-int(cap) // dp.world_size()) + 1  # the largest owner range of any n <= cap

    def catch_up_rows(self, rows=None, to_step=None):
        """Bring `rows` (None = all; with sharded moments: all OWNED rows -- the rest are replicas that their owners
        keep current) up to date with the zero-gradient Adam steps they skipped."""
        if not self.lazy_rows:
            return
        if self.moments_sharded and rows is None:
            lo = self._mom_lo
            hi = lo + self.optimizer.cpu_adam.state[self._parameters]["exp_avg"].shape[0]
            if hi <= lo:
                return
            rows = torch.arange(lo, hi, dtype=torch.int32, device=self._xyz.device)
        from ...clm_kernels import adam_catch_up
        opt = self.optimizer.cpu_adam
        g = opt.param_groups[0]
        p = self._parameters
        st = opt.state[p]
        to_step = opt.global_step if to_step is None else to_step
        if to_step <= 0:
            return
        col_lr = opt._col_lr(p.device)
        from ... import dp
        adam_catch_up(p.data, st["exp_avg"], st["exp_avg_sq"], self._row_last_step, rows, col_lr,
                      g["betas"][0], g["betas"][1], g["eps"], to_step, g["bias_correction"],
                      g=self.parameters_grad_buffer[:p.shape[0]], g_step=self._row_g_step,
                      grad_scale=1.0 / (self.args.bsz * dp.world_size()), keep_grad=self.first_touch_grads,
                      moment_row0=self._mom_lo if self.moments_sharded else 0)

    def flush_lazy_rows(self, exchange=True):
        """Apply every deferred row step that is still waiting.  exchange=False (owner-computes / locality
        camera-DP only): just the optimizer work -- every rank brings the rows it OWNS up to date -- without the
        all-gather that completes the replicas (bench.py times this form; evaluation / saving need the full one).  Under owner-computes / locality camera-DP this
        is a COLLECTIVE while replicas are partial (a batch ran since the last flush): call it on ALL ranks
        before any rank-0-only evaluation / save_ply / capture (trainer.py does); once nothing is dirty it is
        a local no-op, so the implicit calls inside those entry points are safe afterwards."""
        if self.deferred_host_rows:
            self.host_rows_prepare(None, None)
            return
        self.flush_small()  # (single GPU, small attributes stepped per block: everything waiting is applied)
        from ... import dp
        if (not self.lazy_rows) and dp.active() and getattr(self.args, "dp_locality", False) \
                and getattr(self.args, "sparse_adam", False):
            # sparse_adam locality: the owners' rows are always current (eager optimizer); completing the replicas
            # is an all-gather of parameters and moments
            if getattr(self, "_owner_dirty", True) and exchange:
                self._owner_dirty = False
                n = self._parameters.shape[0]
                st = self.optimizer.cpu_adam.state[self._parameters]
                dp.owner_gather_dense([self._parameters.data, st["exp_avg"], st["exp_avg_sq"]], n)
            return
        if self.lazy_rows and dp.active() and (getattr(self.args, "dp_owner_computes", False)
                                               or getattr(self.args, "dp_locality", False)):
            # owner-computes camera-DP: every rank brings the rows it OWNS up to date, then all ranks
            # exchange parameters, moments and stamps, after which each replica is complete and current
            if not getattr(self, "_owner_dirty", True):
                return
            n = self._parameters.shape[0]
            lo, hi = dp.owner_range(n)
            if hi > lo:
                self.catch_up_rows(torch.arange(lo, hi, dtype=torch.int32, device=self._xyz.device))
            if not exchange:
                return
            self._owner_dirty = False
            st = self.optimizer.cpu_adam.state[self._parameters]
            tables = [self._parameters.data]
            if not self.moments_sharded:  # sharded: a row's moments never leave its owner
                tables += [st["exp_avg"], st["exp_avg_sq"]]
            if not self.first_touch_grads:  # clearing policy: the owners' consumed (zeroed) rows replace the partial sums
                tables.append(self.parameters_grad_buffer[:n])
            dp.owner_gather_dense(tables, n)
            if self.small_owner:
                self.small_refresh(moments=True)
            self._row_last_step[:n] = self.optimizer.cpu_adam.global_step
            self._row_g_step[:n] = 0
            return
        # single rank: a pass over all N rows' stamps (3.5 ms at 28 M) -- skipped when no batch has run since the last
        # flush (a densification calls this six times: clone, split, prune, re-sort, ...; `_lazy_dirty` is set by the engine)
        if not getattr(self, "_lazy_dirty", True):
            return
        self.catch_up_rows(None)
        self._lazy_dirty = False

    # ---------------------------------------------------- deferred host row optimizer
    @property
    def deferred_host_rows(self):
        return getattr(self, "_host_last_step", None) is not None

    def drop_host_speculation(self):
        """Forget the rows the engine staged for a hinted batch that is not coming (model resized / flushed /
        evaluated / different cameras): they were stamped as expecting that batch's gradient, which will never
        land (clm_offload/engine.py speculative prefetch)."""
        for attr in ("_host_spec", "_hwin_spec"):  # (batch-wide staging / per-camera windows: engine.py / host_window.py)
            sp = getattr(self, attr, None)
            setattr(self, attr, None)
            if sp is None:
                continue
            sp["thread"].join()
            ev = sp.get("event")
            if ev is not None:
                ev.synchronize()  # the thread's last hipMemcpyAsync into the staging table has landed
            if sp["n"]:
                self._host_g_step[sp["rows_h"][:sp["n"]].long()] = 0

    # ---------------------------------------------------- HBM-resident prefix of the host rows (sh_hbm_budget_gb)
    # sh_residency="host" with a budget: rows [0, K) of the (Z-ordered) SH table, their two moments and their gradient rows
    # LIVE in HBM while batches run -- 4 x 192 B per row, K = budget / 768 B -- and are stepped there by the deferred row
    # optimizer of the HBM engine (clmgs_adam_catch_up, its own stamps); only rows >= K cross the host link and are stepped
    # by the host pool.  The host tables stay the model's storage: every reader outside a batch goes through
    # flush_lazy_rows() / host_rows_prepare(), which write the prefix back first (hbm_prefix_writeback), and every
    # structural change (append / prune / permute / a new optimizer) forgets the device copy (hbm_prefix_drop) after that
    # flush; the next batch loads it again (768 B x K over the link: ~0.2 s at 14 M rows).
    def hbm_prefix_rows(self):
        gb = float(getattr(self.args, "sh_hbm_budget_gb", 0.0) or 0.0)
        if gb <= 0.0 or not self.sh_on_host:
            return 0
        return int(min(self._xyz.shape[0], gb * 1e9 // (4 * 48 * 4)))

    def hbm_prefix_ensure(self):
        """-> the live prefix state (dict: K, m, v, last_step, g_step, dirty, fill) or None without a budget.  A prefix
        that is not loaded yet is created here from the host tables, which are brought up to date first; its parameter
        and gradient rows are the first K rows of the staging tables of host_window.py, which fills them (`fill`)."""
        K = self.hbm_prefix_rows()
        px = getattr(self, "_hbm_prefix", None)
        if px is not None and (px["K"] != K or px["N"] != self._xyz.shape[0]):
            self.hbm_prefix_drop(writeback=True)
            px = None
        if K == 0:
            return None
        if px is None:
            assert self.deferred_host_rows
            self.flush_lazy_rows()  # every host row current, nothing waiting, no speculation outstanding
            opt = self.optimizer.cpu_adam
            st = opt.state[self._parameters]
            dev = self._xyz.device
            px = self._hbm_prefix = dict(
                K=K, N=int(self._xyz.shape[0]), m=st["exp_avg"][:K].to(dev), v=st["exp_avg_sq"][:K].to(dev),
                last_step=torch.full((K,), int(opt.global_step), dtype=torch.int32, device=dev),
                g_step=torch.zeros((K,), dtype=torch.int32, device=dev), dirty=False, fill=True)
        return px

    def _hbm_prefix_tables(self):
        hb = getattr(self, "_hwin_bufs", None)
        px = self._hbm_prefix
        assert hb is not None and hb.get("K", 0) == px["K"] and not px["fill"], "the prefix rows are not in the staging tables"
        return hb["pt"][:px["K"]], hb["gt"][:px["K"]]

    def hbm_prefix_catch_up(self, rows, to_step):
        """Deferred row optimizer on prefix rows (int32 device list, None = all K): the waiting gradient of a row at its
        own step, then the zero-gradient steps it skipped, up to `to_step`; consumed gradient rows are cleared."""
        from ...clm_kernels import adam_catch_up
        px = self._hbm_prefix
        if to_step <= 0 or (rows is not None and rows.numel() == 0):
            return
        p, gr = self._hbm_prefix_tables()
        opt = self.optimizer.cpu_adam
        g = opt.param_groups[0]
        adam_catch_up(p, px["m"], px["v"], px["last_step"], rows, opt._col_lr(p.device), g["betas"][0], g["betas"][1],
                      g["eps"], int(to_step), g["bias_correction"], g=gr, g_step=px["g_step"],
                      grad_scale=1.0 / float(self.args.bsz), keep_grad=False)

    def hbm_prefix_writeback(self):
        """Host tables [0, K) <- the prefix, brought up to the optimizer's current step (no-op unless a batch ran since)."""
        px = getattr(self, "_hbm_prefix", None)
        if px is None or not px["dirty"]:
            return
        K = px["K"]
        opt = self.optimizer.cpu_adam
        step = int(opt.global_step)
        if not self.args.sparse_adam:
            self.hbm_prefix_catch_up(None, step)
        p, _ = self._hbm_prefix_tables()
        st = opt.state[self._parameters]
        self._parameters.data[:K].copy_(p)          # (pinned destination, blocking copies: the host pass that follows
        st["exp_avg"][:K].copy_(px["m"])            #  reads these rows)
        st["exp_avg_sq"][:K].copy_(px["v"])
        self._host_last_step[:K] = step
        self._host_g_step[:K] = 0
        px["dirty"] = False

    def hbm_prefix_drop(self, writeback=False):
        """Forget the device copy of the prefix (structural change, new optimizer).  The callers have flushed -- which
        wrote it back -- before they changed anything; writeback=True: do that here (the budget itself changed)."""
        px = getattr(self, "_hbm_prefix", None)
        if px is None:
            return
        if writeback:
            self.hbm_prefix_writeback()
        assert not px["dirty"], "the HBM-resident prefix was dropped with row steps that never reached the host tables"
        self._hbm_prefix = None
        hb = getattr(self, "_hwin_bufs", None)
        if hb is not None and hb.get("K", 0):
            torch.cuda.synchronize()  # (side-stream copies into the tables: see host_window._buffers)
            self._hwin_bufs = None

    def host_rows_prepare(self, rows_host, stage_host, to_step=None, next_g_step=0, n_rows=None, sync_grads=True):
        """Bring host rows (int32 pinned/CPU row list, None = all) up to `to_step` (default: the optimizer's
        current step): waiting gradients are applied at their own step, skipped zero-gradient steps are
        replayed; copies the current parameter rows into stage_host[k] when given.  Blocks until the rows
        are done (host threads, GIL released).  sync_grads=False: the engine's feeder threads, which have
        ordered themselves after the gradient hand-back already and must not touch the model's bookkeeping."""
        import ctypes
        from ... import _lib
        assert self.deferred_host_rows
        if sync_grads:
            self.drop_host_speculation()
            if self._host_grads_event is not None:  # the waiting gradients must have landed
                self._host_grads_event.synchronize()
                self._host_grads_event = None
            self.hbm_prefix_writeback()  # (sh_hbm_budget_gb: rows [0, K) are stepped in HBM; the host copy is read next)
        opt = self.optimizer.cpu_adam
        g = opt.param_groups[0]
        p = self._parameters
        st = opt.state[p]
        n = p.shape[0]
        to_step = opt.global_step if to_step is None else int(to_step)
        col_lr = opt._col_lr(torch.device("cpu")).contiguous()
        P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        if rows_host is not None:
            assert rows_host.dtype == torch.int32 and not rows_host.is_cuda and rows_host.is_contiguous()
            n_rows = rows_host.numel() if n_rows is None else int(n_rows)
        else:
            n_rows = n
        _lib.check(_lib.lib().clmgs_host_rows_prepare(
            P(p.data), P(self.parameters_grad_buffer), P(st["exp_avg"]), P(st["exp_avg_sq"]),
            P(self._host_last_step), P(self._host_g_step), P(rows_host), n_rows, p.shape[1], P(col_lr),
            float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), to_step, int(next_g_step),
            int(g["bias_correction"]), 1.0 / float(self.args.bsz), 256, P(stage_host),
            int(bool(self.args.sparse_adam))))

    def _rebind_row_state(self, n):
        """Point the row optimizer at the [:n] views after append / prune."""
        self.hbm_prefix_drop()  # (rows moved / were added / removed: the callers flushed before they did it)
        old = self._parameters
        opt = self.optimizer.cpu_adam
        st = opt.state.pop(old, None)
        self._bind_rows(n)
        opt.param_groups[0]["params"][0] = self._parameters
        if st is None:
            st = {"step": opt.global_step}
        m_n = n
        if self.moments_sharded:
            assert self._mom_n == n, "sharded moments were not redistributed for the new row count"
            m_n = dp_range(n)[1] - self._mom_lo
        st["exp_avg"] = self._exp_avg_buffer[:m_n]
        st["exp_avg_sq"] = self._exp_avg_sq_buffer[:m_n]
        opt.state[self._parameters] = st
        self.optimizer.state = self.optimizer.gpu_adam.state | self.optimizer.cpu_adam.state

    def _replace_gpu(self, name, attr, new_tensor, state_fn):
        self._mutations = getattr(self, "_mutations", 0) + 1  # (spatial_sort's "already sorted" shortcut ends here)
        opt = self.optimizer.gpu_adam
        for g in opt.param_groups:
            if g["name"] != name:
                continue
            old = g["params"][0]
            st = opt.state.get(old, None)
            if st is not None:
                for k in ("exp_avg", "exp_avg_sq"):
                    if k in st:
                        st[k] = state_fn(st[k])
                del opt.state[old]
            g["params"][0] = nn.Parameter(new_tensor.requires_grad_(True))
            if st is not None:
                opt.state[g["params"][0]] = st
            setattr(self, attr, g["params"][0])
            return
        raise KeyError(name)

    def _full_row_buffers(self):
        """The [capacity,48] tables indexed by global row id on this rank."""
        return _ROW_BUFFERS[:2] if self.moments_sharded else _ROW_BUFFERS

    def _redistribute_moments(self, idx, n_new):
        """Sharded row moments after a structural change: new row j <- old row idx[j] (int64 on the GPU, the same on
        every rank; -1 = a new row, zero moments).  Owner ranges are index ranges of the CURRENT row count, so an
        append / prune / re-sort moves range borders: every rank sends the rows of its old shard that landed in
        another rank's new range (one all_to_all of [rows, 96] lines; the bulk stays local) and rebuilds its shard."""
        import torch.distributed as dist
        from ... import dp
        G, r = dp.world_size(), dp.rank()
        n_old = self._mom_n
        dev = self._exp_avg_buffer.device
        old_cuts = [(q * n_old) // G for q in range(G + 1)]
        new_cuts = [(q * n_new) // G for q in range(G + 1)]
        lo, hi = old_cuts[r], old_cuts[r + 1]
        lo2, hi2 = new_cuts[r], new_cuts[r + 1]
        idx = idx.to(dev)
        assert idx.numel() == n_new and idx.dtype == torch.int64
        send_rows = []
        for d in range(G):  # in d's new-row order: the receiver lists its positions in the same order
            sel = idx[new_cuts[d]:new_cuts[d + 1]]
            send_rows.append(sel[(sel >= lo) & (sel < hi)] - lo)
        mine = idx[lo2:hi2]
        recv_pos = [torch.nonzero((mine >= old_cuts[q]) & (mine < old_cuts[q + 1])).flatten() for q in range(G)]
        send_n = [int(t.numel()) for t in send_rows]
        recv_n = [int(t.numel()) for t in recv_pos]
        rows = torch.cat(send_rows)
        send = torch.cat((self._exp_avg_buffer[rows], self._exp_avg_sq_buffer[rows]), dim=1)
        recv = torch.empty((sum(recv_n), 96), dtype=torch.float32, device=dev)
        dist.all_to_all_single(recv, send, output_split_sizes=recv_n, input_split_sizes=send_n)
        dp._count("moments_all_to_all", (send.shape[0] - send_n[r]) * 96 * 4)
        m_cap = max(self._exp_avg_buffer.shape[0], self._moment_capacity(self.parameters_buffer.shape[0]))
        del send
        self._exp_avg_buffer = self._exp_avg_sq_buffer = None
        new_m = torch.zeros((m_cap, 48), dtype=torch.float32, device=dev)
        new_v = torch.zeros((m_cap, 48), dtype=torch.float32, device=dev)
        pos = torch.cat(recv_pos)
        if pos.numel():
            new_m[pos] = recv[:, :48]
            new_v[pos] = recv[:, 48:]
        self._exp_avg_buffer, self._exp_avg_sq_buffer = new_m, new_v
        self._mom_lo, self._mom_n = lo2, n_new

    def row_moments_full(self):
        """(exp_avg, exp_avg_sq) of ALL rows on this rank -- with sharded moments a COLLECTIVE (all ranks call it:
        capture()), otherwise the optimizer's own tensors."""
        st = self.optimizer.cpu_adam.state[self._parameters]
        if not self.moments_sharded:
            return st["exp_avg"], st["exp_avg_sq"]
        from ... import dp
        n = self._parameters.shape[0]
        lo, hi = dp.owner_range(n)
        out = []
        for t in (st["exp_avg"], st["exp_avg_sq"]):
            full = torch.zeros((n, 48), dtype=torch.float32, device=t.device)
            full[lo:hi] = t
            dp.owner_gather_dense([full], n)
            out.append(full)
        return out[0], out[1]

    def _grow(self, need):
        cap = self.parameters_buffer.shape[0]
        if need <= cap:
            return
        new_cap = max(need, int(cap * 1.5))
        n = self._parameters.shape[0]
        for attr in self._full_row_buffers():
            old = getattr(self, attr)
            new = self._alloc_rows(new_cap, attr)
            new[:n].copy_(old[:n])
            if attr != "parameters_buffer":
                new[n:].zero_()
            setattr(self, attr, new)
        # (sharded moment tables are re-made at their new size by _redistribute_moments)

    def _append_rows(self, new):
        k = new["xyz"].shape[0]
        n = self._parameters.shape[0]
        self.flush_lazy_rows()
        self._grow(n + k)
        if self.deferred_host_rows:
            cap2 = self.parameters_buffer.shape[0]
            if self._host_last_step.shape[0] < cap2:
                for attr in ("_host_last_step", "_host_g_step"):
                    grown = pinned_empty((cap2,), dtype=torch.int32).zero_()
                    grown[:n] = getattr(self, attr)[:n]
                    setattr(self, attr, grown)
            self._host_last_step[n:n + k] = self.optimizer.cpu_adam.global_step
            self._host_g_step[n:n + k] = 0
        if self.lazy_rows:
            if self._row_last_step.shape[0] < self.parameters_buffer.shape[0]:
                grown = torch.zeros((self.parameters_buffer.shape[0],), dtype=torch.int32, device="cuda")
                grown[:n] = self._row_last_step[:n]
                self._row_last_step = grown
                self._row_g_step = torch.zeros_like(grown)  # nothing waits after the flush above
            self._row_g_step[n:n + k] = 0
            self._row_last_step[n:n + k] = self.optimizer.cpu_adam.global_step
        self.parameters_buffer[n:n + k].copy_(new["shs48"])
        self.parameters_grad_buffer[n:n + k].zero_()
        if self.moments_sharded:
            dev = self.parameters_buffer.device
            self._redistribute_moments(torch.cat((torch.arange(n, dtype=torch.int64, device=dev),
                                                  torch.full((k,), -1, dtype=torch.int64, device=dev))), n + k)
        else:
            self._exp_avg_buffer[n:n + k].zero_()
            self._exp_avg_sq_buffer[n:n + k].zero_()
        ext = {"xyz": new["xyz"], "opacity": new["opacity"], "scaling": new["scaling"],
               "rotation": new["rotation"]}
        for name, attr in self._GPU_GROUPS:
            cur = getattr(self, attr).detach()
            e = ext[name]
            self._replace_gpu(name, attr, torch.cat((cur, e), dim=0),
                              lambda s, e=e: torch.cat((s, torch.zeros_like(e)), dim=0))
        self._rebind_row_state(n + k)

    def _regather_row_tables(self, idx, m):
        """Every [capacity,48] row table <- its rows idx[0..m) (int64, on the GPU), for pruning (ascending kept rows) and
        re-ordering: one HBM pass per table with the library's row mover (clmgs_rows_gather: 64-bit row arithmetic, 16 B
        per lane), and NO scratch table (round 5).  The gradient table holds nothing that is still needed when a
        structural change happens -- every caller has applied the deferred steps (flush_lazy_rows: no gradient waits, all
        stamps are reset), the eager modes leave it zeroed after every batch -- so it is the destination of the first
        gather; every table then lands in the one vacated before it, and the last vacated table, zeroed, is the new
        gradient table: 3 gathers + one clear instead of 4 gathers through a fifth table of the model's capacity
        (5.6 GB at 28 M rows, which was the training run's peak)."""
        from ... import clm_kernels
        idx = idx.contiguous()
        free = self.parameters_grad_buffer
        if __debug__ and self.lazy_rows and os.environ.get("CLMGS_DEBUG_CHECKS") == "1":
            n_ = int(self._parameters.shape[0])  # (a device read: debug runs only)
            assert not bool((self._row_g_step[:n_] > self._row_last_step[:n_]).any()), "a gradient row still waits"
        for attr in self._full_row_buffers():
            if attr == "parameters_grad_buffer":
                continue
            buf = getattr(self, attr)
            if buf.shape != free.shape:
                raise RuntimeError(f"row tables of different capacity ({attr}: {tuple(buf.shape)} vs {tuple(free.shape)}): "
                                   "the table rotation of _regather_row_tables needs equal capacities")
            if m:
                clm_kernels._rows("clmgs_rows_gather", free[:m], buf, None, idx, 0)
            setattr(self, attr, free)
            free = buf
        free.zero_()
        self.parameters_grad_buffer = free
        if self.moments_sharded:
            self._redistribute_moments(idx[:m], m)

    def drop_row_scratch(self):
        """(Rounds 3-4 kept a scratch table of the model's capacity for _regather_row_tables; there is none any more.)"""
        self._row_scratch = None

    def prune_points(self, mask, resort=False):
        """Remove the rows of `mask`.  resort=True (HBM rows): the kept rows are ALSO put in Z-order of their positions by
        the same single pass over every table -- exactly prune_points(mask) followed by spatial_sort() (same keys, same
        stable sort), for the price of one compaction instead of a compaction and a permutation."""
        keep = ~mask
        n = self._parameters.shape[0]
        m = int(keep.sum())
        self.flush_lazy_rows()  # afterwards every surviving row carries the same step stamp
        if self.lazy_rows:
            self._row_last_step[:m] = self.optimizer.cpu_adam.global_step
            self._row_g_step[:m] = 0
        if self.deferred_host_rows:
            self._host_last_step[:m] = self.optimizer.cpu_adam.global_step
            self._host_g_step[:m] = 0
        if self.sh_on_host:
            keep_rows = keep.cpu()
            for attr in _ROW_BUFFERS:
                buf = getattr(self, attr)
                buf[:m].copy_(utils.select_rows(buf[:n], keep_rows))  # in-place compaction (clm/gaussian_model.py:566-570)
            pick = lambda t: utils.select_rows(t, keep).contiguous()
        else:
            idx = torch.nonzero(keep).flatten()
            if resort and m:
                idx = utils.take_rows(idx, utils.morton_order(_gather_f32(self._xyz.detach(), idx)))
            self._regather_row_tables(idx, m)
            pick = lambda t: _gather_f32(t, idx)
        for name, attr in self._GPU_GROUPS:
            cur = getattr(self, attr).detach()
            self._replace_gpu(name, attr, pick(cur), pick)
        self._rebind_row_state(m)
        self.xyz_gradient_accum = pick(self.xyz_gradient_accum)
        self.denom = pick(self.denom)
        self.max_radii2D = pick(self.max_radii2D)
        if self.sh_on_host and resort:
            self.spatial_sort()
        self.invalidate_small_packed()
        self._mutations = getattr(self, "_mutations", 0) + 1
        self._sorted_tag = (self._mutations, m) if resort else None

    def spatial_sort(self):
        # The tag is (model-mutation counter, rows): every path that moves a position or replaces a table bumps the counter
        # (training_setup, prune / append / permute, every engine batch, load) -- an address can be reused by the caching
        # allocator for a same-sized tensor, a counter cannot.
        if getattr(self, "_sorted_tag", None) == (getattr(self, "_mutations", 0), self._xyz.shape[0]):
            return  # prune_points(resort=True) has just left the rows in this very order
        super().spatial_sort()
        self._mutations = getattr(self, "_mutations", 0) + 1

    def permute_rows(self, order):
        n = self._parameters.shape[0]
        order = order.to(self._xyz.device)
        assert order.numel() == n
        self.flush_lazy_rows()  # afterwards every row carries the same step stamp and no waiting gradient
        if self.sh_on_host or getattr(self, "parameters_grad_buffer", None) is None or self.optimizer is None:
            order_rows = order.cpu() if self.sh_on_host else order
            for attr in _ROW_BUFFERS:
                buf = getattr(self, attr, None)
                if buf is None or buf.numel() == 0:
                    continue
                buf[:n].copy_(utils.gather_rows(buf[:n], order_rows))  # out of place, one table at a time
        else:
            self._regather_row_tables(order.to(torch.int64), n)
        order64 = order.to(torch.int64)
        pick = lambda t: _gather_f32(t.contiguous(), order64)  # (host / non-float tensors: utils.take_rows)
        for name, attr in self._GPU_GROUPS:
            cur = getattr(self, attr).detach()
            if self.optimizer is not None:
                self._replace_gpu(name, attr, pick(cur), pick)
            else:
                setattr(self, attr, nn.Parameter(pick(cur).requires_grad_(True)))
        if self.optimizer is not None:
            self._rebind_row_state(n)
            self.xyz_gradient_accum = pick(self.xyz_gradient_accum)
            self.denom = pick(self.denom)
        else:
            self._bind_rows(n)
        self.max_radii2D = pick(self.max_radii2D)
        self.invalidate_small_packed()

    def _shs48_rows(self, mask):
        self.flush_lazy_rows()
        p = self._parameters.detach()
        if mask is None:
            return p.clone() if p.is_cuda else p.cuda()
        if self.sh_on_host:
            return p[mask.cpu()].cuda()
        return utils.select_rows(p, mask)

    def reset_opacity(self):
        if self.small_owner:
            self.flush_lazy_rows()  # every copy current (and identical on all ranks) before all of them are rewritten
        self.flush_small()  # (single GPU, small attributes stepped per block: the opacities must be current)
        new = utils.inverse_sigmoid(torch.min(self.get_opacity.detach(), torch.ones_like(self._opacity) * 0.01))
        self._replace_gpu("opacity", "_opacity", new, lambda s: torch.zeros_like(s))
