"""BUILD-CONTAINER ONLY.  Runs the reference's OWN engine / model / densification / argument code
(/root/reference, imported -- never copied) on the CPU under tests/golden/ref_harness.py with
oracle/gs_oracle.py plugged in as `gsplat` / `clm_kernels`, and commits what it produced:

    python tests/golden/make_engine_golden.py

  engine_no_offload.npz   strategies/no_offload/engine.py:104-177 baseline_accumGrads_impl on a seeded
                          CI-size scene: per-camera losses, the six accumulated .grad tensors and the
                          densification statistics BEFORE any optimizer step; the optimizer groups'
                          lr / eps / betas after the reference's training_setup (bsz scaling,
                          no_offload/gaussian_model.py:223-246); then 3 batches of the train.py:533-578
                          epilogue (grad /= bsz, Adam, zero_grad) with moving cameras: parameters and
                          both Adam moments.
  engine_filters.npz      strategies/base_engine.py:18-76 calculate_filters on the same scene.
  engine_densify.npz      densification.py:5-56 gsplat_densification -> base_gaussian_model.py:364-388
                          densify_and_prune (+ clone / split / prune / postfix of the no_offload model)
                          on the statistics of the batch above: masks, the normal draws of the split,
                          every tensor and both Adam moments after the surgery, and the schedule
                          (which iterations densify / reset opacity).
  engine_naive_offload.npz strategies/naive_offload/engine.py:48-357 naive_offload_train_one_batch, 3 batches
                          dense and 3 with sparse_adam (cpu_adam.CPUAdam stood in by oracle/clm_oracle.py):
                          losses, parameters + both moments of the six groups, statistics, eval image.
  arguments_defaults.json arguments/__init__.py: parser.parse_args([]) of the six ParamGroups.

The GPU tests compare the HIP engines with these files; the CPU tests re-derive them from the oracle
composition (tests/test_golden_engine.py).
"""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as RH  # noqa: E402

N, W, H, BSZ, N_BATCHES = 2000, 96, 64, 4, 3
SEED = 20250927


def ci_scene():
    """Seeded aerial slab + nadir cameras (same generator family as clm_gs_amd/synthetic.py, CPU RNG)."""
    g = torch.Generator().manual_seed(SEED)
    L = 0.5 * math.sqrt(N)
    xyz = torch.rand((N, 3), generator=g)
    xyz[:, 0] = (xyz[:, 0] * 2 - 1) * L
    xyz[:, 1] = (xyz[:, 1] * 2 - 1) * L
    xyz[:, 2] = xyz[:, 2] * 0.1 * L
    scaling = torch.randn((N, 3), generator=g) * 0.4 + math.log(0.7)
    rotation = torch.randn((N, 4), generator=g)
    opacity = torch.randn((N, 1), generator=g) * 1.5
    shs = torch.randn((N, 16, 3), generator=g) * 0.1
    shs[:, 0, :] = torch.randn((N, 3), generator=g)
    n_cams = BSZ * N_BATCHES
    f = 0.8 * W
    fovx, fovy = 2 * math.atan(W / (2 * f)), 2 * math.atan(H / (2 * f))
    h = math.sqrt(0.35 * (2 * L) ** 2 * f * f / (W * H))
    w2cs = []
    for i in range(n_cams):
        cx = float((torch.rand((), generator=g) * 2 - 1) * 0.45 * L)
        cy = float((torch.rand((), generator=g) * 2 - 1) * 0.45 * L)
        ang = float((torch.rand((), generator=g) * 2 - 1) * 0.15)  # small tilt about x: exercises the full R
        R0 = torch.tensor([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])
        Rx = torch.tensor([[1.0, 0, 0], [0, math.cos(ang), -math.sin(ang)], [0, math.sin(ang), math.cos(ang)]])
        R = Rx @ R0
        C = torch.tensor([cx, cy, 0.1 * L + h])
        w2c = torch.eye(4)
        w2c[:3, :3] = R
        w2c[:3, 3] = -R @ C
        w2cs.append(w2c)
    gt = (torch.rand((n_cams, 3, H, W), generator=g) * 255).to(torch.uint8)
    return dict(xyz=xyz, scaling=scaling, rotation=rotation, opacity=opacity, shs=shs, w2c=torch.stack(w2cs),
                fovx=fovx, fovy=fovy, gt=gt, extent=float(L))


def make_ref_model(sc, args, cls):
    m = cls(3)
    m.spatial_lr_scale = 1.0
    m._xyz = torch.nn.Parameter(sc["xyz"].clone().requires_grad_(True))
    m._features_dc = torch.nn.Parameter(sc["shs"][:, :1, :].clone().contiguous().requires_grad_(True))
    m._features_rest = torch.nn.Parameter(sc["shs"][:, 1:, :].clone().contiguous().requires_grad_(True))
    m._scaling = torch.nn.Parameter(sc["scaling"].clone().requires_grad_(True))
    m._rotation = torch.nn.Parameter(sc["rotation"].clone().requires_grad_(True))
    m._opacity = torch.nn.Parameter(sc["opacity"].clone().requires_grad_(True))
    m.max_radii2D = torch.zeros((sc["xyz"].shape[0],))
    m.active_sh_degree = 3
    m.training_setup(args)
    return m


def np_(t):
    return t.detach().cpu().numpy().copy()  # copy: the engines keep modifying these tensors in place


def make_ref_clm_model(sc, args, cls, mode):
    n = sc["xyz"].shape[0]
    m = cls(3)
    m.spatial_lr_scale = 1.0
    cap = args.prealloc_capacity
    m.parameters_buffer = torch.zeros((cap, 48))
    m.parameters_grad_buffer = torch.zeros((cap, 48))
    m.parameters_buffer[:n] = sc["shs"].reshape(n, 48)
    m._xyz = torch.nn.Parameter(sc["xyz"].clone().requires_grad_(True))
    m._scaling = torch.nn.Parameter(sc["scaling"].clone().requires_grad_(True))
    m._rotation = torch.nn.Parameter(sc["rotation"].clone().requires_grad_(True))
    m._opacity = torch.nn.Parameter(sc["opacity"].clone().requires_grad_(True))
    m._parameters = torch.nn.Parameter(m.parameters_buffer[:n].requires_grad_(True))
    m._features_dc, m._features_rest = torch.split(m._parameters, [3, 45], dim=1)
    m.max_radii2D = torch.zeros((n,))
    m.active_sh_degree = 3
    mode.claim_cuda = True  # optimizer.py:113-119 asserts .is_cuda / .is_pinned() while sorting groups
    try:
        m.training_setup(args)
    finally:
        mode.claim_cuda = False
    return m


def clm_stage(sc, cams, Scene, rutils, mode):
    """strategies/clm_offload/engine.py:338-925 clm_offload_train_one_batch, 3 batches."""
    from strategies.clm_offload.engine import clm_offload_train_one_batch, clm_offload_eval_one_cam
    from strategies.clm_offload.gaussian_model import GaussianModelCLMOffload
    args, _ = RH.reference_default_args(clm_offload=True, bsz=BSZ, prealloc_capacity=N + 500)
    rutils.set_args(args)
    m = make_ref_clm_model(sc, args, GaussianModelCLMOffload, mode)
    out = dict(groups_json=json.dumps({g["name"]: dict(lr=float(g["lr"]), eps=float(g["eps"]),
                                                      betas=[float(b) for b in g["betas"]])
                                       for g in m.optimizer.param_groups}),
               columns_lr=np_(m.optimizer.columns_lr))
    gen = torch.Generator().manual_seed(1)
    comm = torch.cuda.Stream()
    iteration = 1
    for b in range(N_BATCHES):
        rutils.set_cur_iter(iteration)
        m.update_learning_rate(iteration)
        batch = cams[b * BSZ:(b + 1) * BSZ]
        losses, ordered_cams, sparsity = clm_offload_train_one_batch(
            m, Scene, batch, m.parameters_grad_buffer, None, None, comm, gen)
        out[f"losses_b{b}"] = np.array([float(l) for l in losses])
        out[f"ordered_cams_b{b}"] = np.array(ordered_cams)
        out[f"sparsity_b{b}"] = np.array(sparsity)
        assert float(m.parameters_grad_buffer.abs().max()) == 0.0  # version 3: consumed rows are zeroed
        iteration += BSZ
    for g in m.optimizer.gpu_adam.param_groups:
        p = g["params"][0]
        st = m.optimizer.gpu_adam.state[p]
        out[f"p_{g['name']}"], out[f"m_{g['name']}"], out[f"v_{g['name']}"] = np_(p), np_(st["exp_avg"]), np_(st["exp_avg_sq"])
    st = m.optimizer.cpu_adam.state[m._parameters]
    out["p_parameters"], out["m_parameters"], out["v_parameters"] = np_(m._parameters), np_(st["exp_avg"]), np_(st["exp_avg_sq"])
    out["xyz_gradient_accum"], out["denom"], out["max_radii2D"] = np_(m.xyz_gradient_accum), np_(m.denom), np_(m.max_radii2D)
    img = clm_offload_eval_one_cam(cams[0], m, None, Scene)
    out["eval_image_cam0"] = np_(img)
    np.savez_compressed(os.path.join(HERE, "engine_clm_offload.npz"), **out)
    print("clm_offload fixture written; losses", out["losses_b0"], "order", out["ordered_cams_b0"])


def naive_stage(sc, cams, Scene, rutils):
    """strategies/naive_offload/engine.py:48-357 naive_offload_train_one_batch (all parameters in host
    memory, whole-model upload per batch, cpu_adam.CPUAdam over six groups), 3 batches dense + the same 3
    batches with sparse_adam=True (sparse_step over the visible rows) -> engine_naive_offload.npz."""
    from strategies.naive_offload.engine import naive_offload_eval_one_cam, naive_offload_train_one_batch
    from strategies.naive_offload.gaussian_model import GaussianModelNaiveOffload
    out = {}
    for tag, sparse in (("dense", False), ("sparse", True)):
        args, _ = RH.reference_default_args(naive_offload=True, bsz=BSZ, sparse_adam=sparse)
        rutils.set_args(args)
        m = make_ref_model(sc, args, GaussianModelNaiveOffload)
        m.sum_visible_count_in_one_batch = torch.zeros((N,))
        if tag == "dense":
            out["groups_json"] = json.dumps({g["name"]: dict(lr=float(g["lr"]), eps=float(g["eps"]),
                                                              betas=[float(b) for b in g["betas"]])
                                             for g in m.optimizer.param_groups})
        iteration = 1
        for b in range(N_BATCHES):
            rutils.set_cur_iter(iteration)
            m.update_learning_rate(iteration)
            losses, vis = naive_offload_train_one_batch(m, Scene, cams[b * BSZ:(b + 1) * BSZ], None, sparse_adam=sparse)
            out[f"{tag}_losses_b{b}"] = np.array([float(l) for l in losses])
            if sparse:
                out[f"{tag}_visibility_b{b}"] = np_(vis)
            assert all(p.grad is None for p in m.all_parameters())  # zero_grad(set_to_none=True), engine.py:336
            iteration += BSZ
        for g in m.optimizer.param_groups:
            p = g["params"][0]
            st = m.optimizer.state[p]
            out[f"{tag}_p_{g['name']}"], out[f"{tag}_m_{g['name']}"], out[f"{tag}_v_{g['name']}"] = \
                np_(p), np_(st["exp_avg"]), np_(st["exp_avg_sq"])
        out[f"{tag}_xyz_gradient_accum"], out[f"{tag}_denom"] = np_(m.xyz_gradient_accum), np_(m.denom)
        out[f"{tag}_max_radii2D"] = np_(m.max_radii2D)
        if tag == "dense":
            out["eval_image_cam0"] = np_(naive_offload_eval_one_cam(m, Scene, cams[0], None))
    np.savez_compressed(os.path.join(HERE, "engine_naive_offload.npz"), **out)
    print("naive_offload fixture written; losses", out["dense_losses_b0"], out["sparse_losses_b2"])


def main():
    RH.install_stubs()
    import utils.general_utils as rutils
    args, _groups = RH.reference_default_args(no_offload=True, bsz=BSZ)
    # arguments defaults, verbatim (JSON-able values only)
    dflt, _ = RH.reference_default_args()
    with open(os.path.join(HERE, "arguments_defaults.json"), "w") as f:
        json.dump({k: v for k, v in sorted(vars(dflt).items())}, f, indent=1, default=str)

    rutils.set_args(args)
    rutils.set_log_file(RH.NullLog())
    rutils.set_img_size(H, W)
    rutils.set_cur_iter(1)
    from utils.timer import Timer
    rutils.set_timers(Timer(args))
    rutils.check_initial_gpu_memory_usage = lambda *a, **k: None
    rutils.check_memory_usage = lambda *a, **k: None

    sc = ci_scene()
    cams = [RH.RefCamera(i, sc["w2c"][i], sc["fovx"], sc["fovy"], W, H, sc["gt"][i]) for i in range(BSZ * N_BATCHES)]

    class Scene:
        cameras_extent = sc["extent"]

    with RH.CudaToCpu() as mode:
        if "densify" in sys.argv[1:]:
            return densify_stage(sc, Scene, rutils)
        if "naive" in sys.argv[1:] or len(sys.argv) == 1:
            naive_stage(sc, cams, Scene, rutils)
            if "naive" in sys.argv[1:]:
                return
            rutils.set_args(args)
        if "clm" in sys.argv[1:] or len(sys.argv) == 1:
            clm_stage(sc, cams, Scene, rutils, mode)
            if "clm" in sys.argv[1:]:
                return
            rutils.set_args(args)
        from strategies.base_engine import calculate_filters
        from strategies.no_offload.engine import baseline_accumGrads_impl
        from strategies.no_offload.gaussian_model import GaussianModelNoOffload
        import densification as D

        m = make_ref_model(sc, args, GaussianModelNoOffload)
        groups = {g["name"]: dict(lr=float(g["lr"]), eps=float(g["eps"]), betas=[float(b) for b in g["betas"]])
                  for g in m.optimizer.param_groups}
        out = dict(xyz=np_(sc["xyz"]), scaling=np_(sc["scaling"]), rotation=np_(sc["rotation"]),
                   opacity=np_(sc["opacity"]), shs48=np_(sc["shs"].reshape(N, 48)), w2c=np_(sc["w2c"]),
                   fovx=sc["fovx"], fovy=sc["fovy"], gt=np_(sc["gt"]), extent=sc["extent"], W=W, H=H, bsz=BSZ,
                   n_batches=N_BATCHES, groups_json=json.dumps(groups))

        # ---- calculate_filters (activated attributes, as clm_offload/engine.py:362-370 passes them)
        filters, cam_ids, g_ids = calculate_filters(cams[:BSZ], m.get_xyz, m.get_opacity, m.get_scaling, m.get_rotation)
        np.savez_compressed(os.path.join(HERE, "engine_filters.npz"), camera_ids=np_(cam_ids), gaussian_ids=np_(g_ids),
                            counts=np.array([f.numel() for f in filters]))

        # ---- batch 1: accumulated gradients before any optimizer step
        iteration = 1
        lrs = []
        for b in range(N_BATCHES):
            rutils.set_cur_iter(iteration)
            lrs.append(float(m.update_learning_rate(iteration)))
            batch = cams[b * BSZ:(b + 1) * BSZ]
            losses, vis = baseline_accumGrads_impl(m, Scene, batch, None)
            assert vis is None
            if b == 0:
                out.update(losses=np.array([float(l) for l in losses]),
                           g_xyz=np_(m._xyz.grad), g_f_dc=np_(m._features_dc.grad), g_f_rest=np_(m._features_rest.grad),
                           g_opacity=np_(m._opacity.grad), g_scaling=np_(m._scaling.grad), g_rotation=np_(m._rotation.grad),
                           xyz_gradient_accum=np_(m.xyz_gradient_accum), denom=np_(m.denom), max_radii2D=np_(m.max_radii2D))
                stats0 = (m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone())
            out[f"losses_b{b}"] = np.array([float(l) for l in losses])
            # train.py:533-578
            for p in m.all_parameters():
                if p.grad is not None:
                    p.grad /= args.bsz
            m.optimizer.step()
            m.optimizer.zero_grad(set_to_none=True)
            iteration += BSZ
        out["xyz_lr"] = np.array(lrs)
        for g in m.optimizer.param_groups:
            p = g["params"][0]
            st = m.optimizer.state[p]
            out[f"p_{g['name']}"] = np_(p)
            out[f"m_{g['name']}"] = np_(st["exp_avg"])
            out[f"v_{g['name']}"] = np_(st["exp_avg_sq"])
        out["stats3_accum"], out["stats3_denom"] = np_(m.xyz_gradient_accum), np_(m.denom)
        out["stats3_max_radii2D"] = np_(m.max_radii2D)
        np.savez_compressed(os.path.join(HERE, "engine_no_offload.npz"), **out)

        densify_stage(sc, Scene, rutils)


def densify_stage(sc, Scene, rutils):
    """Reference control flow + model surgery on the reference's own 3-batch state (loaded back from
    engine_no_offload.npz, so this stage can be regenerated alone: `make_engine_golden.py densify`)."""
    import densification as D
    from strategies.no_offload.gaussian_model import GaussianModelNoOffload
    s3 = np.load(os.path.join(HERE, "engine_no_offload.npz"))
    # percent_dense * extent ~ the median of max(scale): both clone and split fire; min_opacity prunes ~2 %
    args, _ = RH.reference_default_args(no_offload=True, bsz=BSZ, percent_dense=0.045, min_opacity=0.05)
    rutils.set_args(args)
    m = make_ref_model(sc, args, GaussianModelNoOffload)
    with torch.no_grad():
        for g in m.optimizer.param_groups:
            p = g["params"][0]
            p.copy_(torch.from_numpy(s3[f"p_{g['name']}"]))
            m.optimizer.state[p] = {"step": torch.tensor(3.0), "exp_avg": torch.from_numpy(s3[f"m_{g['name']}"]).clone(),
                                    "exp_avg_sq": torch.from_numpy(s3[f"v_{g['name']}"]).clone()}
    m.xyz_gradient_accum = torch.from_numpy(s3["stats3_accum"]).clone()
    m.denom = torch.from_numpy(s3["stats3_denom"]).clone()
    m.max_radii2D = torch.from_numpy(s3["stats3_max_radii2D"]).clone()

    # schedule: which iterations densify / reset, straight from gsplat_densification
    sched = []
    real_dp, real_ro = m.densify_and_prune, m.reset_opacity
    probe = dict(d=0, r=0)
    m.densify_and_prune = lambda *a, **k: probe.__setitem__("d", probe["d"] + 1)
    m.reset_opacity = lambda *a, **k: probe.__setitem__("r", probe["r"] + 1)
    for it in range(1, 15200, BSZ):
        probe["d"] = probe["r"] = 0
        rutils.set_cur_iter(it)
        D.gsplat_densification(it, Scene, m, None)
        if probe["d"] or probe["r"]:
            sched.append((it, probe["d"], probe["r"]))
    m.densify_and_prune, m.reset_opacity = real_dp, real_ro

    # record the unit-normal draws of densify_and_split (the CPU and GPU generators differ; the test
    # feeds the same draws to the build's densify_and_split)
    draws = []
    real_normal = torch.normal

    def rec_normal(mean=None, std=None, *a, **k):
        z = real_normal(torch.zeros_like(std), torch.ones_like(std), generator=torch.Generator().manual_seed(7))
        draws.append(z.clone())
        return mean + std * z
    torch.normal = rec_normal
    thr = float(np.quantile((s3["stats3_accum"] / np.maximum(s3["stats3_denom"], 1)).ravel(), 0.9))  # ~10 % densify
    args.densify_grad_threshold = thr
    it_d = 597  # > densify_from_iter (500); [597, 601) contains 600 -> densifies; size_threshold None (< 3000)
    rutils.set_cur_iter(it_d)
    D.gsplat_densification(it_d, Scene, m, None)
    torch.normal = real_normal
    assert len(draws) == 1
    d = dict(iteration=it_d, grad_threshold=thr, percent_dense=float(m.percent_dense), min_opacity=args.min_opacity,
             extent=sc["extent"], n_before=N, n_after=int(m.get_xyz.shape[0]), split_z=np_(draws[0]),
             n_split=draws[0].shape[0] // 2, schedule=np.array(sched))
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        st = m.optimizer.state[p]
        d[f"p_{g['name']}"] = np_(p)
        d[f"m_{g['name']}"] = np_(st["exp_avg"])
        d[f"v_{g['name']}"] = np_(st["exp_avg_sq"])
    d["max_radii2D"] = np_(m.max_radii2D)
    d["xyz_gradient_accum"], d["denom"] = np_(m.xyz_gradient_accum), np_(m.denom)
    # opacity reset (base: reset_opacity at every opacity_reset_interval): state after it
    m.reset_opacity()
    for g in m.optimizer.param_groups:
        if g["name"] == "opacity":
            p = g["params"][0]
            d["reset_p_opacity"], d["reset_m_opacity"] = np_(p), np_(m.optimizer.state[p]["exp_avg"])
    np.savez_compressed(os.path.join(HERE, "engine_densify.npz"), **d)
    print("densify fixture written:", N, "->", d["n_after"], "rows,", d["n_split"], "split; schedule", sched[:4], "...",
          [x for x in sched if x[2]][:3])


if __name__ == "__main__":
    main()
