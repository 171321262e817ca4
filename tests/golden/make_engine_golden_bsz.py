"""BUILD-CONTAINER ONLY.  The reference's OWN clm_offload engine at the batch sizes its release scripts use beyond
4 / 8 (strategies/clm_offload/engine.py:137-147: bitmap int16 at bsz 16, int64 at bsz 64; :159-166: the sampling
rule changes at bsz >= 32; release_scripts/bigcity.sh:73-92 runs bsz 64), imported from /root/reference and run on
the CPU under ref_harness.py with oracle/ as the absent native modules:

    python tests/golden/make_engine_golden_bsz.py            # both sizes, dense + sparse_adam (~10 min of CPU)
    python tests/golden/make_engine_golden_bsz.py 16         # one size

  engine_clm_offload_bsz16.npz / engine_clm_offload_bsz64.npz
      inputs      scene (2048 Gaussians at bsz 16, 4096 at bsz 64; 96x64, 2 x bsz cameras), gt images
      pre_*       strategies/no_offload/engine.py:104-177 baseline_accumGrads_impl on batch 0 at this bsz: the batch
                  gradient BEFORE any optimizer step (what clm's optimizers are about to consume), losses, statistics
      dense_* / sparse_*   strategies/clm_offload/engine.py:338-925 clm_offload_train_one_batch, 2 batches:
                  losses, ordered_cams, sparsity, and what order_calculation (:135-298) returned inside each batch --
                  the finish_indices_filters partition (concatenated + sizes), cnt_h / cnt_d / cnt_g, filter sizes in
                  processing order, visibility_mask (sparse) -- then parameters + both Adam moments of all five groups,
                  the densification statistics.

The GPU tests (tests/test_gpu_golden_engine_bsz.py) put the same inputs through this build's engines.
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
import make_engine_golden as MG  # noqa: E402

W, H, N_BATCHES = 96, 64, 2
N_BY_BSZ = {16: 2048, 64: 4096}  # bsz 64 samples N // bsz^2 rows for the distance matrix (engine.py:159-162)
N = None  # set per size in main()
SEED = 20251001


def scene(n_cams):
    """Seeded aerial slab + nadir cameras with a small tilt (the family of make_engine_golden.ci_scene); cameras see
    ~20 % of the slab each, so that the H / D / G retention sets of neighbouring cameras are all non-trivial.  Ground
    truth: an integer pattern (compresses; any uint8 image is a valid target)."""
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
    f = 0.8 * W
    fovx, fovy = 2 * math.atan(W / (2 * f)), 2 * math.atan(H / (2 * f))
    h = math.sqrt(0.20 * (2 * L) ** 2 * f * f / (W * H))
    w2cs = []
    for _ in range(n_cams):
        cx = float((torch.rand((), generator=g) * 2 - 1) * 0.6 * L)
        cy = float((torch.rand((), generator=g) * 2 - 1) * 0.6 * L)
        ang = float((torch.rand((), generator=g) * 2 - 1) * 0.15)
        R0 = torch.tensor([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])
        Rx = torch.tensor([[1.0, 0, 0], [0, math.cos(ang), -math.sin(ang)], [0, math.sin(ang), math.cos(ang)]])
        R = Rx @ R0
        C = torch.tensor([cx, cy, 0.1 * L + h])
        w2c = torch.eye(4)
        w2c[:3, :3] = R
        w2c[:3, 3] = -R @ C
        w2cs.append(w2c)
    c = torch.arange(n_cams).view(-1, 1, 1, 1)
    ch = torch.arange(3).view(1, -1, 1, 1)
    y = torch.arange(H).view(1, 1, -1, 1)
    x = torch.arange(W).view(1, 1, 1, -1)
    gt = ((x * 5 + y * 9 + (x * y) // 7 + c * 37 + ch * 71) % 256).to(torch.uint8)
    return dict(xyz=xyz, scaling=scaling, rotation=rotation, opacity=opacity, shs=shs, w2c=torch.stack(w2cs),
                fovx=fovx, fovy=fovy, gt=gt, extent=float(L))


def np_(t):
    return t.detach().cpu().numpy().copy()


def pre_stage(sc, cams, Scene, rutils, bsz, out):
    from strategies.no_offload.engine import baseline_accumGrads_impl
    from strategies.no_offload.gaussian_model import GaussianModelNoOffload
    args, _ = RH.reference_default_args(no_offload=True, bsz=bsz)
    rutils.set_args(args)
    rutils.set_cur_iter(1)
    m = MG.make_ref_model(sc, args, GaussianModelNoOffload)
    m.update_learning_rate(1)
    losses, vis = baseline_accumGrads_impl(m, Scene, cams[:bsz], None)
    assert vis is None
    out.update(pre_losses=np.array([float(l) for l in losses]),
               pre_g_xyz=np_(m._xyz.grad), pre_g_opacity=np_(m._opacity.grad), pre_g_scaling=np_(m._scaling.grad),
               pre_g_rotation=np_(m._rotation.grad),
               pre_g_shs48=np_(torch.cat((m._features_dc.grad, m._features_rest.grad), dim=1).reshape(N, 48)),
               pre_xyz_gradient_accum=np_(m.xyz_gradient_accum), pre_denom=np_(m.denom), pre_max_radii2D=np_(m.max_radii2D))


def clm_stage(sc, cams, Scene, rutils, mode, bsz, sparse, out):
    import strategies.clm_offload.engine as E
    from strategies.clm_offload.gaussian_model import GaussianModelCLMOffload
    tag = "sparse" if sparse else "dense"
    args, _ = RH.reference_default_args(clm_offload=True, bsz=bsz, prealloc_capacity=N + 512, sparse_adam=sparse)
    rutils.set_args(args)
    m = MG.make_ref_clm_model(sc, args, GaussianModelCLMOffload, mode)
    if not sparse:
        out["groups_json"] = json.dumps({g["name"]: dict(lr=float(g["lr"]), eps=float(g["eps"]),
                                                         betas=[float(b) for b in g["betas"]])
                                         for g in m.optimizer.param_groups})
        out["columns_lr"] = np_(m.optimizer.columns_lr)
    rec = []
    real_oc = E.order_calculation

    def recording_oc(filters, batched_cameras, n_gaussians, bsz_, perm_generator, args_):
        r = real_oc(filters, batched_cameras, n_gaussians, bsz_, perm_generator, args_)
        fin, _c, flt, sparsity, ordered, cnt_h, cnt_d, cnt_g, vis = r
        rec.append(dict(fin_cat=np.concatenate([np_(f).astype(np.int32) for f in fin]),
                        fin_sizes=np.array([f.numel() for f in fin]),
                        filter_sizes=np.array([f.numel() for f in flt]),
                        ordered=np.array(ordered), cnt_h=np_(cnt_h), cnt_d=np_(cnt_d), cnt_g=np_(cnt_g),
                        bitmap_dtype=str({4: "int8", 8: "int8", 16: "int16", 32: "int32", 64: "int64"}[bsz_]),
                        vis=None if vis is None else np_(vis)))
        return r
    E.order_calculation = recording_oc
    try:
        gen = torch.Generator().manual_seed(1)
        comm = torch.cuda.Stream()
        iteration = 1
        for b in range(N_BATCHES):
            rutils.set_cur_iter(iteration)
            m.update_learning_rate(iteration)
            batch = cams[b * bsz:(b + 1) * bsz]
            losses, ordered_cams, sparsity = E.clm_offload_train_one_batch(
                m, Scene, batch, m.parameters_grad_buffer, None, None, comm, gen)
            out[f"{tag}_losses_b{b}"] = np.array([float(l) for l in losses])
            out[f"{tag}_ordered_cams_b{b}"] = np.array(ordered_cams)
            out[f"{tag}_sparsity_b{b}"] = np.array(sparsity)
            r = rec[b]
            assert list(r["ordered"]) == list(ordered_cams)
            for k in ("fin_cat", "fin_sizes", "filter_sizes", "cnt_h", "cnt_d", "cnt_g"):
                out[f"{tag}_{k}_b{b}"] = r[k]
            if r["vis"] is not None:
                out[f"{tag}_visibility_b{b}"] = r["vis"]
            out["bitmap_dtype"] = r["bitmap_dtype"]
            assert float(m.parameters_grad_buffer.abs().max()) == 0.0  # version 3: consumed rows are zeroed
            iteration += bsz
    finally:
        E.order_calculation = real_oc
    for g in m.optimizer.gpu_adam.param_groups:
        p = g["params"][0]
        st = m.optimizer.gpu_adam.state[p]
        out[f"{tag}_p_{g['name']}"], out[f"{tag}_m_{g['name']}"], out[f"{tag}_v_{g['name']}"] = \
            np_(p), np_(st["exp_avg"]), np_(st["exp_avg_sq"])
    st = m.optimizer.cpu_adam.state[m._parameters]
    out[f"{tag}_p_parameters"], out[f"{tag}_m_parameters"], out[f"{tag}_v_parameters"] = \
        np_(m._parameters), np_(st["exp_avg"]), np_(st["exp_avg_sq"])
    out[f"{tag}_xyz_gradient_accum"], out[f"{tag}_denom"], out[f"{tag}_max_radii2D"] = \
        np_(m.xyz_gradient_accum), np_(m.denom), np_(m.max_radii2D)
    print(f"bsz {bsz} {tag}: losses b0[:4]", out[f"{tag}_losses_b0"][:4], "order b0[:8]", out[f"{tag}_ordered_cams_b0"][:8],
          "cnt_h/d/g b0[:3]", out[f"{tag}_cnt_h_b0"][:3], out[f"{tag}_cnt_d_b0"][:3], out[f"{tag}_cnt_g_b0"][:3], flush=True)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 64]
    RH.install_stubs()
    import utils.general_utils as rutils
    args, _ = RH.reference_default_args(clm_offload=True, bsz=sizes[0])
    rutils.set_args(args)
    rutils.set_log_file(RH.NullLog())
    rutils.set_img_size(H, W)
    rutils.set_cur_iter(1)
    from utils.timer import Timer
    rutils.set_timers(Timer(args))
    rutils.check_initial_gpu_memory_usage = lambda *a, **k: None
    rutils.check_memory_usage = lambda *a, **k: None
    global N
    for bsz in sizes:
        N = N_BY_BSZ[bsz]
        n_cams = bsz * N_BATCHES
        sc = scene(n_cams)
        cams = [RH.RefCamera(i, sc["w2c"][i], sc["fovx"], sc["fovy"], W, H, sc["gt"][i]) for i in range(n_cams)]

        class Scene:
            cameras_extent = sc["extent"]
        out = dict(xyz=np_(sc["xyz"]), scaling=np_(sc["scaling"]), rotation=np_(sc["rotation"]), opacity=np_(sc["opacity"]),
                   shs48=np_(sc["shs"].reshape(N, 48)), w2c=np_(sc["w2c"]), fovx=sc["fovx"], fovy=sc["fovy"], gt=np_(sc["gt"]),
                   extent=sc["extent"], W=W, H=H, bsz=bsz, n_batches=N_BATCHES)
        with RH.CudaToCpu() as mode:
            pre_stage(sc, cams, Scene, rutils, bsz, out)
            for sparse in (False, True):
                clm_stage(sc, cams, Scene, rutils, mode, bsz, sparse, out)
        np.savez_compressed(os.path.join(HERE, f"engine_clm_offload_bsz{bsz}.npz"), **out)
        print(f"engine_clm_offload_bsz{bsz}.npz written", flush=True)


if __name__ == "__main__":
    main()
