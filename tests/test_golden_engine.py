"""CPU-only: the reference-engine fixtures (tests/golden/engine_*.npz, arguments_defaults.json --
written by the reference's OWN Python in the build container, tests/golden/make_engine_golden.py)
against the oracle composition and the product's host-side logic."""
import json
import math
import os
from types import SimpleNamespace

import numpy as np
import torch

from oracle import clm_oracle as CO
from oracle import gs_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_default_args_match_reference_argparse_defaults():
    """clm_gs_amd.utils.default_args is a hand-typed table: every flag it shares with the reference's
    six ParamGroups (arguments/__init__.py, parser.parse_args([])) must carry the reference's default."""
    from clm_gs_amd import utils
    ref = json.load(open(os.path.join(G, "arguments_defaults.json")))
    mine = vars(utils.default_args())
    shared = sorted(set(ref) & set(mine))
    assert len(shared) >= 35, shared
    # deliberate differences: none of the shared flags may differ
    diff = {k: (mine[k], ref[k]) for k in shared if mine[k] != ref[k]}
    assert not diff, diff
    for k in ("bsz", "lr_scale_mode", "densification_interval", "opacity_reset_interval", "densify_from_iter",
              "densify_until_iter", "densify_grad_threshold", "percent_dense", "min_opacity", "grid_size_H",
              "grid_size_D", "position_lr_init", "feature_lr", "lambda_dssim"):
        assert k in shared, k


def test_densification_schedule_matches_reference_control_flow():
    """Which image counters trigger densify_and_prune / reset_opacity (densification.py:5-56 run in the
    build container with bsz 4) == the product's gsplat_densification with a probe model."""
    from clm_gs_amd import densification as D
    from clm_gs_amd import utils
    d = _load("engine_densify.npz")
    args = utils.default_args(bsz=4)
    utils.set_args(args)
    calls = []

    class Probe:
        optimizer = SimpleNamespace(zero_grad=lambda set_to_none=True: None)

        def densify_and_prune(self, max_grad, min_opacity, extent, size_threshold):
            calls.append(("d", size_threshold))

        def reset_opacity(self):
            calls.append(("r", None))

    sched = []
    for it in range(1, 15200, 4):
        calls.clear()
        utils.set_cur_iter(it)
        D.gsplat_densification(it, SimpleNamespace(cameras_extent=1.0), Probe(), None)
        if calls:
            sched.append((it, sum(c[0] == "d" for c in calls), sum(c[0] == "r" for c in calls)))
            for kind, thr in calls:  # size_threshold 20 only once iteration > opacity_reset_interval
                if kind == "d":
                    assert thr == (20 if it > args.opacity_reset_interval else None)
    assert np.array_equal(np.array(sched), d["schedule"])


def test_reference_engines_agree_with_each_other():
    """The reference's no_offload engine + torch Adam and its clm_offload engine (retention pipeline +
    FusedCPUAdam thread) produced the same 3-batch state in the build container -- the strategy-vs-strategy
    agreement the reference argues correctness with (release_scripts/mip360_README.md:52-62), and the
    check that oracle/clm_oracle.py's stand-ins did not bend the CLM run."""
    a, c = _load("engine_no_offload.npz"), _load("engine_clm_offload.npz")

    def rl(x, y):
        return np.linalg.norm((x - y).ravel()) / np.linalg.norm(y.ravel())
    for n in ("xyz", "opacity", "scaling", "rotation"):
        assert rl(c["p_" + n] - a[n], a["p_" + n] - a[n]) < 1e-5
        assert rl(c["m_" + n], a["m_" + n]) < 1e-5 and rl(c["v_" + n], a["v_" + n]) < 1e-5
    sh = np.concatenate([a["p_f_dc"], a["p_f_rest"]], 1).reshape(-1, 48)
    assert rl(c["p_parameters"] - a["shs48"], sh - a["shs48"]) < 1e-5
    assert np.array_equal(a["stats3_denom"], c["denom"]) and np.array_equal(a["stats3_max_radii2D"], c["max_radii2D"])
    for b in range(3):  # same per-camera losses, in the CLM engine's camera order
        order = c[f"ordered_cams_b{b}"]
        assert np.allclose(c[f"losses_b{b}"], a[f"losses_b{b}"][order], atol=1e-6)


def test_oracle_composition_reproduces_reference_engine_loss_and_filters():
    """One camera of the fixture batch through the oracle's render_one_camera composition == the loss the
    reference's baseline_accumGrads_impl produced; packed projection == calculate_filters."""
    d, f = _load("engine_no_offload.npz"), _load("engine_filters.npz")
    W, H, bsz = int(d["W"]), int(d["H"]), int(d["bsz"])
    xyz, opa = _t(d["xyz"]), torch.sigmoid(_t(d["opacity"]))
    sca, rot = torch.exp(_t(d["scaling"])), torch.nn.functional.normalize(_t(d["rotation"]))
    fx, fy = W / (2 * math.tan(float(d["fovx"]) * 0.5)), H / (2 * math.tan(float(d["fovy"]) * 0.5))
    K = torch.tensor([[fx, 0, W / 2.0], [0, fy, H / 2.0], [0, 0, 1]], dtype=torch.float32)
    with torch.no_grad():
        vm = _t(d["w2c"][:bsz])
        cam_ids, g_ids, *_ = O.fully_fused_projection(xyz, None, rot, sca, vm, K[None].expand(bsz, 3, 3), W, H, packed=True)
        assert np.array_equal(cam_ids.numpy(), f["camera_ids"]) and np.array_equal(g_ids.numpy(), f["gaussian_ids"])
        sel = g_ids[cam_ids == 1]  # camera 1, only its visible rows (what the clm engine renders)
        img, _, _, _ = O.render_one_camera(xyz[sel], opa[sel], sca[sel], rot[sel], _t(d["shs48"]).reshape(-1, 16, 3)[sel],
                                           3, vm[1], K, W, H)
        loss = O.training_loss(img, _t(d["gt"][1]))
    assert abs(loss.item() - float(d["losses"][1])) < 1e-6


def test_clm_standins_against_definitions():
    """oracle/clm_oracle.py (the fake clm_kernels of the reference-engine run): bit ops and movers on
    random data against brute-force Python."""
    g = torch.Generator().manual_seed(3)
    n, bsz = 500, 8
    filters = [torch.randperm(n, generator=g)[: 100 + 30 * i].sort().values for i in range(bsz)]
    bm = torch.zeros(n, dtype=torch.int8)
    for i, f in enumerate(filters):
        CO.scatter_to_bit(bm, f, bsz - 1 - i)
    ffs = torch.empty(n, dtype=torch.uint8)
    CO.extract_ffs(bm, ffs)
    sets = [set(f.tolist()) for f in filters]
    for r in range(n):
        last = max([i for i in range(bsz) if r in sets[i]], default=None)
        assert int(ffs[r]) == (0 if last is None else bsz - last)
    tmp = torch.empty((bsz - 1, 16), dtype=torch.int32)
    CO.compute_cnt_h(bm, tmp)
    assert tmp.sum(dim=1).tolist() == [len(sets[i] & sets[i + 1]) for i in range(bsz - 1)]
    params = torch.randn(n, 48, generator=g)
    shs = torch.empty(len(filters[0]), 48)
    CO.send_shs2gpu_stream(shs, params, filters[0])
    assert torch.equal(shs, params[filters[0]])
    buf = torch.zeros(n, 48)
    CO.send_shs2cpu_grad_buffer_stream(shs, buf, filters[0], True)
    CO.send_shs2cpu_grad_buffer_stream(shs, buf, filters[0], True)
    assert torch.equal(buf[filters[0]], 2 * shs)
    dist = [[0, 5, 9, 1], [5, 0, 2, 8], [9, 2, 0, 7], [1, 8, 7, 0]]
    t = CO.find_tour(dist)
    assert sum(dist[t[i]][t[i + 1]] for i in range(3)) == 8


def test_trainer_log_lines_are_what_the_reference_scraper_parses():
    """Row f1: the lines clm_gs_amd.trainer writes were fed to the reference's OWN
    release_scripts/log2csv.py (build container, tests/golden/make_log_golden.py); the fixture holds the
    lines and what the reference parsed.  Here: the formatters still produce exactly those lines, and the
    parsed values are the ones put in."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_log_golden", os.path.join(G, "make_log_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fx = json.load(open(os.path.join(G, "log_contract.json")))
    assert mod.sample_log() == fx["log"]
    p = fx["parsed_by_reference"]
    assert p["test_psnr"] == 32.101051330566406 and p["train_psnr"] == 33.42594528198242
    assert p["num_3dgs"] == 1215377 and p["iterations"] == 30001
    assert abs(p["total_time_s"] - 351.06) < 1e-9 and abs(p["throughput"] - 85.46) < 1e-9
    assert p["max_gpu_memory_gb"] == 1.75 and p["pinned_cpu_memory_gb"] is not None


def test_reference_naive_offload_run_agrees_with_reference_no_offload_run():
    """engine_naive_offload.npz (the reference's naive_offload_train_one_batch, 3 batches, host CPUAdam stood in
    by oracle/clm_oracle.CPUAdam) vs engine_no_offload.npz (the reference's baseline engine + torch Adam):
    two engines of the reference, two optimizers, one trajectory -- the stand-in did not bend the run.  The
    sparse run differs from the dense one exactly on rows some batch did not see."""
    a, b = _load("engine_naive_offload.npz"), _load("engine_no_offload.npz")
    for bi in range(3):
        assert np.allclose(a[f"dense_losses_b{bi}"], b[f"losses_b{bi}"], atol=1e-6)
    for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
        for k in "pmv":
            x, y = a[f"dense_{k}_{n}"], b[f"{k}_{n}"]
            assert np.linalg.norm(x - y) <= 1e-6 * np.linalg.norm(y) + 1e-12, (n, k)
    assert np.array_equal(a["dense_denom"], b["stats3_denom"])
    assert np.array_equal(a["dense_max_radii2D"], b["stats3_max_radii2D"])
    seen_all = a["sparse_visibility_b0"] & a["sparse_visibility_b1"] & a["sparse_visibility_b2"]
    assert 0 < seen_all.sum() < seen_all.size
    never = ~(a["sparse_visibility_b0"] | a["sparse_visibility_b1"] | a["sparse_visibility_b2"])
    assert never.any() and np.array_equal(a["sparse_p_xyz"][never], b["xyz"][never])  # never visible: never stepped
    assert not np.allclose(a["sparse_m_xyz"][~seen_all], a["dense_m_xyz"][~seen_all])


def test_large_batch_fixtures_are_self_consistent():
    """engine_clm_offload_bsz{16,64}.npz (tests/golden/make_engine_golden_bsz.py: the reference's clm engine at bsz 16 /
    64): the recorded order_calculation outputs obey the identities of engine.py:194-235 (partition of [0, N) by last
    use; cnt_h + cnt_d = |F_{i+1}|, cnt_g + cnt_d = |F_i|), the oracle's packed projection selects filters of the
    recorded sizes, the dense and the sparse_adam runs saw the same first batch, and sparse_adam left the rows no
    camera saw untouched."""
    for bsz, word in ((16, "int16"), (64, "int64")):
        d = _load(f"engine_clm_offload_bsz{bsz}.npz")
        N = d["xyz"].shape[0]
        assert int(d["bsz"]) == bsz and str(d["bitmap_dtype"]) == word
        W, H = int(d["W"]), int(d["H"])
        fx = W / (2 * math.tan(float(d["fovx"]) * 0.5))
        fy = H / (2 * math.tan(float(d["fovy"]) * 0.5))
        K = torch.tensor([[fx, 0, W / 2.0], [0, fy, H / 2.0], [0, 0, 1]])
        ref = O.fully_fused_projection(_t(d["xyz"]), None, torch.nn.functional.normalize(_t(d["rotation"])),
                                       torch.exp(_t(d["scaling"])), _t(d["w2c"][:bsz]), K[None].expand(bsz, 3, 3), W, H,
                                       packed=True)
        sets = [set(ref[1][ref[0] == c].tolist()) for c in range(bsz)]
        for b in range(int(d["n_batches"])):
            order = d[f"dense_ordered_cams_b{b}"].tolist()
            assert sorted(order) == list(range(bsz))
            sizes = d[f"dense_fin_sizes_b{b}"]
            assert sizes.shape == (bsz + 1,) and int(sizes.sum()) == N
            assert np.array_equal(np.sort(d[f"dense_fin_cat_b{b}"]), np.arange(N))
            fs = d[f"dense_filter_sizes_b{b}"]
            ch, cd, cg = d[f"dense_cnt_h_b{b}"], d[f"dense_cnt_d_b{b}"], d[f"dense_cnt_g_b{b}"]
            assert np.array_equal(ch + cd, fs[1:]) and np.array_equal(cg + cd, fs[:-1])
            assert np.allclose(d[f"dense_sparsity_b{b}"], fs / float(N))
            if b == 0:
                assert [len(sets[i]) for i in order] == fs.tolist()
                groups = np.split(d["dense_fin_cat_b0"], np.cumsum(sizes)[:-1])
                later = set()
                for k in range(bsz - 1, -1, -1):
                    assert set(groups[k + 1].tolist()) == sets[order[k]] - later
                    later |= sets[order[k]]
                assert set(groups[0].tolist()) == set(range(N)) - later
                assert cd.tolist() == [len(sets[order[i]] & sets[order[i + 1]]) for i in range(bsz - 1)]
        assert np.allclose(d["dense_losses_b0"], d["sparse_losses_b0"], atol=0) and np.array_equal(
            d["dense_ordered_cams_b0"], d["sparse_ordered_cams_b0"])
        by_cam = dict(zip(d["dense_ordered_cams_b0"].tolist(), d["dense_losses_b0"].tolist()))
        assert np.allclose([by_cam[k] for k in range(bsz)], d["pre_losses"], atol=1e-7)
        never = ~(d["sparse_visibility_b0"] | d["sparse_visibility_b1"])
        if never.any():
            assert np.array_equal(d["sparse_p_parameters"][never], d["shs48"][never])
            assert not d["sparse_m_parameters"][never].any() and not d["sparse_v_xyz"][never].any()
