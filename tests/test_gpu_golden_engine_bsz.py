"""-m gpu: the clm_offload engine at the reference's LARGER batch sizes (bsz 16: int16 visibility bitmap; bsz 64:
int64 bitmap, the N // bsz^2 sampling rule -- strategies/clm_offload/engine.py:137-147, 159-166; BigCity runs bsz 64,
release_scripts/bigcity.sh:73-92) against fixtures the REFERENCE'S OWN engine produced.

tests/golden/engine_clm_offload_bsz{16,64}.npz were written in the build container by
tests/golden/make_engine_golden_bsz.py: /root/reference's clm_offload_train_one_batch + order_calculation (recorded
from inside the batch) + baseline_accumGrads_impl, on the CPU with oracle/ as the absent native modules, dense and with
sparse_adam.  Here the same inputs go through this build's engines (hbm fused, hbm op-by-op, host-resident):

  * order_calculation pieces at the reference's processing order: finish_indices_filters partition group for group,
    cnt_h / cnt_d / cnt_g, the sparse_adam visibility mask -- bit-exact (integers);
  * the bitmap ops on the int16 / int64 words vs the oracle's index arithmetic -- bit-exact;
  * pre-optimizer batch gradient (sum over bsz cameras) and statistics;
  * 2 batches: parameters + both Adam moments of all five groups, dense and sparse_adam.

Tolerances as tests/test_gpu_golden_engine.py (fp32 vs fp32, different summation orders -- here over 16 / 64 cameras).
"""
import json
import os

import numpy as np
import pytest
import torch

from tests.scenes import rel_l2

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REPORT = {}


def _rec(key, val):
    REPORT[key] = float(val)
    out = os.path.join(os.path.dirname(G), "..", "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report_bsz.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass
    return val


_FX = {}


def _fx(bsz):
    if bsz not in _FX:
        _FX[bsz] = np.load(os.path.join(G, f"engine_clm_offload_bsz{bsz}.npz"))
    return _FX[bsz]


def _res(residency):
    """"host_batch": host-resident rows staged as the union of the batch (the round-2..5 form); "host": per-camera windows."""
    if residency == "host_batch":
        return {"sh_residency": "host", "host_staging": "batch"}
    if residency == "host_budget":  # ... with about half of the rows (K = 1 002) resident in HBM (sh_hbm_budget_gb)
        return {"sh_residency": "host", "sh_hbm_budget_gb": 7.7e-4}
    if residency == "host_budget_all":  # ... with every row resident: nothing left for the host path to stage
        return {"sh_residency": "host", "sh_hbm_budget_gb": 1.0}
    return {"sh_residency": residency}


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _setup(bsz, **over):
    from clm_gs_amd import utils
    from clm_gs_amd.cameras import Camera
    from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload
    d = _fx(bsz)
    W, H = int(d["W"]), int(d["H"])
    assert int(d["bsz"]) == bsz
    args = utils.default_args(bsz=bsz, **over)
    args.clm_offload = True
    utils.set_args(args)
    utils.set_img_size(H, W)
    utils.set_cur_iter(1)
    cams = [Camera(i, _t(d["w2c"][i]), float(d["fovx"]), float(d["fovy"]), W, H, _t(d["gt"][i]))
            for i in range(d["w2c"].shape[0])]
    m = GaussianModelCLMOffload(3)
    m.create_from_tensors(_t(d["xyz"]).clone(), _t(d["shs48"]).clone(), _t(d["scaling"]).clone(),
                          _t(d["rotation"]).clone(), _t(d["opacity"]).clone(), spatial_lr_scale=1.0)
    m.active_sh_degree = 3
    m.training_setup(args)

    class Scene:
        cameras_extent = float(d["extent"])
    return args, m, cams, Scene, d


def _batch(m, Scene, batch, comm, gen):
    from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
    return clm_offload_train_one_batch(m, Scene, batch, m.parameters_grad_buffer, None, None, comm, gen)


# ------------------------------------------------------------------ a7: order_calculation at bsz 16 / 64
@pytest.mark.parametrize("bsz", [16, 64])
def test_finish_groups_and_retention_counts_match_reference_order_calculation(dev, bsz):
    """The reference's order_calculation ran INSIDE its clm batch; its outputs for the order it chose are in the fixture.
    This build's finish_groups on its own filters, put in the same processing order, must give the same partition of
    [0, N) group for group (as sets: torch.sort of the ffs bytes is not stable across devices), the same retention-set
    sizes and the same visibility mask; order_calculation's own TSP order keeps the invariants."""
    from clm_gs_amd.strategies.base_engine import select_filters
    from clm_gs_amd.strategies.clm_offload.engine import _BITMAP_DTYPE, finish_groups, order_calculation
    args, m, cams, Scene, d = _setup(bsz, sparse_adam=True)
    N = m._xyz.shape[0]
    assert str(_BITMAP_DTYPE[bsz]).endswith(str(d["bitmap_dtype"]))
    with torch.no_grad():
        filters, touched = select_filters(cams[:bsz], m._xyz.detach(), m._scaling.detach(), m._rotation.detach())
    order = d["dense_ordered_cams_b0"].tolist()
    assert sorted(order) == list(range(bsz))
    f_ord = [filters[i] for i in order]
    assert [f.numel() for f in f_ord] == d["dense_filter_sizes_b0"].tolist()
    assert np.allclose([f.numel() / float(N) for f in f_ord], d["dense_sparsity_b0"], rtol=0, atol=1e-12)
    groups, vis, cnt_h, cnt_d, cnt_g, bitmap = finish_groups(f_ord, N, bsz, True)
    assert bitmap.dtype == _BITMAP_DTYPE[bsz] and bitmap.element_size() * 8 >= bsz
    sizes = d["dense_fin_sizes_b0"].tolist()
    assert [g.numel() for g in groups] == sizes and sum(sizes) == N
    ref_groups = np.split(d["dense_fin_cat_b0"], np.cumsum(sizes)[:-1])
    for k, (g, r) in enumerate(zip(groups, ref_groups)):
        assert np.array_equal(g.cpu().numpy(), np.sort(r)), f"group {k}"
    assert cnt_h == d["dense_cnt_h_b0"].tolist() and cnt_d == d["dense_cnt_d_b0"].tolist() and cnt_g == d["dense_cnt_g_b0"].tolist()
    assert np.array_equal(vis.cpu().numpy(), d["sparse_visibility_b0"])
    assert torch.equal(torch.nonzero(vis).flatten(), touched.long())
    # this build's own order (deterministic stride sample for the distance matrix, same TSP + rotation rule)
    gen = torch.Generator(device="cuda").manual_seed(1)
    fin, cams2, filters2, sparsity, ordered, ch, cd, cg, vis2, _ = order_calculation(list(filters), cams[:bsz], N, bsz, gen, args)
    assert sorted(ordered) == list(range(bsz)) and len(fin) == bsz + 1
    assert all(f.dtype == torch.int32 and f.is_pinned() for f in fin if f.numel())
    assert torch.equal(torch.cat([f.long() for f in fin]).sort().values, torch.arange(N))
    if args.reorder_by_min_sparsity_at_end:
        assert filters2[-1].numel() == min(f.numel() for f in filters2)
    sets = [set(f.tolist()) for f in filters2]
    for i in range(bsz - 1):
        assert cd[i] == len(sets[i] & sets[i + 1]) and ch[i] + cd[i] == len(sets[i + 1]) and cg[i] + cd[i] == len(sets[i])
    later = set()
    for k in range(bsz - 1, -1, -1):
        assert set(fin[k + 1].tolist()) == sets[k] - later
        later |= sets[k]
    assert torch.equal(vis2, vis)


@pytest.mark.parametrize("bsz,dtype", [(16, torch.int16), (32, torch.int32), (64, torch.int64)])
def test_bitmap_ops_on_wide_words_match_oracle(dev, bsz, dtype):
    """scatter_to_bit / extract_ffs / compute_cnt_h / pair_overlap_count (clm_kernels; engine.py:150-153, 200-204,
    224-232) on int16 / int32 / int64 words, incl. the sign bit (micro-batch 0 is the MSB), vs oracle/clm_oracle.py."""
    from clm_gs_amd import clm_kernels as K
    from oracle import clm_oracle as CO
    N = 20_000
    g = torch.Generator().manual_seed(100 + bsz)
    filters = [torch.randperm(N, generator=g)[: 500 + 37 * i].sort().values for i in range(bsz)]
    bm = torch.zeros(N, dtype=dtype, device=dev)
    ref = torch.zeros(N, dtype=dtype)
    for i, f in enumerate(filters):
        K.scatter_to_bit(bm, f.to(dev), bsz - 1 - i)
        CO.scatter_to_bit(ref, f, bsz - 1 - i)
    assert torch.equal(bm.cpu(), ref)
    assert int((ref < 0).sum()) == filters[0].numel()  # the MSB is in use
    ffs = torch.empty(N, dtype=torch.uint8, device=dev)
    K.extract_ffs(bm, ffs)
    ffs_ref = torch.empty(N, dtype=torch.uint8)
    CO.extract_ffs(ref, ffs_ref)
    assert torch.equal(ffs.cpu(), ffs_ref)
    tmp = torch.zeros((bsz - 1, 64 * 256), dtype=torch.int32, device=dev)
    K.compute_cnt_h(bm, tmp, 64, 256)
    tmp_ref = torch.zeros((bsz - 1, 64 * 256), dtype=torch.int32)
    CO.compute_cnt_h(ref, tmp_ref, 64, 256)
    sets = [set(f.tolist()) for f in filters]
    want = [len(sets[i] & sets[i + 1]) for i in range(bsz - 1)]
    assert tmp.sum(dim=1).cpu().tolist() == want == tmp_ref.sum(dim=1).tolist()
    assert K.pair_overlap_count(bm, bsz).cpu().tolist() == want


# ------------------------------------------------------------------ a8: the batch at bsz 16 / 64
MODES = [("hbm", True), ("hbm", False), ("host", True), ("host_batch", True), ("host_budget", True), ("host_budget_all", True)]


@pytest.mark.parametrize("residency,fused", MODES)
@pytest.mark.parametrize("bsz", [16, 64])
def test_clm_offload_large_batch_pre_optimizer_gradients_match_reference(dev, bsz, residency, fused):
    """Sum over bsz cameras of the gradients the optimizers are about to consume == the reference engine's batch gradient."""
    args, m, cams, Scene, d = _setup(bsz, **_res(residency), fused_front_end=fused, debug_skip_optimizer=True)
    comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
    losses, order, sparsity = _batch(m, Scene, cams[:bsz], comm, gen)
    torch.cuda.synchronize()
    tag = f"bsz{bsz}.pre.{residency}.{'fused' if fused else 'opbyop'}"
    N = m._xyz.shape[0]
    assert sorted(order) == list(range(bsz))
    e = max(abs(l.item() - float(d["pre_losses"][k])) for k, l in zip(order, losses))
    assert _rec(f"{tag}.loss.max_abs", e) < 2e-6
    assert sorted(round(s * N) for s in sparsity) == sorted(d["dense_filter_sizes_b0"].tolist())
    g_sh = m.parameters_grad_buffer[:N].detach().cpu()
    assert _rec(f"{tag}.grad.shs.rel_l2", rel_l2(g_sh, _t(d["pre_g_shs48"]))) < 5e-5
    if residency == "hbm" and fused:
        gk = m.small_grad().cpu()
        small = {"xyz": gk[:, 0:3], "opacity": gk[:, 3:4], "scaling": gk[:, 4:7], "rotation": gk[:, 7:11]}
    else:
        small = {"xyz": m._xyz.grad.cpu(), "opacity": m._opacity.grad.cpu(), "scaling": m._scaling.grad.cpu(),
                 "rotation": m._rotation.grad.cpu()}
    for name, g in small.items():
        e = rel_l2(g, _t(d[f"pre_g_{name}"]))
        assert _rec(f"{tag}.grad.{name}.rel_l2", e) < 5e-5, (name, e)
    assert torch.equal(m.max_radii2D.cpu(), _t(d["pre_max_radii2D"]))
    assert torch.equal(m.denom.cpu(), _t(d["pre_denom"]))
    assert _rec(f"{tag}.xyz_gradient_accum.rel_l2", rel_l2(m.xyz_gradient_accum.cpu(), _t(d["pre_xyz_gradient_accum"]))) < 5e-5


def _adam_close(tag, name, p, m_, v, d, pre, init):
    p_ref, m_ref, v_ref = _t(d[f"{pre}_p_{name}"]), _t(d[f"{pre}_m_{name}"]), _t(d[f"{pre}_v_{name}"])
    e_m = _rec(f"{tag}.{name}.exp_avg.rel_l2", rel_l2(m_.cpu().reshape(m_ref.shape), m_ref))
    e_v = _rec(f"{tag}.{name}.exp_avg_sq.rel_l2", rel_l2(v.cpu().reshape(v_ref.shape), v_ref))
    init = init.reshape(p_ref.shape)
    e_p = _rec(f"{tag}.{name}.delta.rel_l2", rel_l2(p.detach().cpu().reshape(p_ref.shape) - init, p_ref - init))
    e_abs = _rec(f"{tag}.{name}.param.rel_l2", rel_l2(p.detach().cpu().reshape(p_ref.shape), p_ref))
    assert e_m < 3e-4, (tag, name, "exp_avg", e_m)
    assert e_v < 1e-4, (tag, name, "exp_avg_sq", e_v)
    assert e_p < 5e-4, (tag, name, "delta", e_p)
    assert e_abs < 2e-5, (tag, name, "param", e_abs)


@pytest.mark.parametrize("sparse", [False, True])
@pytest.mark.parametrize("residency,fused", MODES)
@pytest.mark.parametrize("bsz", [16, 64])
def test_clm_offload_large_batch_two_batches_match_reference_engine(dev, bsz, residency, fused, sparse):
    """2 batches of bsz cameras against the reference's clm_offload_train_one_batch run (retention pipeline over bsz
    micro-batches, FusedCPUAdam thread consuming bsz + 1 finish groups, torch Adam / SelectiveAdam for the GPU groups)."""
    from clm_gs_amd import utils
    pre = "sparse" if sparse else "dense"
    args, m, cams, Scene, d = _setup(bsz, **_res(residency), fused_front_end=fused, sparse_adam=sparse)
    comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
    it = 1
    for b in range(int(d["n_batches"])):
        utils.set_cur_iter(it)
        m.update_learning_rate(it)
        if residency.startswith("host") and b + 1 < int(d["n_batches"]):
            from clm_gs_amd.strategies.clm_offload.engine import hint_next_batch
            hint_next_batch(m, cams[(b + 1) * bsz:(b + 2) * bsz])
        losses, order, _ = _batch(m, Scene, cams[b * bsz:(b + 1) * bsz], comm, gen)
        ref = dict(zip(d[f"{pre}_ordered_cams_b{b}"].tolist(), d[f"{pre}_losses_b{b}"].tolist()))
        for k, l in zip(order, losses):
            assert abs(l.item() - ref[k]) < 5e-5, (b, k, l.item(), ref[k])
        it += bsz
    torch.cuda.synchronize()
    m.flush_lazy_rows()
    tag = f"bsz{bsz}.{pre}2.{residency}.{'fused' if fused else 'opbyop'}"
    init = {"xyz": _t(d["xyz"]), "opacity": _t(d["opacity"]), "scaling": _t(d["scaling"]),
            "rotation": _t(d["rotation"]), "parameters": _t(d["shs48"])}
    for g in m.optimizer.gpu_adam.param_groups:
        p = g["params"][0]
        st = m.optimizer.gpu_adam.state[p]
        _adam_close(tag, g["name"], p, st["exp_avg"], st["exp_avg_sq"], d, pre, init[g["name"]])
    st = m.optimizer.cpu_adam.state[m._parameters]
    _adam_close(tag, "parameters", m._parameters, st["exp_avg"], st["exp_avg_sq"], d, pre, init["parameters"])
    assert torch.equal(m.denom.cpu(), _t(d[f"{pre}_denom"]))
    assert torch.equal(m.max_radii2D.cpu(), _t(d[f"{pre}_max_radii2D"]))
    assert _rec(f"{tag}.xyz_gradient_accum.rel_l2",
                rel_l2(m.xyz_gradient_accum.cpu(), _t(d[f"{pre}_xyz_gradient_accum"]))) < 5e-5
