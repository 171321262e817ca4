"""-m gpu: the strategy engines (through the C ABI) against the oracle and against each other.

* no_offload batch: per-camera loss and the six accumulated gradients vs autograd of the oracle;
* clm_offload (HBM-resident) == clm_offload (host-resident, retention pipeline + host Adam thread)
  == no_offload + torch fused Adam after one optimizer step (the reference argues correctness the
  same way: strategy-vs-strategy agreement, release_scripts/mip360_README.md:52-62);
* densification surgery keeps every tensor and optimizer state aligned.
"""

import pytest
import torch

from oracle import gs_oracle as O
from tests.scenes import rel_l2

pytestmark = pytest.mark.gpu

W, H, N, BSZ = 96, 64, 3000, 4


FUSED = True


def _setup(strategy, residency="hbm", sparse=False, seed=0):
    from clm_gs_amd import utils
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
    staging = {}
    if residency == "host_batch":  # host-resident rows staged as the union of the batch (host_batch.py)
        residency, staging = "host", {"host_staging": "batch"}
    if residency == "host_budget":  # host-resident rows, about half of them (K = 1 500) kept in HBM (sh_hbm_budget_gb)
        residency, staging = "host", {"sh_hbm_budget_gb": 1500 * 768 / 1e9 + 1e-9}
    args = utils.default_args(bsz=BSZ, sh_residency=residency, sparse_adam=sparse, fused_front_end=FUSED, **staging)
    setattr(args, strategy, True)
    utils.set_args(args)
    utils.set_img_size(H, W)
    utils.set_cur_iter(1)
    sc = synth_gaussians(N, seed=seed, device="cuda")
    cams = nadir_cameras(BSZ, N, W, H, 0.35, seed=seed, device="cuda")
    g = torch.Generator().manual_seed(5)
    for c in cams:
        c.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
    return args, sc, cams


def _make(strategy, sc, args):
    if strategy == "no_offload":
        from clm_gs_amd.strategies.no_offload import GaussianModelNoOffload as M
    else:
        from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload as M
    m = M(3)
    m.create_from_tensors(sc["xyz"].clone(), sc["shs48"].clone(), sc["scaling"].clone(),
                          sc["rotation"].clone(), sc["opacity"].clone(), spatial_lr_scale=1.0)
    m.active_sh_degree = 3
    m.training_setup(args)
    return m


class _Scene:
    cameras_extent = 30.0


def test_no_offload_batch_matches_oracle(dev):
    from clm_gs_amd.strategies.no_offload import baseline_accumGrads_impl
    args, sc, cams = _setup("no_offload")
    m = _make("no_offload", sc, args)
    losses, vis = baseline_accumGrads_impl(m, _Scene, cams, None)
    assert vis is None and len(losses) == BSZ
    # oracle: same composition on the CPU in float64
    P = {k: sc[k].detach().cpu().double().requires_grad_() for k in ("xyz", "opacity", "scaling", "rotation", "shs48")}
    tot = []
    for c in cams:
        vm = c.world_view_transform.t().cpu().double()
        img, _, _, _ = O.render_one_camera(P["xyz"], torch.sigmoid(P["opacity"]), torch.exp(P["scaling"]),
                                           torch.nn.functional.normalize(P["rotation"]),
                                           P["shs48"].reshape(-1, 16, 3), 3, vm, c.K.cpu().double(), W, H)
        l = O.training_loss(img, c.original_image.cpu())
        l.backward()
        tot.append(l.item())
    for a, b in zip(losses, tot):
        assert abs(a.item() - b) < 2e-5
    assert rel_l2(m._xyz.grad.cpu(), P["xyz"].grad) < 1e-3
    assert rel_l2(m._opacity.grad.cpu(), P["opacity"].grad) < 1e-3
    assert rel_l2(m._scaling.grad.cpu(), P["scaling"].grad) < 1e-3
    assert rel_l2(m._rotation.grad.cpu(), P["rotation"].grad) < 1e-3
    gsh = torch.cat((m._features_dc.grad, m._features_rest.grad), dim=1).reshape(-1, 48)
    assert rel_l2(gsh.cpu(), P["shs48"].grad) < 1e-3
    # densification statistics (densification.py:105-147)
    assert (m.denom.sum() > 0) and (m.xyz_gradient_accum.sum() > 0) and (m.max_radii2D.max() > 0)


def _one_step(strategy, residency, sparse=False):
    args, sc, cams = _setup(strategy, residency, sparse)
    m = _make(strategy, sc, args)
    if strategy == "no_offload":
        from clm_gs_amd.strategies.no_offload import baseline_accumGrads_impl
        losses, _ = baseline_accumGrads_impl(m, _Scene, cams, None)
        for p in m.all_parameters():
            p.grad /= BSZ
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        shs = torch.cat((m._features_dc, m._features_rest), dim=1).reshape(-1, 48).detach()
        order = list(range(BSZ))
    else:
        from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
        comm = torch.cuda.Stream()
        gen = torch.Generator(device="cuda").manual_seed(1)
        losses, order, sparsity = clm_offload_train_one_batch(m, _Scene, cams, m.parameters_grad_buffer, None, None, comm, gen)
        assert len(sparsity) == BSZ and all(0 < s <= 1 for s in sparsity) and sorted(order) == list(range(BSZ))
        m.flush_lazy_rows()  # deferred row optimizers (HBM lazy rows / host rows): apply what is pending
        shs = m._parameters.detach().cuda() if not m._parameters.is_cuda else m._parameters.detach()
        if m._parameters.is_cuda and m.first_touch_grads:
            # first-touch policy: a consumed gradient row is marked by its stamps, not cleared
            assert bool((m._row_g_step[:N] <= m._row_last_step[:N]).all()) and int(m._row_last_step[:N].min()) == 1
        elif m._parameters.is_cuda:
            assert float(m.parameters_grad_buffer[:N].abs().max()) == 0.0, "consumed grad rows must be zeroed"
        else:  # host rows: a consumed gradient row is marked by its stamp, not overwritten
            assert int(m._host_g_step[:N].max()) == 0 and int(m._host_last_step[:N].min()) == 1
    torch.cuda.synchronize()
    lo = [0.0] * BSZ
    for k, l in zip(order, losses):
        lo[k] = l.item()
    return dict(xyz=m._xyz.detach().clone(), opacity=m._opacity.detach().clone(), scaling=m._scaling.detach().clone(),
                rotation=m._rotation.detach().clone(), shs=shs.clone(), losses=lo, model=m, init=sc)


def _frac_differs(a, b, init, lr_tol):
    """Adam's first step is ~ +-lr per element: count elements whose step differs noticeably."""
    da, db = (a - init).cpu(), (b - init).cpu()
    scale = db.abs().max().item() + 1e-30
    return ((da - db).abs() > lr_tol * scale).float().mean().item()


@pytest.mark.parametrize("sparse", [False, True])
def test_clm_hbm_equals_host_equals_no_offload_after_one_step(dev, sparse):
    if sparse:
        a = _one_step("clm_offload", "hbm", True)
        b = _one_step("clm_offload", "host", True)
        pairs = [(a, b)]
    else:
        a = _one_step("clm_offload", "hbm")
        b = _one_step("clm_offload", "host")
        c = _one_step("no_offload", "hbm")
        pairs = [(a, b), (a, c)]
    for x, y in pairs:
        for u, v in zip(x["losses"], y["losses"]):
            assert abs(u - v) < 1e-5
        for k, init_k in (("xyz", "xyz"), ("opacity", "opacity"), ("scaling", "scaling"), ("rotation", "rotation"), ("shs", "shs48")):
            frac = _frac_differs(x[k], y[k], x["init"][init_k], 0.02)
            assert frac < 0.01, (k, frac)
    assert (a["xyz"] - a["init"]["xyz"]).abs().max() > 0


def test_fused_front_end_equals_op_by_op_path(dev):
    """fused.py (2 front-end kernels, no autograd) == the gsplat/clm_kernels op chain."""
    global FUSED
    try:
        FUSED = True
        a = _one_step("clm_offload", "hbm")
        FUSED = False
        b = _one_step("clm_offload", "hbm")
    finally:
        FUSED = True
    for u, v in zip(a["losses"], b["losses"]):
        assert abs(u - v) < 1e-6
    for k, init_k in (("xyz", "xyz"), ("opacity", "opacity"), ("scaling", "scaling"), ("rotation", "rotation"), ("shs", "shs48")):
        frac = _frac_differs(a[k], b[k], a["init"][init_k], 0.02)
        assert frac < 0.005, (k, frac)
    ma, mb = a["model"], b["model"]
    assert torch.allclose(ma.denom, mb.denom) and torch.allclose(ma.max_radii2D, mb.max_radii2D)
    assert rel_l2(ma.xyz_gradient_accum.cpu(), mb.xyz_gradient_accum.cpu()) < 1e-4


def test_camera_schedules_are_bit_identical(dev):
    """The pipelined three-stream schedule (no end-of-batch synchronisation) and the single-stream schedule run
    the same kernels on the same data in a data-race-free order: two batches must end bit-identical
    (the backward is atomic-free, so nothing depends on timing)."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
    from clm_gs_amd.synthetic import nadir_cameras
    outs = []
    for mode in (True, "pipeline", False):
        args, sc, _ = _setup("clm_offload", "hbm")
        args.overlap_cameras = mode
        m = _make("clm_offload", sc, args)
        allc = nadir_cameras(2 * BSZ, N, W, H, 0.3, seed=4, device="cuda")
        g = torch.Generator().manual_seed(8)
        for c in allc:
            c.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
        comm = torch.cuda.Stream()
        losses = []
        for b in range(2):
            utils.set_cur_iter(1 + b * BSZ)
            lo, _, _ = clm_offload_train_one_batch(m, _Scene, allc[b * BSZ:(b + 1) * BSZ],
                                                   m.parameters_grad_buffer, None, None, comm,
                                                   torch.Generator(device="cuda"))
            losses += lo
        m.flush_lazy_rows()
        torch.cuda.synchronize()
        outs.append([torch.stack(losses), m._xyz.detach().clone(), m._opacity.detach().clone(),
                     m._scaling.detach().clone(), m._rotation.detach().clone(), m._parameters.detach().clone(),
                     m.max_radii2D.clone(), m.xyz_gradient_accum.clone(), m.denom.clone()])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_first_touch_gradient_stores_equal_clear_and_accumulate(dev):
    """The fused HBM engine stores a row's SH gradient on its first touch of a step and never clears the
    table (first_touch_grads, default) == the clear-by-the-consumer + read-modify-write policy: three
    batches with moving cameras (rows seen in batch 0 and 2 but not 1 carry a consumed, uncleared gradient
    through batch 1) end bit-identical, and stale rows really are left in the table."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
    from clm_gs_amd.synthetic import nadir_cameras
    outs = []
    for ft in (True, False):
        args, sc, _ = _setup("clm_offload", "hbm")
        args.first_touch_grads = ft
        m = _make("clm_offload", sc, args)
        assert m.first_touch_grads == ft
        allc = nadir_cameras(3 * BSZ, N, W, H, 0.3, seed=4, device="cuda")
        g = torch.Generator().manual_seed(8)
        for c in allc:
            c.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
        comm = torch.cuda.Stream()
        for b in (0, 1, 0):  # the cameras of batch 0 come back after one batch elsewhere
            utils.set_cur_iter(1 + len(outs) * 0 + b * BSZ)
            clm_offload_train_one_batch(m, _Scene, allc[b * BSZ:(b + 1) * BSZ], m.parameters_grad_buffer, None,
                                        None, comm, torch.Generator(device="cuda"))
        m.flush_lazy_rows()
        torch.cuda.synchronize()
        st = m.optimizer.cpu_adam.state[m._parameters]
        outs.append([m._parameters.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                     m._xyz.detach().clone(), m._opacity.detach().clone()])
        stale = float(m.parameters_grad_buffer[:N].abs().max())
        assert (stale > 0) == ft  # consumed rows: left in place / cleared
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_lazy_dense_adam_equals_eager(dev):
    """Deferred zero-gradient Adam replay == streaming every row every batch (3 batches, moving
    cameras so rows go untouched for 1-2 steps and come back)."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
    from clm_gs_amd.synthetic import nadir_cameras
    outs = []
    for lazy in (True, False):
        args, sc, cams = _setup("clm_offload", "hbm")
        args.lazy_dense_adam = lazy
        m = _make("clm_offload", sc, args)
        assert m.lazy_rows == lazy
        allc = nadir_cameras(3 * BSZ, N, W, H, 0.12, seed=3, device="cuda")
        g = torch.Generator().manual_seed(6)
        for c in allc:
            c.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
        comm = torch.cuda.Stream()
        for b in range(3):
            utils.set_cur_iter(1 + b * BSZ)
            clm_offload_train_one_batch(m, _Scene, allc[b * BSZ:(b + 1) * BSZ], m.parameters_grad_buffer,
                                        None, None, comm, torch.Generator(device="cuda"))
        if lazy:
            stale = (m._row_last_step[:N] < 3).float().mean().item()
            assert stale > 0.05, "the test must leave some rows behind"
            m.flush_lazy_rows()
            assert int(m._row_last_step[:N].min()) == 3
        torch.cuda.synchronize()
        st = m.optimizer.cpu_adam.state[m._parameters]
        outs.append((m._parameters.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()))
    for a, b in zip(outs[0], outs[1]):
        assert rel_l2(a.cpu(), b.cpu()) < 2e-6


@pytest.mark.parametrize("strategy,residency", [("clm_offload", "hbm"), ("no_offload", "hbm")])
def test_capture_restore_resumes_training(dev, strategy, residency):
    """Working checkpoint incl. optimizer state (row f3): 2 batches -> capture -> 1 batch  ==
    restore -> the same 1 batch."""
    from clm_gs_amd import utils

    def one_batch(m, cams, it):
        utils.set_cur_iter(it)
        m.update_learning_rate(it)
        if strategy == "clm_offload":
            from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
            clm_offload_train_one_batch(m, _Scene, cams, m.parameters_grad_buffer, None, None,
                                        torch.cuda.Stream(), torch.Generator(device="cuda"))
        else:
            from clm_gs_amd.strategies.no_offload import baseline_accumGrads_impl
            baseline_accumGrads_impl(m, _Scene, cams, None)
            for p in m.all_parameters():
                p.grad /= BSZ
            m.optimizer.step()
            m.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize()

    args, sc, cams = _setup(strategy, residency)
    m = _make(strategy, sc, args)
    one_batch(m, cams, 1)
    one_batch(m, cams, 5)
    state = m.capture()
    before = m._xyz.detach().clone()
    one_batch(m, cams, 9)
    m2 = type(m)(3)
    m2.restore(state, args)
    assert torch.equal(m2._xyz.detach(), before)
    one_batch(m2, cams, 9)
    sh1 = m._shs48_rows(None)
    sh2 = m2._shs48_rows(None)
    for a, b, init in ((m._xyz, m2._xyz, before), (m._opacity, m2._opacity, state["opacity"].cuda()),
                       (sh1, sh2, state["shs48"].cuda())):
        assert _frac_differs(a.detach(), b.detach(), init, 0.02) < 0.01


def test_restore_onto_a_model_that_is_mid_training_drops_its_waiting_steps(dev):
    """train -> restore(old_state) -> train on the SAME model object: the deferred small-attribute steps and the
    deferred row steps the interrupted run still had waiting belong to tensors / an optimizer that restore() replaces;
    they must be dropped with them (not replayed onto the restored tensors, not trip the "row count changed with steps
    waiting" assertion), and the run must continue exactly like a fresh model restored from the same state."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch

    def one_batch(m, cams, it):
        utils.set_cur_iter(it)
        m.update_learning_rate(it)
        clm_offload_train_one_batch(m, _Scene, cams, m.parameters_grad_buffer, None, None,
                                    torch.cuda.Stream(), torch.Generator(device="cuda"))
        torch.cuda.synchronize()

    args, sc, cams = _setup("clm_offload", "hbm")
    m = _make("clm_offload", sc, args)
    one_batch(m, cams, 1)
    state = m.capture()               # (capture flushes: `state` is a consistent checkpoint after batch 1)
    one_batch(m, cams, 5)
    one_batch(m, cams, 9)             # steps of these two batches are waiting (deferred) when restore() arrives
    assert m.small_deferred and not m._small_def_clean()
    m.restore(state, args)            # same object
    assert m._small_def_clean() and m._sorted_tag is None
    one_batch(m, cams, 5)
    m.flush_lazy_rows()
    m.flush_small()
    m2 = type(m)(3)
    m2.restore(state, args)
    one_batch(m2, cams, 5)
    m2.flush_lazy_rows()
    m2.flush_small()
    for a, b in zip(m.all_parameters(), m2.all_parameters()):
        assert torch.equal(a.detach(), b.detach())


def test_order_calculation_invariants(dev):
    from clm_gs_amd.strategies.base_engine import calculate_filters
    from clm_gs_amd.strategies.clm_offload.engine import order_calculation
    args, sc, cams = _setup("clm_offload", "host")
    m = _make("clm_offload", sc, args)
    with torch.no_grad():
        filters, cam_ids, g_ids = calculate_filters(cams, m.get_xyz, m.get_opacity, m.get_scaling, m.get_rotation, return_ids=True)
    # same index sets as the oracle's packed projection (base_engine.py:36-73)
    vms = torch.stack([c.world_view_transform.t() for c in cams]).cpu()
    Ks = torch.stack([c.K for c in cams]).cpu()
    ref = O.fully_fused_projection(sc["xyz"].cpu(), None, torch.nn.functional.normalize(sc["rotation"]).cpu(),
                                   torch.exp(sc["scaling"]).cpu(), vms, Ks, W, H, packed=True)
    assert torch.equal(cam_ids.cpu(), ref[0]) and torch.equal(g_ids.cpu(), ref[1])
    gen = torch.Generator(device="cuda").manual_seed(1)
    fin, cams2, filters2, sparsity, order, cnt_h, cnt_d, cnt_g, vis, bitmap = order_calculation(
        list(filters), cams, N, BSZ, gen, args)
    assert len(fin) == BSZ + 1 and sum(f.numel() for f in fin) == N
    allrows = torch.cat([f.to(torch.int64) for f in fin]).sort().values
    assert torch.equal(allrows, torch.arange(N)), "finish lists must partition [0, N)"
    sets = [set(f.tolist()) for f in filters2]
    for i in range(BSZ - 1):
        assert cnt_d[i] == len(sets[i] & sets[i + 1])
        assert cnt_h[i] + cnt_d[i] == len(sets[i + 1]) and cnt_g[i] + cnt_d[i] == len(sets[i])
    # group k+1 = rows whose LAST use is micro-batch k
    for k in range(BSZ):
        want = sets[k] - set().union(*sets[k + 1:]) if k < BSZ - 1 else sets[k]
        assert set(fin[k + 1].tolist()) == want
    assert set(fin[0].tolist()) == set(range(N)) - set().union(*sets)


@pytest.mark.parametrize("strategy,residency", [("no_offload", "hbm"), ("clm_offload", "hbm"), ("clm_offload", "host"),
                                                ("clm_offload", "host_budget")])
def test_densify_and_prune_keeps_state_aligned(dev, strategy, residency):
    r = _one_step(strategy, residency)
    m = r["model"]
    n0 = m.get_xyz.shape[0]
    m.xyz_gradient_accum = torch.rand_like(m.xyz_gradient_accum) * 1e-3
    m.denom = torch.ones_like(m.denom)
    m.split_generator = torch.Generator(device="cuda").manual_seed(3)
    m.densify_and_prune(0.0002, 0.005, 30.0, None)
    n1 = m.get_xyz.shape[0]
    assert n1 != n0
    for t in (m._opacity, m._scaling, m._rotation, m.xyz_gradient_accum, m.denom, m.max_radii2D):
        assert t.shape[0] == n1
    if strategy == "clm_offload":
        assert m._parameters.shape == (n1, 48)
        st = m.optimizer.cpu_adam.state[m._parameters]
        assert st["exp_avg"].shape == (n1, 48) and st["exp_avg_sq"].shape == (n1, 48)
        for g in m.optimizer.gpu_adam.param_groups:
            st = m.optimizer.gpu_adam.state[g["params"][0]]
            assert st["exp_avg"].shape[0] == n1
    else:
        assert m._features_rest.shape == (n1, 15, 3)
        for g in m.optimizer.param_groups:
            assert m.optimizer.state[g["params"][0]]["exp_avg"].shape[0] == n1
    m.reset_opacity()
    assert m.get_opacity.max() <= 0.0100001
    # and the engine still runs on the resized model
    from clm_gs_amd import utils
    from clm_gs_amd.synthetic import nadir_cameras
    utils.set_cur_iter(5)
    cams = nadir_cameras(BSZ, N, W, H, 0.35, seed=0, device="cuda")
    g = torch.Generator().manual_seed(5)
    for c in cams:
        c.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
    if strategy == "clm_offload":
        from clm_gs_amd.strategies.clm_offload import clm_offload_eval_one_cam, clm_offload_train_one_batch
        losses, _, _ = clm_offload_train_one_batch(m, _Scene, cams, m.parameters_grad_buffer, None, None,
                                                   torch.cuda.Stream(), torch.Generator(device="cuda"))
        img = clm_offload_eval_one_cam(cams[0], m, None, _Scene)
        assert img.shape == (3, H, W)
    else:
        from clm_gs_amd.strategies.no_offload import baseline_accumGrads_impl
        losses, _ = baseline_accumGrads_impl(m, _Scene, cams, None)
    assert all(torch.isfinite(l) for l in losses)


@pytest.mark.parametrize("sparse", [False, True])
def test_naive_offload_equals_clm_offload_after_two_steps(dev, sparse):
    """Row f4: everything on the host, whole-model copies per batch -> same losses and the same
    parameters as clm_offload (hbm) after two optimizer steps; eval renders agree.  sparse_adam:
    the reference's strategies differ by construction (clm updates the small attributes with
    SelectiveAdam = no bias correction, naive with the bias-corrected host Adam), so there only the
    SH rows, the first losses and the untouched rows are compared."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.clm_offload import clm_offload_eval_one_cam, clm_offload_train_one_batch
    from clm_gs_amd.strategies.naive_offload import (GaussianModelNaiveOffload, naive_offload_eval_one_cam,
                                                     naive_offload_train_one_batch, render_single_image)
    args, sc, cams = _setup("clm_offload", "hbm", sparse)
    ref = _make("clm_offload", sc, args)
    comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
    ref_losses = []
    for it in (1, 1 + BSZ):
        utils.set_cur_iter(it)
        ref.update_learning_rate(it)
        l, order, _ = clm_offload_train_one_batch(ref, _Scene, cams, ref.parameters_grad_buffer, None, None, comm, gen)
        lo = [0.0] * BSZ
        for k, v in zip(order, l):
            lo[k] = v.item()
        ref_losses.append(lo)
        if it == 1:
            ref.flush_lazy_rows()  # the SH-row step of a batch is applied at the rows' next touch (or here)
            ref_shs_step1 = ref._parameters.detach().clone()

    args2 = utils.default_args(bsz=BSZ, sparse_adam=sparse)
    args2.naive_offload = True
    utils.set_args(args2)
    m = GaussianModelNaiveOffload(3)
    m.create_from_tensors(sc["xyz"].clone(), sc["shs48"].clone(), sc["scaling"].clone(), sc["rotation"].clone(),
                          sc["opacity"].clone(), spatial_lr_scale=1.0)
    m.active_sh_degree = 3
    m.training_setup(args2)
    assert not m._small.is_cuda and m._small.is_pinned() and m._parameters.is_pinned()
    assert m._xyz.shape == (N, 3) and m._features_rest.shape == (N, 15, 3)
    for step, it in enumerate((1, 1 + BSZ)):
        utils.set_cur_iter(it)
        m.update_learning_rate(it)
        losses, vis = naive_offload_train_one_batch(m, _Scene, cams, None, sparse_adam=sparse)
        assert (vis is not None) == sparse
        if step == 0 or not sparse:
            for u, v in zip(ref_losses[step], [x.item() for x in losses]):
                assert abs(u - v) < 2e-5
        if sparse and step == 0:
            untouched = ~vis.cpu()
            assert untouched.any() and torch.equal(m._small.detach()[untouched][:, :11],
                                                   torch.cat((sc["xyz"], sc["opacity"], sc["scaling"], sc["rotation"]), 1).cpu()[untouched])
            assert torch.equal(m._parameters.detach()[untouched], sc["shs48"].cpu()[untouched])
            frac = _frac_differs(m._parameters.detach().cuda(), ref_shs_step1, sc["shs48"].cuda(), 0.02)
            assert frac < 0.01, frac
    ref.flush_lazy_rows() if getattr(ref, "lazy_rows", False) else None
    torch.cuda.synchronize()
    init = sc
    names = () if sparse else (
        ("xyz", m._xyz, ref._xyz), ("opacity", m._opacity, ref._opacity), ("scaling", m._scaling, ref._scaling),
        ("rotation", m._rotation, ref._rotation), ("shs48", m._parameters, ref._parameters))
    for name, a, b in names:
        frac = _frac_differs(a.detach().cuda(), b.detach().cuda().reshape(a.shape), init[name].cuda().reshape(a.shape), 0.02)
        assert frac < 0.01, (name, frac)
    img_n = naive_offload_eval_one_cam(m, _Scene, cams[0], None)
    img_c = clm_offload_eval_one_cam(cams[0], ref, None, _Scene)
    assert sparse or (img_n - img_c).abs().max() < 5e-3
    assert torch.equal(render_single_image(m, _Scene, cams[0]), torch.clamp(img_n, 0, 1))


def test_naive_offload_densify_and_prune(dev):
    """Row f4: densification on the host-resident model -- pinned tables, gradients, host Adam state
    and device statistics stay aligned; the engine and eval still run afterwards."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.naive_offload import (GaussianModelNaiveOffload, naive_offload_eval_one_cam,
                                                     naive_offload_train_one_batch)
    args, sc, cams = _setup("clm_offload", "hbm", False)
    args.clm_offload, args.naive_offload = False, True
    m = GaussianModelNaiveOffload(3)
    m.create_from_tensors(sc["xyz"].clone(), sc["shs48"].clone(), sc["scaling"].clone(), sc["rotation"].clone(),
                          sc["opacity"].clone(), spatial_lr_scale=1.0)
    m.active_sh_degree = 3
    m.training_setup(args)
    utils.set_cur_iter(1)
    naive_offload_train_one_batch(m, _Scene, cams, None)
    n0 = m.get_xyz.shape[0]
    before = m._small.detach().clone()
    m.xyz_gradient_accum = torch.rand_like(m.xyz_gradient_accum) * 1e-3
    m.denom = torch.ones_like(m.denom)
    m.split_generator = torch.Generator(device="cuda").manual_seed(3)
    m.densify_and_prune(0.0002, 0.005, 30.0, None)
    n1 = m.get_xyz.shape[0]
    assert n1 != n0 and not m._xyz.is_cuda and m._small.is_pinned() and m._parameters.is_pinned()
    assert m._small.shape == (n1, 12) and m._parameters.shape == (n1, 48)
    assert m._small.grad.shape == (n1, 12) and m._parameters.grad.shape == (n1, 48)
    for opt, p in ((m.small_adam, m._small), (m.row_adam, m._parameters)):
        st = opt.state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
        assert opt.param_groups[0]["params"][0] is p
    for t in (m.xyz_gradient_accum, m.denom, m.max_radii2D):
        assert t.shape[0] == n1 and t.is_cuda
    assert m._features_rest.shape == (n1, 15, 3)
    # same surgery as the clm model on the same inputs: same surviving / new rows
    ref = _make("clm_offload", sc, utils.get_args())
    with torch.no_grad():
        ref._xyz.copy_(before[:, 0:3].cuda()); ref._opacity.copy_(before[:, 3:4].cuda())
        ref._scaling.copy_(before[:, 4:7].cuda()); ref._rotation.copy_(before[:, 7:11].cuda())
    torch.manual_seed(0)
    ref.xyz_gradient_accum = torch.rand_like(ref.xyz_gradient_accum) * 1e-3
    assert ref.get_xyz.shape[0] == n0
    m.reset_opacity()
    assert float(torch.sigmoid(m._opacity).max()) <= 0.0100001
    utils.set_cur_iter(5)
    losses, _ = naive_offload_train_one_batch(m, _Scene, cams, None)
    assert all(torch.isfinite(l) for l in losses)
    assert naive_offload_eval_one_cam(m, _Scene, cams[0], None).shape == (3, H, W)


@pytest.mark.parametrize("strategy,residency", [("clm_offload", "hbm"), ("clm_offload", "host"), ("clm_offload", "host_budget"),
                                                ("no_offload", "hbm")])
def test_spatial_sort_mid_training_is_a_pure_relabelling(dev, strategy, residency):
    """permute_rows / spatial_sort (rows along a Z-order curve of x, y) between two batches: parameters,
    both Adam moments and the densification statistics afterwards are the un-sorted run's, row for row
    under the permutation (same arithmetic per row; only the tie order of equal depths can differ)."""
    from clm_gs_amd import utils

    def run(sort_after_first):
        args, sc, cams = _setup(strategy, residency)
        m = _make(strategy, sc, args)
        comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
        perm = None
        for b in range(2):
            utils.set_cur_iter(1 + b * BSZ)
            m.update_learning_rate(1 + b * BSZ)
            if strategy == "no_offload":
                from clm_gs_amd.strategies.no_offload import baseline_accumGrads_impl
                baseline_accumGrads_impl(m, _Scene, cams, None)
                for p in m.all_parameters():
                    p.grad /= BSZ
                m.optimizer.step()
                m.optimizer.zero_grad(set_to_none=True)
            else:
                from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
                clm_offload_train_one_batch(m, _Scene, cams, m.parameters_grad_buffer, None, None, comm, gen)
            if b == 0 and sort_after_first:
                perm = utils.morton_order(m._xyz.detach())
                m.permute_rows(perm)
        torch.cuda.synchronize()
        if hasattr(m, "flush_lazy_rows"):
            m.flush_lazy_rows()
        out = {"xyz": m._xyz.detach().cpu(), "rot": m._rotation.detach().cpu(), "accum": m.xyz_gradient_accum.cpu(),
               "denom": m.denom.cpu(), "radii": m.max_radii2D.cpu()}
        if strategy == "no_offload":
            out["sh"] = m.get_features.detach().reshape(-1, 48).cpu()
            out["m_xyz"] = m.optimizer.state[m._xyz]["exp_avg"].cpu()
        else:
            out["sh"] = m._parameters.detach().cpu()
            out["m_sh"] = m.optimizer.cpu_adam.state[m._parameters]["exp_avg"].cpu()
            out["m_xyz"] = m.optimizer.gpu_adam.state[m._xyz]["exp_avg"].cpu()
        return out, perm

    ref, _ = run(False)
    got, perm = run(True)
    perm = perm.cpu()
    assert not torch.equal(perm, torch.arange(perm.numel()))
    assert torch.equal(got["denom"], ref["denom"][perm]) and torch.equal(got["radii"], ref["radii"][perm])
    for k in ref:
        assert rel_l2(got[k], ref[k][perm]) < 1e-5, k


def test_camera_without_intersections_trains_with_zero_gradients(dev):
    """A training camera that sees nothing (clmgs_rasterize_bwd rejected the empty emit_slot of a
    camera with zero tile intersections).  Two forms: a non-empty row set whose radii are all 0 (every
    Gaussian behind the camera) and an empty filter.  Both must run the whole forward + loss + backward,
    render the background-free black image and leave every gradient exactly zero."""
    from clm_gs_amd import fused
    args, sc, cams = _setup("clm_offload")
    m = _make("clm_offload", sc, args)
    cam = cams[0]
    with torch.no_grad():  # move the whole scene behind the camera (nadir camera looks down -z)
        m._xyz[:, 2] += 1.0e4
    m.invalidate_small_packed()
    for rows in (None, torch.empty(0, dtype=torch.int64, device="cuda")):
        for p in (m._xyz, m._opacity, m._scaling, m._rotation):
            p.grad = torch.zeros_like(p)
        V = N if rows is None else 0
        sh = m._parameters.data if rows is None else torch.empty((0, 48), device="cuda")
        g_sh = torch.zeros((max(V, 1), 48), device="cuda")[:V]
        from clm_gs_amd import _lib
        p = fused.camera_forward(m, cam, rows, sh, 0, None, cam.original_image)
        fused.camera_backward(m, p, g_sh, update_stats=False)
        assert V == 0 or _lib.STATS["n_emitted"][-1] == 0
        loss = fused.camera_loss(p)
        torch.cuda.synchronize()
        assert float(p.out.abs().max()) == 0.0 and 0.0 < loss.item() < 1.0
        assert float(g_sh.abs().sum()) == 0.0
        for q in (m._xyz, m._opacity, m._scaling, m._rotation):
            assert float(q.grad.abs().max()) == 0.0


@pytest.mark.parametrize("residency", ["hbm", "host", "host_budget"])
def test_reference_camera_order_reorders_and_reports_like_the_reference(dev, residency):
    """reference_camera_order=True: the batch's cameras are processed in order_calculation's TSP order (the
    reference's engine.py:135-298) and `ordered_cams` / `sparsity` / `losses` are reported in that order; the
    batch gradient does not depend on the order, so one optimizer step ends where the plain order ends."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.base_engine import select_filters
    from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
    from clm_gs_amd.strategies.clm_offload.engine import order_calculation
    res = {}
    for ref_order in (False, True):
        args, sc, cams = _setup("clm_offload", residency)
        args.reference_camera_order = ref_order
        m = _make("clm_offload", sc, args)
        comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
        with torch.no_grad():
            filters, _ = select_filters(cams, m._xyz.detach(), m._scaling.detach(), m._rotation.detach())
        want = order_calculation(list(filters), list(cams), N, BSZ, gen, args)[4]
        losses, order, sparsity = clm_offload_train_one_batch(m, _Scene, cams, m.parameters_grad_buffer, None, None, comm, gen)
        torch.cuda.synchronize()
        m.flush_lazy_rows()
        assert sorted(order) == list(range(BSZ))
        assert order == (want if ref_order else list(range(BSZ)))
        for k, s in zip(order, sparsity):
            assert abs(s - filters[k].numel() / float(N)) < 1e-9
        by_cam = {k: l.item() for k, l in zip(order, losses)}
        res[ref_order] = (by_cam, [t.detach().clone().cuda() for t in (m._xyz, m._opacity, m._scaling, m._rotation, m._parameters)])
    for k in range(BSZ):
        assert abs(res[True][0][k] - res[False][0][k]) < 1e-6
    for a, b, init in zip(res[True][1], res[False][1], (sc["xyz"], sc["opacity"], sc["scaling"], sc["rotation"], sc["shs48"])):
        assert _frac_differs(a, b.reshape(a.shape), init.cuda().reshape(a.shape), 0.02) < 0.01


def test_device_side_counts_equal_exact_sizes_and_survive_overflow(dev):
    """device_side_counts (fused.py): the consumers of a camera's intersection list run against a predicted
    capacity and read the count on the device; two batches must end bit for bit where the exact-size path ends --
    also when the prediction is far too small (isect_capacity_margin 0.5: every camera is verified, found over
    capacity and redone exactly before anything was accumulated)."""
    from clm_gs_amd import _lib, fused, utils
    from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
    res = {}
    for mode, margin in (("exact", None), ("device", 1.25), ("overflow", 0.5)):
        args, sc, cams = _setup("clm_offload")
        args.device_side_counts = mode != "exact"
        if margin:
            args.isect_capacity_margin = margin
            args.isect_capacity_floor = 4096 if margin > 1 else 0
        fused._CAPACITY.clear(); fused._CAP_HELD.clear()
        _lib.STATS["isect_capacity_redo"] = 0
        m = _make("clm_offload", sc, args)
        comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
        it, losses = 1, []
        for b in range(2):
            utils.set_cur_iter(it)
            m.update_learning_rate(it)
            l, _, _ = clm_offload_train_one_batch(m, _Scene, cams, m.parameters_grad_buffer, None, None, comm, gen)
            losses += [x.item() for x in l]
            it += BSZ
        torch.cuda.synchronize()
        m.flush_lazy_rows()
        res[mode] = (losses, [t.detach().clone() for t in (m._xyz, m._opacity, m._scaling, m._rotation, m._parameters)],
                     _lib.STATS["isect_capacity_redo"], len(_lib.STATS["n_isects"]))
    assert res["exact"][2] == 0 and res["device"][2] == 0 and res["overflow"][2] >= BSZ  # batch 1 predicts from batch 0
    for mode in ("device", "overflow"):
        assert all(abs(a - b) < 1e-6 for a, b in zip(res[mode][0], res["exact"][0]))
        for a, b in zip(res[mode][1], res["exact"][1]):
            assert torch.equal(a, b), mode
    fused._CAPACITY.clear(); fused._CAP_HELD.clear()


@pytest.mark.parametrize("staging", ["host", "host_batch"])
def test_host_speculative_prefetch_is_exact(dev, staging):
    """(both staging forms: per-camera windows, host_window.py, and the union of the batch)
    Host-resident mode with hint_next_batch: the next batch's untouched rows are brought up to date and shipped
    while the present batch renders, verified against the exact selection when the batch arrives.  Four batches
    (overlapping cameras, so late / staged / wasted rows all occur) must end BIT FOR BIT where the run without
    hints ends -- also with a wrong hint (dropped: its rows are un-stamped) and with a hint that is never used
    (flush drops it)."""
    from clm_gs_amd import _lib, utils
    from clm_gs_amd.strategies.clm_offload import clm_offload_eval_one_cam, clm_offload_train_one_batch
    from clm_gs_amd.strategies.clm_offload.engine import hint_next_batch
    from clm_gs_amd.synthetic import nadir_cameras
    res = {}
    for mode in ("plain", "hinted", "wrong"):
        args, sc, _ = _setup("clm_offload", staging)
        cams = nadir_cameras(5 * BSZ, N, W, H, 0.35, seed=11, device="cuda")
        g = torch.Generator().manual_seed(5)
        for c in cams:
            c.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
        batches = [cams[0:4], cams[2:6], cams[9:13], cams[4:8]]  # overlapping / disjoint / returning cameras
        m = _make("clm_offload", sc, args)
        comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
        _lib.STATS["host_late_rows"] = []
        it, losses = 1, []
        for b, batch in enumerate(batches):
            utils.set_cur_iter(it)
            m.update_learning_rate(it)
            if mode == "hinted" and b + 1 < len(batches):
                hint_next_batch(m, batches[b + 1])
            if mode == "wrong":
                hint_next_batch(m, cams[13:17] if b % 2 == 0 else (batches[b + 1] if b + 1 < len(batches) else cams[16:20]))
            l, _, _ = clm_offload_train_one_batch(m, _Scene, batch, m.parameters_grad_buffer, None, None, comm, gen)
            losses += [x.item() for x in l]
            it += BSZ
        late = list(_lib.STATS["host_late_rows"])
        img = clm_offload_eval_one_cam(cams[1], m, None, _Scene)  # drops an outstanding speculation, then reads rows
        m.flush_lazy_rows()
        torch.cuda.synchronize()
        st = m.optimizer.cpu_adam.state[m._parameters]
        res[mode] = (losses, [t.detach().clone() for t in (m._xyz, m._opacity, m._scaling, m._rotation, m._parameters,
                                                           st["exp_avg"], st["exp_avg_sq"])], img.clone(), late,
                     m._host_g_step.clone(), m._host_last_step.clone())
    touched = res["plain"][3]
    # from the second batch on only the late rows go through the feeder (per-camera windows: the rows of the hinted batch's
    # FIRST camera are staged early -- none when the previous batch touched them all; batch-wide staging: the whole batch's)
    late_h = res["hinted"][3]
    assert late_h[0] == touched[0] and all(a <= b for a, b in zip(late_h[1:], touched[1:])), (late_h, touched)
    assert sum(late_h[1:]) < sum(touched[1:]), (late_h, touched)
    if staging == "host_batch":
        assert all(a < b for a, b in zip(late_h[1:], touched[1:])), (late_h, touched)
    for mode in ("hinted", "wrong"):
        assert res[mode][0] == res["plain"][0], mode
        for a, b in zip(res[mode][1], res["plain"][1]):
            assert torch.equal(a, b), mode
        assert torch.equal(res[mode][2], res["plain"][2])
        assert torch.equal(res[mode][5], res["plain"][5])                      # every row current as of the same step
        assert torch.equal(res[mode][4], res["plain"][4])                      # and no stamp left behind by a dropped hint


@pytest.mark.parametrize("staging", ["host", "host_batch", "host_budget"])
def test_host_staging_tables_grow_under_an_outstanding_speculation(dev, staging):
    """The staging tables are sized from the batches seen so far; a batch whose cameras see far more rows makes them grow
    while rows staged speculatively for it sit in the OLD tables: everything must then be treated as late, the untouched
    speculative rows un-stamped -- and the run must end bit for bit where the run without hints ends."""
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
    from clm_gs_amd.strategies.clm_offload.engine import hint_next_batch
    from clm_gs_amd.synthetic import nadir_cameras
    res = {}
    from clm_gs_amd.synthetic import synth_gaussians
    n, w_, h_ = 60_000, 160, 120  # (capacities are bucketed with a floor of 4 096 rows: the module's 3 000-row scene never grows)
    for mode in ("plain", "hinted"):
        over = {"host_staging": "batch"} if staging == "host_batch" else {}
        if staging == "host_budget":  # half of the rows resident in HBM: the re-allocation carries them to the new tables
            over = {"sh_hbm_budget_gb": 30_000 * 768 / 1e9 + 1e-9}
        args = utils.default_args(bsz=BSZ, sh_residency="host", **over)
        args.clm_offload = True
        utils.set_args(args)
        utils.set_img_size(h_, w_)
        utils.set_cur_iter(1)
        sc = synth_gaussians(n, seed=3, device="cuda")
        small = nadir_cameras(2 * BSZ, n, w_, h_, 0.05, seed=21, device="cuda")
        big = nadir_cameras(BSZ, n, w_, h_, 0.6, seed=22, device="cuda")
        g = torch.Generator().manual_seed(6)
        for c in small + big:
            c.original_image = (torch.rand(3, h_, w_, generator=g) * 255).to(torch.uint8).cuda()
        batches = [small[:BSZ], big, small[BSZ:], big]
        m = _make("clm_offload", sc, args)
        comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
        it, losses, gens = 1, [], []
        for b, batch in enumerate(batches):
            utils.set_cur_iter(it)
            m.update_learning_rate(it)
            if mode == "hinted" and b + 1 < len(batches):
                hint_next_batch(m, batches[b + 1])
            l, _, _ = clm_offload_train_one_batch(m, _Scene, batch, m.parameters_grad_buffer, None, None, comm, gen)
            losses += [x.item() for x in l]
            hb = getattr(m, "_hwin_bufs", None) or getattr(m, "_host_bufs", None)
            gens.append(hb["gen"])
            it += BSZ
        m.flush_lazy_rows()
        torch.cuda.synchronize()
        st = m.optimizer.cpu_adam.state[m._parameters]
        res[mode] = (losses, [t.detach().clone() for t in (m._xyz, m._opacity, m._scaling, m._rotation, m._parameters,
                                                           st["exp_avg"], st["exp_avg_sq"])],
                     m._host_g_step.clone(), m._host_last_step.clone(), gens)
    assert res["hinted"][4][1] > res["hinted"][4][0], res["hinted"][4]  # the tables DID grow at the big batch
    assert res["hinted"][0] == res["plain"][0]
    for a, b in zip(res["hinted"][1], res["plain"][1]):
        assert torch.equal(a, b)
    assert torch.equal(res["hinted"][2], res["plain"][2]) and torch.equal(res["hinted"][3], res["plain"][3])
    if staging == "host_budget":  # ... and where the run without a budget ends (same arithmetic on the other processor)
        res["budget"] = res["hinted"]
        args = utils.default_args(bsz=BSZ, sh_residency="host")
        args.clm_offload = True
        utils.set_args(args)
        m = _make("clm_offload", synth_gaussians(n, seed=3, device="cuda"), args)
        small = nadir_cameras(2 * BSZ, n, w_, h_, 0.05, seed=21, device="cuda")
        big = nadir_cameras(BSZ, n, w_, h_, 0.6, seed=22, device="cuda")
        g = torch.Generator().manual_seed(6)
        for c in small + big:
            c.original_image = (torch.rand(3, h_, w_, generator=g) * 255).to(torch.uint8).cuda()
        comm, gen, it = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1), 1
        for batch in [small[:BSZ], big, small[BSZ:], big]:
            utils.set_cur_iter(it)
            m.update_learning_rate(it)
            clm_offload_train_one_batch(m, _Scene, batch, m.parameters_grad_buffer, None, None, comm, gen)
            it += BSZ
        m.flush_lazy_rows()
        torch.cuda.synchronize()
        st = m.optimizer.cpu_adam.state[m._parameters]
        for a, b, name in zip(res["budget"][1][4:], (m._parameters, st["exp_avg"], st["exp_avg_sq"]), ("sh", "m", "v")):
            assert rel_l2(a.cuda(), b.detach().cuda()) < 2e-6, name


def test_deferred_small_adam_equals_eager(dev):
    """Round 5: xyz / opacity / scaling / rotation stepped per block of 256 rows, only when one of the batch's cameras
    may see the block (GaussianModelCLMOffload.small_deferred, clmgs_adam_small_deferred) == the eager dense Adam of
    every batch: 24 batches of moving cameras over a scene of 60 000 rows in Z-order (most blocks wait for several
    batches, some for the full clmgs_small_deferred_kmax() steps and are then forced), a per-image learning-rate
    schedule, a densification + re-sort and an opacity reset in the middle -- the visibility FILTERS of every batch, the
    losses, the four tensors, their Adam moments, the SH rows and the densification statistics end bit-identical; and
    blocks really were skipped (otherwise the test proves nothing)."""
    from clm_gs_amd import _lib, utils
    from clm_gs_amd.densification import gsplat_densification
    from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload, clm_offload_train_one_batch
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
    n, w, h, nb = 60_000, 160, 120, 24
    outs, skipped = [], None
    for deferred in (True, False):
        torch.manual_seed(0)  # densify_and_split draws from the global generator: both runs must draw the same numbers
        args = utils.default_args(bsz=BSZ, sh_residency="hbm", deferred_small_adam=deferred, densify_from_iter=0,
                                  densification_interval=BSZ * 10, densify_until_iter=10 ** 6, opacity_reset_interval=BSZ * 15,
                                  densify_grad_threshold=0.00002, position_lr_max_steps=200)
        args.clm_offload = True
        utils.set_args(args)
        utils.set_img_size(h, w)
        utils.set_cur_iter(1)
        sc = synth_gaussians(n, seed=2, device="cuda")
        order = utils.morton_order(sc["xyz"])
        for k in ("xyz", "scaling", "rotation", "opacity", "shs48"):
            sc[k] = utils.gather_rows(sc[k], order)
        m = GaussianModelCLMOffload(3)
        m.create_from_tensors(sc["xyz"], sc["shs48"], sc["scaling"], sc["rotation"], sc["opacity"], spatial_lr_scale=2.0)
        m.active_sh_degree = 3
        m.training_setup(args)
        assert m.small_deferred == deferred
        m.fuse_sort_into_prune = True
        cams = nadir_cameras(nb * BSZ, n, w, h, 0.05, seed=6, device="cuda")   # small footprints: most blocks are far away
        g = torch.Generator().manual_seed(8)
        for c in cams:
            c.original_image = (torch.rand(3, h, w, generator=g) * 255).to(torch.uint8).cuda()
        comm = torch.cuda.Stream()
        log, behind = [], []
        it = 1
        for b in range(nb):
            utils.set_cur_iter(it)
            m.update_learning_rate(it)
            # a walk that returns: batches 0-7 move on, 8-15 revisit earlier places, 16-23 move on again
            sel = cams[(b % 8) * BSZ:(b % 8 + 1) * BSZ] if 8 <= b < 16 else cams[b * BSZ:(b + 1) * BSZ]
            losses, _, sparsity = clm_offload_train_one_batch(m, _Scene, sel, m.parameters_grad_buffer, None, None, comm,
                                                              torch.Generator(device="cuda"))
            log.append(torch.stack(losses).clone())
            log.append(torch.tensor(sparsity))
            log.append(torch.tensor([int(_lib.STATS["touched_rows"][-1])]))
            if deferred:
                bl = m._small_def["blk_last"]
                behind.append(int((m.optimizer.cpu_adam.global_step - 1 - bl).max()))
            n0 = m.get_xyz.shape[0]
            gsplat_densification(it, _Scene, m, None)
            if m.get_xyz.shape[0] != n0:
                m.spatial_sort()
                log.append(torch.tensor([m.get_xyz.shape[0]]))
            it += BSZ
        if deferred:
            skipped = max(behind)
        m.flush_lazy_rows()
        torch.cuda.synchronize()
        st = m.optimizer.cpu_adam.state[m._parameters]
        outs.append(log + [m._xyz.detach().clone(), m._opacity.detach().clone(), m._scaling.detach().clone(),
                           m._rotation.detach().clone(), m._parameters.detach().clone(), st["exp_avg"].clone(),
                           m.small_packed().clone(), m.max_radii2D.clone(), m.xyz_gradient_accum.clone(), m.denom.clone()]
                    + [m.optimizer.gpu_adam.state[p][k].clone() for p in (m._xyz, m._opacity, m._scaling, m._rotation)
                       for k in ("exp_avg", "exp_avg_sq", "step")])
    assert skipped >= 8, skipped  # some block waited that many batches before a camera came near / it was forced
    assert len(outs[0]) == len(outs[1])
    for i, (a, b) in enumerate(zip(*outs)):
        assert a.shape == b.shape and torch.equal(a.cpu(), b.cpu()), i


def test_split_catch_up_equals_single_pass(dev):
    """Round 5: the deferred SH-row steps of a batch run camera by camera on a side stream (engine `split_catch_up`: the
    first camera waits for its own rows only) == one pass over the union of the touched rows: 12 batches of four
    overlapping cameras (rows shared by 2-4 cameras of a batch, rows that wait several batches), a densification +
    re-sort in the middle -- losses, SH rows, their moments and stamps, the small tensors: all bit-identical."""
    from clm_gs_amd import utils
    from clm_gs_amd.densification import gsplat_densification
    from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload, clm_offload_train_one_batch
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
    n, w, h, nb = 40_000, 160, 120, 12
    outs = []
    for split in (True, False):
        torch.manual_seed(0)
        args = utils.default_args(bsz=BSZ, sh_residency="hbm", split_catch_up=split, densify_from_iter=0,
                                  densification_interval=BSZ * 6, densify_until_iter=10 ** 6, opacity_reset_interval=10 ** 6,
                                  densify_grad_threshold=0.00002)
        args.clm_offload = True
        utils.set_args(args)
        utils.set_img_size(h, w)
        utils.set_cur_iter(1)
        sc = synth_gaussians(n, seed=4, device="cuda")
        order = utils.morton_order(sc["xyz"])
        for k in ("xyz", "scaling", "rotation", "opacity", "shs48"):
            sc[k] = utils.gather_rows(sc[k], order)
        m = GaussianModelCLMOffload(3)
        m.create_from_tensors(sc["xyz"], sc["shs48"], sc["scaling"], sc["rotation"], sc["opacity"], spatial_lr_scale=2.0)
        m.active_sh_degree = 3
        m.training_setup(args)
        m.fuse_sort_into_prune = True
        cams = nadir_cameras(nb * BSZ, n, w, h, 0.35, seed=9, device="cuda")   # large footprints: cameras of a batch overlap
        g = torch.Generator().manual_seed(8)
        for c in cams:
            c.original_image = (torch.rand(3, h, w, generator=g) * 255).to(torch.uint8).cuda()
        comm = torch.cuda.Stream()
        log, shared = [], 0
        it = 1
        for b in range(nb):
            utils.set_cur_iter(it)
            m.update_learning_rate(it)
            sel = cams[b * BSZ:(b + 1) * BSZ]
            losses, _, sparsity = clm_offload_train_one_batch(m, _Scene, sel, m.parameters_grad_buffer, None, None, comm,
                                                              torch.Generator(device="cuda"))
            log.append(torch.stack(losses).clone())
            log.append(torch.tensor(sparsity))
            n0 = m.get_xyz.shape[0]
            gsplat_densification(it, _Scene, m, None)
            if m.get_xyz.shape[0] != n0:
                m.spatial_sort()
            it += BSZ
        st = m.optimizer.cpu_adam.state[m._parameters]
        torch.cuda.synchronize()
        log += [m._row_last_step[:m.get_xyz.shape[0]].clone(), m._row_g_step[:m.get_xyz.shape[0]].clone()]  # (before the flush)
        m.flush_lazy_rows()
        torch.cuda.synchronize()
        outs.append(log + [m._parameters.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                           m._xyz.detach().clone(), m._opacity.detach().clone(), m._scaling.detach().clone(),
                           m._rotation.detach().clone()])
    assert len(outs[0]) == len(outs[1])
    for i, (a, b) in enumerate(zip(*outs)):
        assert a.shape == b.shape and torch.equal(a.cpu(), b.cpu()), i
    assert float(torch.stack([x for x in outs[0][1:2 * nb:2]]).sum()) > 0


@pytest.mark.parametrize("sparse", [False, True])
def test_host_rows_with_an_hbm_budget_end_where_the_host_rows_end(dev, sparse):
    """sh_hbm_budget_gb: rows [0, K) of the SH table live in HBM -- rendered from there, stepped
    there by the HBM engine's deferred row optimizer -- and only the rest goes through the host path.  Six batches with
    hints, an evaluation (reads the host table: the prefix is written back), a densification (structure changes: the device
    copy is dropped and loaded again) and a final flush must end where the plain host-resident run ends: the two row
    optimizers are the same arithmetic (host: C++, device: HIP; FMA contraction may differ in the last bit)."""
    from clm_gs_amd import _lib, utils
    from clm_gs_amd.strategies.clm_offload import clm_offload_eval_one_cam, clm_offload_train_one_batch
    from clm_gs_amd.strategies.clm_offload.engine import hint_next_batch
    from clm_gs_amd.synthetic import nadir_cameras
    res = {}
    for mode in ("host", "host_budget"):
        args, sc, cams = _setup("clm_offload", mode, sparse)
        more = nadir_cameras(3 * BSZ, N, W, H, 0.35, seed=3, device="cuda")
        g = torch.Generator().manual_seed(7)
        for c in more:
            c.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
        batches = [cams, more[:BSZ], more[BSZ:2 * BSZ], cams, more[2 * BSZ:], more[:BSZ]]
        m = _make("clm_offload", sc, args)
        comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
        it, losses, imgs = 1, [], []
        for b, cs in enumerate(batches):
            utils.set_cur_iter(it)
            m.update_learning_rate(it)
            if b + 1 < len(batches) and b not in (2, 3):
                hint_next_batch(m, batches[b + 1])
            l, _, _ = clm_offload_train_one_batch(m, _Scene, cs, m.parameters_grad_buffer, None, None, comm, gen)
            losses += [x.item() for x in l]
            it += BSZ
            if mode == "host_budget":
                px = m._hbm_prefix
                assert px is not None and px["dirty"] and not px["fill"] and px["K"] == 1500 == m._hwin_bufs["K"]
                assert _lib.STATS["host_touched_rows"][-1] < _lib.STATS["touched_rows"][-1]
            if b == 2:
                imgs.append(clm_offload_eval_one_cam(cams[1], m, None, _Scene).clone())
                if mode == "host_budget":
                    assert not m._hbm_prefix["dirty"]  # written back, still loaded
            if b == 3:
                m.xyz_gradient_accum = torch.full_like(m.xyz_gradient_accum, 1e-3) * (torch.arange(m._xyz.shape[0], device="cuda")[:, None] % 7 == 0)
                m.denom = torch.ones_like(m.denom)
                m.split_generator = torch.Generator(device="cuda").manual_seed(3)
                m.densify_and_prune(0.0002, 0.005, 30.0, None)
                if mode == "host_budget":
                    assert m._hbm_prefix is None  # dropped with the structure it described
        torch.cuda.synchronize()
        m.flush_lazy_rows()
        st = m.optimizer.cpu_adam.state[m._parameters]
        res[mode] = (losses, imgs, [t.detach().clone().cuda() for t in (m._xyz, m._opacity, m._parameters, st["exp_avg"], st["exp_avg_sq"])])
    assert res["host"][2][0].shape == res["host_budget"][2][0].shape
    for a, b in zip(res["host"][0], res["host_budget"][0]):
        assert abs(a - b) < 2e-6
    for a, b in zip(res["host"][1], res["host_budget"][1]):
        assert float((a - b).abs().max()) < 1e-5
    for a, b, name in zip(res["host"][2], res["host_budget"][2], ("xyz", "opacity", "sh", "m", "v")):
        assert rel_l2(a, b) < 2e-6, name
