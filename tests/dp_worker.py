"""Worker for tests/test_gpu_dp.py (run under torch.distributed.run, 2 ranks sharing cuda:0).

Both ranks hold a replica and train `STEPS` batches of `BSZ` cameras each (rank r takes cameras
r, r+2, ... of every global batch).  Rank 0 then also trains a fresh replica alone on the same
global batches (bsz = 2*BSZ) and the three parameter sets are compared.  The process group is
gloo so that two ranks can share the one GPU of the test box (RCCL refuses duplicate devices);
the exchange code is backend agnostic.
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

W, H, N, BSZ, STEPS = 96, 64, 4000, 4, 3
VIS = 0.35
if os.environ.get("CLMGS_DPW_SIZE"):  # "W,H,N,BSZ,STEPS,VIS": the full-size run of tests/test_gpu_dp.py
    _w, _h, _n, _b, _s, _v = os.environ["CLMGS_DPW_SIZE"].split(",")
    W, H, N, BSZ, STEPS, VIS = int(_w), int(_h), int(_n), int(_b), int(_s), float(_v)


class _Scene:
    cameras_extent = 30.0


def _model(sc, args):
    from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload
    m = GaussianModelCLMOffload(3)
    m.create_from_tensors(sc["xyz"].clone(), sc["shs48"].clone(), sc["scaling"].clone(),
                          sc["rotation"].clone(), sc["opacity"].clone(), spatial_lr_scale=1.0)
    m.active_sh_degree = 3
    m.training_setup(args)
    return m


def _train(m, batches, args, world=1):
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.clm_offload import clm_offload_train_one_batch
    comm = torch.cuda.Stream()
    gen = torch.Generator(device="cuda").manual_seed(1)
    it = 1
    for batch in batches:
        utils.set_cur_iter(it)
        m.update_learning_rate(it)
        clm_offload_train_one_batch(m, _Scene, batch, m.parameters_grad_buffer, None, None, comm, gen)
        it += len(batch) * world  # the image counter strides by the global batch
    m.flush_lazy_rows()
    torch.cuda.synchronize()
    # row moments: sharded by owner range under the locality exchange; row_moments_full() assembles them (collective)
    _train.info = {"small_owner": bool(getattr(m, "small_owner", False)),
                   "moments_sharded": bool(getattr(m, "moments_sharded", False)),
                   "moment_rows_held": int(m._exp_avg_buffer.shape[0]), "n": int(m._parameters.shape[0])}
    _train.moments = [t.clone() for t in m.row_moments_full()]
    if world > 1 and getattr(args, "dp_locality", False) and not args.sparse_adam and N <= 100000:
        # capture() / restore() with sharded row moments: capture assembles the full tables (a collective), restore
        # keeps this rank's shard of them -- the restored model holds the same state
        from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload
        state = m.capture()
        m2 = GaussianModelCLMOffload(3)
        m2.restore(state, args)
        same = all(torch.equal(a.detach(), b.detach()) for a, b in zip(m.all_parameters(), m2.all_parameters()))
        same &= all(torch.equal(a, b) for a, b in zip(_train.moments, m2.row_moments_full()))
        same &= m2.moments_sharded == m.moments_sharded and m2._mom_lo == m._mom_lo
        for p_, q_ in zip(m._small_tensors(), m2._small_tensors()):
            sa, sb = m.optimizer.gpu_adam.state[p_], m2.optimizer.gpu_adam.state[q_]
            same &= bool(torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]))
        _train.info["capture_restore_same"] = bool(same)
        del m2, state
    return [m._xyz.detach().clone(), m._opacity.detach().clone(), m._scaling.detach().clone(),
            m._rotation.detach().clone(), m._parameters.detach().clone()]


def trainer_mode(rank, world, owner=False, locality=False):
    """Both ranks run trainer.training (engine exchange + reduced densification statistics + the
    shared split generator); after clone / split / prune the replicas must still be identical."""
    import io
    from clm_gs_amd import trainer, utils
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
    n0, w, h = 6000, 128, 96
    args = utils.default_args(bsz=4, sh_residency="hbm", densify_from_iter=16, densification_interval=16,
                              densify_until_iter=48, densify_grad_threshold=0.00002, dp_owner_computes=owner,
                              dp_locality=locality)
    args.clm_offload = True
    utils.set_args(args)
    utils.set_img_size(h, w)
    sc = synth_gaussians(n0, seed=3, device="cuda")
    cams = nadir_cameras(16, n0, w, h, 0.35, seed=3, device="cuda")
    g = torch.Generator().manual_seed(9)
    for c in cams:
        c.original_image = (torch.rand(3, h, w, generator=g) * 255).to(torch.uint8).cuda()
    m = _model(sc, args)
    log = io.StringIO()
    trainer.training(m, _Scene, cams, [], log, iterations=64)
    shard_same = None
    if locality:
        # sharded row moments (the default) against the replicated tables: the same arithmetic on the same rows, so
        # the two runs must agree bit for bit through clone / split / prune / re-sort (shards rebuilt every time)
        assert m.moments_sharded and m._exp_avg_buffer.shape[0] <= m.parameters_buffer.shape[0] // world + 2
        m.flush_lazy_rows()
        args.dp_shard_moments = False
        m2 = _model(sc, args)
        assert not m2.moments_sharded
        trainer.training(m2, _Scene, cams, [], io.StringIO(), iterations=64)
        m2.flush_lazy_rows()
        shard_same = m2.get_xyz.shape[0] == m.get_xyz.shape[0] and all(
            bool(torch.equal(a.detach(), b.detach())) for a, b in
            zip((m._xyz, m._opacity, m._scaling, m._rotation, m._parameters),
                (m2._xyz, m2._opacity, m2._scaling, m2._rotation, m2._parameters)))
        full = m.row_moments_full()
        st2 = m2.optimizer.cpu_adam.state[m2._parameters]
        shard_same = bool(shard_same and torch.equal(full[0], st2["exp_avg"]) and torch.equal(full[1], st2["exp_avg_sq"]))
        del m2
    n = torch.tensor([m.get_xyz.shape[0]], device="cuda")
    n0t = n.clone()
    dist.broadcast(n0t, src=0)
    same = bool(n0t.item() == n.item())
    m.flush_lazy_rows()
    if same:
        for t in (m._xyz, m._opacity, m._scaling, m._rotation, m._parameters):
            other = t.detach().clone()
            dist.broadcast(other, src=0)
            same &= bool(torch.equal(other, t.detach()))
    flags = [None] * world
    dist.all_gather_object(flags, same)
    dist.barrier()
    if rank == 0:
        text = log.getvalue()
        print("DPRESULT " + json.dumps({
            "replicas_equal": all(flags), "n_before": n0, "n_after": int(n.item()),
            "global_stride": "iteration[1,9)" in text and "iteration[9,17)" in text,
            "split": "Number of split gaussians" in text, "sharded_equals_replicated_moments": shard_same}))
    dist.destroy_process_group()


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("DPW_WATCHDOG_S", "200")), exit=True)  # a hang prints its stacks
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    if len(sys.argv) > 1 and sys.argv[1].startswith("trainer"):
        return trainer_mode(rank, world, owner=sys.argv[1] == "trainer_owner", locality=sys.argv[1] == "trainer_locality")
    from clm_gs_amd import dp, utils
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians

    mode_arg = sys.argv[1] if len(sys.argv) > 1 else ""
    sparse = mode_arg.endswith("_sparse")
    args = utils.default_args(bsz=BSZ, sh_residency="hbm", sparse_adam=sparse)
    args.clm_offload = True
    for kv in filter(None, os.environ.get("CLMGS_DPW_OPTS", "").split(",")):  # e.g. "dp_small_owner=0,dp_small_refresh=2"
        k, v = kv.split("=")
        setattr(args, k, type(getattr(args, k))(int(v)))
    utils.set_args(args)
    utils.set_img_size(H, W)
    sc = synth_gaussians(N, seed=0, device="cuda")
    cams = nadir_cameras(STEPS * BSZ * world, N, W, H, VIS, seed=0, device="cuda")
    g = torch.Generator().manual_seed(5)
    for c in cams:
        c.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
    G = BSZ * world
    global_batches = [cams[s * G:(s + 1) * G] for s in range(STEPS)]

    if len(sys.argv) > 1 and sys.argv[1] == "owner":
        args.dp_owner_computes = True  # rows owned by index range: all-gather params / reduce-scatter grads
    wire = None
    if mode_arg.startswith("locality"):
        # locality exchange: rows in Z-order (index ranges = regions), cameras dealt to the rank owning most of
        # their rows; the global batch of a step is the union of the ranks' batches (the solo run trains on it)
        args.dp_locality = True
        order = utils.morton_order(sc["xyz"])
        for k in ("xyz", "scaling", "rotation", "opacity", "shs48"):
            sc[k] = utils.gather_rows(sc[k], order)
        probe = _model(sc, args)
        ranks_of, shares = dp.deal_cameras(cams, probe, world)
        del probe
        pools = [[c for c, q in zip(cams, ranks_of) if q == r] for r in range(world)]
        assert all(len(p) == STEPS * BSZ for p in pools), [len(p) for p in pools]
        per_rank = [[pools[r][s * BSZ:(s + 1) * BSZ] for s in range(STEPS)] for r in range(world)]
        global_batches = [sum((per_rank[r][s] for r in range(world)), []) for s in range(STEPS)]
        dp.reset_wire()
        mine = _train(_model(sc, args), per_rank[rank], args, world)
        wire = dp.wire_bytes()
        mine_info, mine_moments = _train.info, _train.moments
        local_share = float(sum(int(shares[c, q]) for c, q in enumerate(ranks_of)) / max(1, int(shares.sum())))
    else:
        mine = _train(_model(sc, args), [gb[rank::world] for gb in global_batches], args, world)
    # replicas identical bit for bit (same reduced gradients, same optimizer arithmetic)
    same = True
    for i, t in enumerate(mine):
        other = t.clone()
        dist.broadcast(other, src=0)
        eq = bool(torch.equal(other, t))
        if not eq:
            print(f"DPDIFF rank {rank} tensor {i}: max abs {float((other - t).abs().max()):.3e} "
                  f"rows differing {int(((other - t).abs().reshape(t.shape[0], -1).max(dim=1).values > 0).sum())}", flush=True)
        same &= eq
    flags = [None] * world
    dist.all_gather_object(flags, same)
    dist.barrier()
    dist.destroy_process_group()
    assert dp.world_size() == 1
    if rank != 0:
        return
    args1 = utils.default_args(bsz=G, sh_residency="hbm", sparse_adam=sparse)
    args1.clm_offload = True
    utils.set_args(args1)
    solo = _train(_model(sc, args1), global_batches, args1)
    err = []
    for a, b in zip(mine, solo):
        err.append(float((a - b).norm() / b.norm().clamp_min(1e-12)))
    res = {"replicas_equal": all(flags), "rel_l2_vs_single": err}
    if wire is not None:
        res.update(wire=wire, local_share=local_share, **mine_info)
        res["moments_rel_l2_vs_single"] = [float((a - b).norm() / b.norm().clamp_min(1e-30))
                                           for a, b in zip(mine_moments, _train.moments)]
    print("DPRESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
