"""Worker for tests/test_gpu_nccl.py: ONE rank under torch.distributed.run with the `nccl` backend
(= RCCL on ROCm) bound to cuda:0.  A 1-rank group runs every collective as the identity, so
(a) each dp.* collective must return its input unchanged, and (b) two training batches with the
exchange forced on (CLMGS_DP_FORCE=1) must leave bit-identical parameters to the same batches with no
process group in play.  This is the RCCL load / device-binding / stream-ordering check a single-GPU box
allows before the driver's multi-GPU scaling run."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

W, H, N, BSZ = 128, 96, 6000, 4


class _Scene:
    cameras_extent = 5.0


def _train(force, locality=False):
    from clm_gs_amd import utils
    from clm_gs_amd.strategies.clm_offload import GaussianModelCLMOffload, clm_offload_train_one_batch
    from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
    os.environ["CLMGS_DP_FORCE"] = "1" if force else "0"
    args = utils.default_args(bsz=BSZ, dp_locality=locality)
    args.clm_offload = True
    utils.set_args(args)
    utils.set_img_size(H, W)
    sc = synth_gaussians(N, seed=4, device="cuda")
    cams = nadir_cameras(2 * BSZ, N, W, H, 0.35, seed=4, device="cuda")
    g = torch.Generator().manual_seed(11)
    for c in cams:
        c.original_image = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8).cuda()
    m = GaussianModelCLMOffload(3)
    m.create_from_tensors(sc["xyz"], sc["shs48"], sc["scaling"], sc["rotation"], sc["opacity"], spatial_lr_scale=1.0)
    m.active_sh_degree = 3
    m.training_setup(args)
    comm, gen = torch.cuda.Stream(), torch.Generator(device="cuda").manual_seed(1)
    it = 1
    for b in range(2):
        utils.set_cur_iter(it)
        m.update_learning_rate(it)
        clm_offload_train_one_batch(m, _Scene, cams[b * BSZ:(b + 1) * BSZ], m.parameters_grad_buffer, None, None, comm, gen)
        it += BSZ
    torch.cuda.synchronize()
    m.flush_lazy_rows()
    return [t.detach().clone() for t in (m._xyz, m._opacity, m._scaling, m._rotation, m._parameters)]


def big_index_mode():
    """Every exchange of dp.py with MORE THAN 2^26 row ids in one call (the size at which raw `t[idx]` / index_select
    silently drop rows on this stack, profiles/repro_index_defect.py): a 1-rank RCCL group runs the collectives as the
    identity, so every table must come back unchanged -- through the pack / unpack index paths at full size."""
    from clm_gs_amd import dp
    os.environ["CLMGS_DP_FORCE"] = "1"
    n = 70_000_000
    g = torch.Generator(device="cuda").manual_seed(0)
    keep = torch.rand(n, generator=g, device="cuda") < 0.975
    idx = torch.nonzero(keep).flatten()
    assert idx.numel() > (1 << 26), idx.numel()
    del keep
    res = {"n_rows": int(idx.numel())}
    i64 = torch.arange(n, device="cuda", dtype=torch.int64)
    t4 = torch.stack([((i64 * 7 + j) % (1 << 24)).to(torch.float32) for j in range(4)], dim=1).contiguous()
    t12 = torch.stack([((i64 * 3 + j) % (1 << 24)).to(torch.float32) for j in range(12)], dim=1).contiguous()
    del i64
    a4, a12 = t4.clone(), t12.clone()
    dp.allreduce_tables_rows([a4, a12], idx, n, dense_above=2.0)     # packed path: take_rows / put_rows of 68 M rows
    res["tables_packed"] = bool(torch.equal(a4, t4) and torch.equal(a12, t12))
    dp.allreduce_rows(a12, None, average=False, rows=idx[: idx.numel() // 3])
    res["rows_packed"] = bool(torch.equal(a12, t12))
    pl = dp.owner_plan(idx, n)
    dp.owner_reduce_rows(a4, pl)
    dp.owner_gather_rows(a4, pl)
    res["owner_exchange"] = bool(torch.equal(a4, t4))
    del pl
    bp = dp.border_plan(idx, n)
    dp.border_params_out(a12, bp)
    stamp = torch.zeros(n, dtype=torch.int32, device="cuda")
    from clm_gs_amd import utils
    utils.fill_rows(stamp, idx, 9)
    dp.border_grads_home([a12, a4], stamp, 9, bp)
    counts = dp.publish_small(a12, stamp, 9, n, bp)
    own = dp.border_own_rows(bp)
    res["locality_exchange"] = bool(torch.equal(a12, t12) and torch.equal(a4, t4) and counts == [idx.numel()]
                                    and torch.equal(own, idx) and int((stamp == 9).sum()) == idx.numel())
    dist.barrier()
    dist.destroy_process_group()
    print("NCCLRESULT " + json.dumps(res))


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        return big_index_mode()
    from clm_gs_amd import dp
    assert dist.get_backend() == "nccl" and dp.world_size() == 1
    os.environ["CLMGS_DP_FORCE"] = "1"
    assert dp.active()
    res = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    n = 5000
    grads = [torch.randn(n, d, generator=g, device="cuda") for d in (3, 1, 3, 4)]
    ref = [x.clone() for x in grads]
    dp.allreduce_small_grads(grads, average=False)
    res["small_grads"] = all(torch.equal(a, b) for a, b in zip(grads, ref))
    touched = torch.rand(n, generator=g, device="cuda") < 0.3
    res["touched"] = bool(torch.equal(dp.allreduce_touched(touched), touched))
    rows = torch.randn(n, 48, generator=g, device="cuda")
    r0 = rows.clone()
    dp.allreduce_rows(rows, touched, average=False)
    res["rows_packed"] = bool(torch.equal(rows, r0))
    dp.allreduce_rows(rows, torch.ones_like(touched), average=False)
    res["rows_dense"] = bool(torch.equal(rows, r0))
    t12, t48 = torch.randn(n, 12, generator=g, device="cuda"), torch.randn(n, 48, generator=g, device="cuda")
    a12, a48 = t12.clone(), t48.clone()
    idx = torch.nonzero(touched).flatten()
    dp.allreduce_tables_rows([a12, a48], idx, n)
    dp.allreduce_tables_rows([a12, a48], idx, n, dense_above=0.0)
    res["tables"] = bool(torch.equal(a12, t12) and torch.equal(a48, t48))

    class M:
        pass
    m = M()
    m.xyz_gradient_accum, m.denom, m.max_radii2D = torch.rand(n, 1, device="cuda"), torch.ones(n, 1, device="cuda"), torch.rand(n, device="cuda")
    keep = (m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone())
    dp.allreduce_densify_stats(m)
    res["densify_stats"] = all(torch.equal(a, b) for a, b in zip(keep, (m.xyz_gradient_accum, m.denom, m.max_radii2D)))
    pl = dp.owner_plan(idx, n)  # owner-computes exchange on RCCL: reduce_scatter_tensor / all_gather_into_tensor
    b48 = t48.clone()
    dp.owner_reduce_rows(b48, pl)
    dp.owner_gather_rows(b48, pl)
    dense = [t48.clone(), torch.arange(n, device="cuda", dtype=torch.float32)]
    dp.owner_gather_dense(dense, n)
    res["owner_exchange"] = bool(torch.equal(b48, t48) and torch.equal(dense[0], t48) and pl.lo == 0 and pl.hi == idx.numel())
    # locality exchange on RCCL: all_to_all_single with split sizes, all_gather_into_tensor
    bp = dp.border_plan(idx, n)
    p48 = t48.clone()
    dp.border_params_out(p48, bp)
    stamp = torch.zeros(n, dtype=torch.int32, device="cuda")
    stamp[idx] = 5
    g48, g12 = t48.clone(), t12.clone()
    dp.border_grads_home([g48, g12], stamp, 5, bp)
    counts = dp.publish_small(g12, stamp, 5, n, bp)
    res["locality_exchange"] = bool(torch.equal(p48, t48) and torch.equal(g48, t48) and torch.equal(g12, t12)
                                    and bp.border.numel() == 0 and counts == [idx.numel()]
                                    and torch.equal(dp.border_own_rows(bp), idx))
    forced = _train(True)
    plain = _train(False)
    res["train_forced_equals_plain"] = all(torch.equal(a, b) for a, b in zip(forced, plain))
    local = _train(True, locality=True)
    res["train_locality_equals_plain"] = all(torch.equal(a, b) for a, b in zip(local, plain))
    dist.barrier()
    dist.destroy_process_group()
    print("NCCLRESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
