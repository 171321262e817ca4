"""CPU-only, world_size 2 over gloo: the camera-DP exchange (clm_gs_amd/dp.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clm_gs_amd import dp
    N = 1000
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.randn(N, d, generator=g) for d in (3, 1, 3, 4)]
    ref = [x.clone() for x in grads]
    touched = torch.zeros(N, dtype=torch.bool)
    touched[torch.randperm(N, generator=g)[:300]] = True
    rows = torch.zeros(N, 48)
    rows[touched] = torch.randn(int(touched.sum()), 48, generator=g)
    rows_ref, touched_ref = rows.clone(), touched.clone()
    dp.allreduce_small_grads(grads)
    tg = dp.allreduce_touched(touched)
    dp.allreduce_rows(rows, tg)

    class M:
        pass
    m = M()
    m.xyz_gradient_accum = torch.full((N, 1), float(rank + 1))
    m.denom = torch.full((N, 1), 1.0)
    m.max_radii2D = torch.full((N,), float(rank * 5))
    dp.allreduce_densify_stats(m)
    # packed one-collective exchange of several row tables over the global touched set
    t_small = torch.zeros(N, 12); t_rows = torch.zeros(N, 48)
    t_small[touched] = torch.randn(int(touched.sum()), 12, generator=g)
    t_rows[touched] = torch.randn(int(touched.sum()), 48, generator=g)
    ts_ref, tr_ref = t_small.clone(), t_rows.clone()
    rows_g = torch.nonzero(tg).flatten()
    a_small, a_rows = t_small.clone(), t_rows.clone()
    dp.allreduce_tables_rows([a_small, a_rows], rows_g, N)                    # packed branch
    b_small, b_rows = t_small.clone(), t_rows.clone()
    dp.allreduce_tables_rows([b_small, b_rows], rows_g, N, dense_above=0.0)   # in-place branch
    # owner-computes exchange: reduce-scatter of the globally touched rows to their owners (index ranges),
    # all-gather of the owners' rows
    pl = dp.owner_plan(rows_g, N)
    o_rows = t_rows.clone()
    dp.owner_reduce_rows(o_rows, pl)
    lo, hi = dp.owner_range(N)
    own = (rows_g >= lo) & (rows_g < hi)
    owner_ok = bool((o_rows[rows_g[~own]] == 0).all())          # handed over
    owner_ok &= pl.hi - pl.lo == int(own.sum()) and sum(pl.bounds[q + 1] - pl.bounds[q] for q in range(world)) == rows_g.numel()
    p_rows = torch.full((N, 48), float(rank + 1))                  # "parameters": owners hold value rank+1
    dp.owner_gather_rows(p_rows, pl)
    want_owner = torch.zeros(N, dtype=torch.long)
    for q in range(world):
        a, b = dp.owner_range(N, q, world)
        want_owner[a:b] = q + 1
    owner_ok &= bool((p_rows[rows_g] == want_owner[rows_g, None].float()).all())
    untouched = torch.ones(N, dtype=torch.bool); untouched[rows_g] = False
    owner_ok &= bool((p_rows[untouched] == float(rank + 1)).all())
    dense = [torch.full((N, 48), float(rank + 1)), torch.full((N,), float(rank + 1))]
    dp.owner_gather_dense(dense, N)
    owner_ok &= bool((dense[0] == want_owner[:, None].float()).all() and (dense[1] == want_owner.float()).all())
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(ref=ref, rows_ref=rows_ref, touched_ref=touched_ref,
                                          ts_ref=ts_ref, tr_ref=tr_ref, o_rows=o_rows, owner_ok=owner_ok))
    if rank == 0:
        want_small = [sum(gd["ref"][i] for gd in gathered) / world for i in range(4)]
        want_t = gathered[0]["touched_ref"] | gathered[1]["touched_ref"]
        want_rows = sum(gd["rows_ref"] for gd in gathered) / world
        ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(grads, want_small))
        ok &= torch.equal(tg, want_t)
        ok &= torch.allclose(rows, want_rows, atol=1e-6)
        want_ts = sum(gd["ts_ref"] for gd in gathered)
        want_tr = sum(gd["tr_ref"] for gd in gathered)
        for got_s, got_r in ((a_small, a_rows), (b_small, b_rows)):
            ok &= torch.allclose(got_s, want_ts, atol=1e-6) and torch.allclose(got_r, want_tr, atol=1e-6)
        ok &= bool((m.xyz_gradient_accum == 3.0).all() and (m.denom == 2.0).all() and (m.max_radii2D == 5.0).all())
        # the owners' reduced rows, put together, are the all-reduced table
        ok &= all(gd["owner_ok"] for gd in gathered)
        ok &= torch.allclose(sum(gd["o_rows"] for gd in gathered), want_tr, atol=1e-6)
        out.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_exchange_world2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True


def test_dp_is_noop_without_process_group():
    from clm_gs_amd import dp
    assert dp.world_size() == 1 and dp.rank() == 0
    t = torch.zeros(4, dtype=torch.bool)
    assert dp.allreduce_touched(t) is t
