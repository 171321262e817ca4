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


def _locality_worker(rank, world, port, out, mode="mixed"):
    """The four locality collectives against plain dense arithmetic: every rank 'renders' a touched set
    concentrated in its own range plus a spill-over into the other ranks' ranges.  Edge shapes (`mode`):
    "idle" = rank 0's cameras see nothing at all; "foreign" = the last rank sees ONLY other ranks' rows;
    "disjoint" = nobody leaves its own range (no border row anywhere)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clm_gs_amd import dp
    N, step = 3001, 7                                                                    # not a multiple of any world size used
    dp.reset_wire()
    lo, hi = dp.owner_range(N)
    g = torch.Generator().manual_seed(50 + rank)
    pick = torch.zeros(N, dtype=torch.bool)
    pick[lo + torch.randperm(hi - lo, generator=g)[: (hi - lo) // 2]] = True          # half of my own range
    if mode != "disjoint":
        pick[torch.randperm(N, generator=g)[:200]] = True                                # spill-over anywhere
    if mode == "idle" and rank == 0:
        pick[:] = False
    if mode == "foreign" and rank == world - 1:
        pick[lo:hi] = False
    T = torch.nonzero(pick).flatten()
    split = mode == "parts"
    first_rows, last_rows = (T[::3].clone(), T[1::2].clone()) if split else (None, None)
    so = mode == "small_owner"   # round 4: nothing is published (step F off), the candidates' lines are fetched (step S)
    pl = dp.border_plan(T, N, first_rows=first_rows, last_rows=last_rows, publish_counts=not so)
    fails = []
    if so:
        if pl.own_counts is not None:
            fails.append(20)
        # step S with the batch's border rows as the candidate list (ascending): the owners' current packed lines
        packed = torch.full((N, 12), -7.0)
        packed[lo:hi] = (torch.arange(lo, hi).float()[:, None] + torch.arange(12).float()[None, :] * 0.001)
        cand = torch.sort(pl.border).values
        lines = dp.small_fetch(cand, N, packed)
        if not torch.equal(lines, cand.float()[:, None] + torch.arange(12).float()[None, :] * 0.001):
            fails.append(21)
    if not (torch.equal(torch.sort(torch.cat((pl.mine, pl.border))).values, T)):
        fails.append(1)
    if not (bool(((pl.mine >= lo) & (pl.mine < hi)).all()) and bool(((pl.border < lo) | (pl.border >= hi)).all())):
        fails.append(2)
    if not (bool(((pl.serve_rows >= lo) & (pl.serve_rows < hi)).all())):
        fails.append(3)
    # B: parameters: every owner holds value (1000 * owner + row-dependent) in its range, garbage elsewhere
    owner_of = torch.zeros(N, dtype=torch.long)
    for q in range(world):
        a, b = dp.owner_range(N, q, world)
        owner_of[a:b] = q
    truth = (owner_of * 1000 + torch.arange(N) % 97).float()[:, None].repeat(1, 48)
    params = torch.full((N, 48), -1.0)
    params[lo:hi] = truth[lo:hi]
    if split:
        # the exchange in two parts: after part 0 exactly the first camera's border rows are current
        dp.border_params_out(params, pl, "params0")
        isb = (T < lo) | (T >= hi)
        in_first = torch.isin(T, first_rows)
        if not (torch.equal(params[T[isb & in_first]], truth[T[isb & in_first]])
                and bool((params[T[isb & ~in_first]] == -1.0).all())):
            fails.append(16)
        g0 = pl.border[pl.parts["grads0"].border_idx]
        if not (not bool(torch.isin(g0, last_rows).any())
                and g0.numel() + pl.parts["grads1"].border_idx.numel() == pl.border.numel()):
            fails.append(17)  # grads0 = the border rows the LAST camera does not touch
        dp.border_params_out(params, pl, "params1")
    else:
        dp.border_params_out(params, pl)
    if not (torch.equal(params[T], truth[T])):  # everything I render from is current
        fails.append(4)
    untouched_foreign = torch.ones(N, dtype=torch.bool)
    untouched_foreign[T] = False
    untouched_foreign[lo:hi] = False
    if not (bool((params[untouched_foreign] == -1.0).all())):  # nothing else travelled
        fails.append(5)
    # render: first-touch gradient tables (stale garbage + old stamps everywhere, fresh rows stamped `step`)
    stamp = torch.full((N,), 3, dtype=torch.int32)
    g_sh, g_small = torch.full((N, 48), 9.0), torch.full((N, 12), 9.0)      # 9.0 = stale content
    my_sh, my_small = torch.randn(T.numel(), 48, generator=g), torch.randn(T.numel(), 12, generator=g)
    g_sh[T], g_small[T], stamp[T] = my_sh, my_small, step
    dense_sh, dense_small = torch.zeros(N, 48), torch.zeros(N, 12)           # what an all-reduce would sum
    dense_sh[T], dense_small[T] = my_sh, my_small
    if mode == "undrawn":
        # filter rows the backward did not draw (radius 0 at render time): they are in the plan, but their lines are
        # an earlier step's (stale 9.0, old stamp) and must not reach anybody's sum
        dead = T[::7]
        g_sh[dead], g_small[dead], stamp[dead] = 9.0, 9.0, 3
        dense_sh[dead], dense_small[dead] = 0.0, 0.0
    if split:  # both slices travel first (the early one under the last camera's backward), then both are applied
        r0 = dp.border_grads_send([g_sh, g_small], stamp, step, pl, "grads0")
        r1 = dp.border_grads_send([g_sh, g_small], stamp, step, pl, "grads1")
        dp.border_grads_apply([g_sh, g_small], stamp, step, pl, r0)
        dp.border_grads_apply([g_sh, g_small], stamp, step, pl, r1)
    else:
        dp.border_grads_home([g_sh, g_small], stamp, step, pl)
    own_touched = dp.border_own_rows(pl)
    if not so:
        dp.publish_small(g_small, stamp, step, N, pl)
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(dense_sh=dense_sh, dense_small=dense_small, T=T))
    want_sh = sum(x["dense_sh"] for x in gathered)
    want_small = sum(x["dense_small"] for x in gathered)
    U = torch.zeros(N, dtype=torch.bool)
    for x in gathered:
        U[x["T"]] = True
    own_U = torch.nonzero(U[lo:hi]).flatten() + lo
    if not (torch.equal(own_touched, own_U)):
        fails.append(6)
    # what the optimizers consume: a row's line counts only under this step's stamp (first-touch policy)
    eff = lambda tab, rows_: torch.where((stamp[rows_] == step)[:, None], tab[rows_], torch.zeros(()))
    if not (torch.allclose(eff(g_sh, own_U), want_sh[own_U], atol=1e-6)):  # owners hold the summed SH gradient rows
        fails.append(7)
    if mode != "undrawn" and not (bool((stamp[own_U] == step).all())):
        fails.append(8)
    rows_U = torch.nonzero(U).flatten()
    if so:
        # the OWNERS hold the summed small rows (step D brought the lines home); nobody else was told anything
        if not (torch.allclose(eff(g_small, own_U), want_small[own_U], atol=1e-6)):
            fails.append(22)
    else:
        if not (torch.allclose(eff(g_small, rows_U), want_small[rows_U], atol=1e-6)):  # everybody holds the summed small rows
            fails.append(9)
        if mode != "undrawn" and not (bool((stamp[rows_U] == step).all())):
            fails.append(10)
    if not (bool((stamp[~U] == 3).all()) and bool((g_small[~U] == 9.0).all())):  # untouched rows: untouched
        fails.append(11)
    # wire accounting: the model of exchange_bytes == what the collectives counted
    acct = dp.exchange_bytes([x["T"] for x in gathered], N)
    w = dp.wire_bytes()
    if so:
        # the model's small-owner column minus its amortised refresh term (no refresh ran here)
        own_max = max(dp.owner_range(N, q, world)[1] - dp.owner_range(N, q, world)[0] for q in range(world))
        want_bytes = acct["locality_small_owner"][rank] - (world - 1) * 44.0 * own_max / 8.0
        if not (abs(w["total"] - want_bytes) < 1e-6 * max(1.0, w["total"])):
            fails.append(23)
    elif not (abs(w["total"] - acct["locality"][rank]) < 1e-6 * max(1.0, w["total"])):
        fails.append(12)
    if not (acct["union"] == int(U.sum()) and acct["border"][rank] == int(pl.border.numel())):
        fails.append(13)
    if not (acct["locality"][rank] < acct["allreduce"][rank]):
        fails.append(14)
    if mode == "disjoint" and not (pl.border.numel() == 0 and pl.serve_rows.numel() == 0):
        fails.append(15)
    res = [None] * world
    dist.all_gather_object(res, list(fails))
    if rank == 0:
        out.put(not any(res) or res)
    dist.barrier()
    dist.destroy_process_group()


def _run_world(target, world, *extra):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, out) + extra) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True


def test_dp_locality_exchange_world2_gloo():
    _run_world(_locality_worker, 2)


def test_dp_locality_exchange_world3_gloo():
    _run_world(_locality_worker, 3)


def test_dp_locality_exchange_edge_shapes_gloo():
    """A rank whose cameras see nothing, a rank that sees only foreign rows, and a batch without any border row."""
    _run_world(_locality_worker, 2, "idle")
    _run_world(_locality_worker, 3, "foreign")
    _run_world(_locality_worker, 4, "disjoint")


def test_dp_locality_exchange_in_parts_gloo():
    """Round 4: the exchange split so that it can hide behind rendering -- parameters of the first camera's border rows
    first (params0), the rest while it renders (params1); gradient lines of the border rows the last camera does not
    touch early (grads0), the rest after its backward (grads1) -- leaves every table exactly where the one-piece
    exchange leaves it (world 2 and 3, incl. the wire-byte account)."""
    _run_world(_locality_worker, 2, "parts")
    _run_world(_locality_worker, 3, "parts")


def test_dp_locality_exchange_small_attributes_at_their_owners_gloo():
    """Round 4 (dp_small_owner): the plan without the published counts, step S for the batch's border rows, B and D as
    before, no step F -- the owners hold the summed small-gradient lines, and the wire-byte model's
    `locality_small_owner` column equals what the collectives counted (world 2, 3, 4)."""
    for world in (2, 3, 4):
        _run_world(_locality_worker, world, "small_owner")


def test_dp_locality_exchange_ignores_rows_the_backward_did_not_draw():
    """a filter row with radius 0 at render time keeps an earlier step's gradient lines under first-touch
    stores; the exchange takes its SIZES from the plan but its CONTENT from the stamps -- stale lines travel as zeros."""
    _run_world(_locality_worker, 2, "undrawn")
    _run_world(_locality_worker, 3, "undrawn")


def test_assign_cameras_by_locality_is_balanced_and_local():
    from clm_gs_amd import dp
    # 12 cameras over 4 ranks; camera c sees mostly rank c % 4's range, a few see two ranges equally
    shares = torch.zeros(12, 4, dtype=torch.int64)
    for c in range(12):
        shares[c, c % 4] = 1000
        shares[c, (c + 1) % 4] = 100 + 10 * c
    shares[3] = torch.tensor([500, 500, 0, 0])
    deal = dp.assign_cameras(shares)
    assert sorted(deal.count(q) for q in range(4)) == [3, 3, 3, 3]
    kept = sum(int(shares[c, deal[c]]) for c in range(12))
    assert kept >= 0.9 * int(shares.max(dim=1).values.sum())
    # more first choices than room: the cameras that lose least are the ones moved
    shares = torch.tensor([[100, 0], [90, 80], [95, 10], [5, 50]])
    assert dp.assign_cameras(shares) == [0, 1, 0, 1]


def test_assign_cameras_guarantees_a_minimum_per_rank():
    """a cap of ceil(n / G) alone left the last ranks short (30 cameras / 8 ranks: ..., 2) or empty; every
    rank now gets between floor(n / G) and ceil(n / G) cameras whatever the preferences."""
    from clm_gs_amd import dp
    g = torch.Generator().manual_seed(0)
    for n, G in ((30, 8), (9, 8), (17, 4), (8, 8), (5, 8)):
        shares = torch.randint(0, 1000, (n, G), generator=g)
        shares[:, 0] += 5000  # everybody prefers rank 0
        deal = dp.assign_cameras(shares)
        counts = [deal.count(q) for q in range(G)]
        assert sum(counts) == n and min(counts) >= n // G and max(counts) <= -(-n // G), (n, G, counts)


def _moment_shard_worker(rank, world, port, out):
    """GaussianModelCLMOffload._redistribute_moments on CPU tensors: shards of two [n,48] tables follow their rows
    through an append (new rows: zero moments), a prune and a permutation, with owner ranges moving every time."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clm_gs_amd import dp
    from clm_gs_amd.strategies.clm_offload.gaussian_model import GaussianModelCLMOffload as GM

    class Fake:
        _moment_capacity = staticmethod(GM._moment_capacity)
    g = torch.Generator().manual_seed(11)  # the same tables and index maps on every rank
    n = 1003
    full_m, full_v = torch.randn(n, 48, generator=g), torch.rand(n, 48, generator=g)
    f = Fake()
    lo, hi = dp.owner_range(n)
    f.parameters_buffer = torch.empty(1200, 48)
    cap = GM._moment_capacity(1200)
    f._exp_avg_buffer, f._exp_avg_sq_buffer = torch.zeros(cap, 48), torch.zeros(cap, 48)
    f._exp_avg_buffer[:hi - lo], f._exp_avg_sq_buffer[:hi - lo] = full_m[lo:hi], full_v[lo:hi]
    f._mom_lo, f._mom_n = lo, n
    ok = True
    steps = [torch.cat((torch.arange(n), torch.full((57,), -1, dtype=torch.int64))),           # append 57 rows
             None, None]
    for k in range(3):
        if k == 0:
            idx = steps[0]
        elif k == 1:                                                                           # prune ~30 %, ascending
            idx = torch.nonzero(torch.rand(full_m.shape[0], generator=g) > 0.3).flatten()
        else:                                                                                  # re-sort
            idx = torch.randperm(full_m.shape[0], generator=g)
        want_m = torch.where(idx[:, None] >= 0, full_m[idx.clamp_min(0)], torch.zeros(1))
        want_v = torch.where(idx[:, None] >= 0, full_v[idx.clamp_min(0)], torch.zeros(1))
        GM._redistribute_moments(f, idx, idx.numel())
        full_m, full_v = want_m, want_v
        lo, hi = dp.owner_range(idx.numel())
        ok &= f._mom_lo == lo and f._mom_n == idx.numel() and f._exp_avg_buffer.shape[0] >= hi - lo
        ok &= bool(torch.equal(f._exp_avg_buffer[:hi - lo], full_m[lo:hi]))
        ok &= bool(torch.equal(f._exp_avg_sq_buffer[:hi - lo], full_v[lo:hi]))
        ok &= bool((f._exp_avg_buffer[hi - lo:] == 0).all())
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        out.put(all(flags))
    dist.destroy_process_group()


def test_sharded_row_moments_follow_their_rows_gloo():
    """VERDICT r3 item 7 (second half): m / v of the SH row table live at the owner of a row range only; append / prune /
    re-sort move the range borders and the shards are rebuilt by one all_to_all (world 2, 3, 4)."""
    for world in (2, 3, 4):
        _run_world(_moment_shard_worker, world)


def _small_fetch_worker(rank, world, port, out):
    """Step S of the small-attribute owner-computes exchange: a rank asks for foreign rows (random, incl. nothing at
    all on one rank), gets the owners' current packed lines in the order of its list, and small_scatter writes them to
    the mirror and the four tensors -- and to nothing else."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clm_gs_amd import dp
    N = 2003
    lo, hi = dp.owner_range(N)
    owner_of = torch.zeros(N, dtype=torch.long)
    for q in range(world):
        a, b = dp.owner_range(N, q, world)
        owner_of[a:b] = q
    truth = ((owner_of * 1000 + torch.arange(N) % 89).float()[:, None] + torch.arange(12).float()[None, :] * 0.01)
    truth[:, 11] = 0.0
    packed = torch.full((N, 12), -1.0)
    packed[lo:hi] = truth[lo:hi]                               # owners are current in their own range only
    tensors = [torch.full((N, w), -1.0) for w in (3, 1, 3, 4)]
    g = torch.Generator().manual_seed(70 + rank)
    pick = torch.zeros(N, dtype=torch.bool)
    if not (rank == 1 and world > 2):                          # one rank asks for nothing
        pick[torch.randperm(N, generator=g)[:300]] = True
    pick[lo:hi] = False
    rows = torch.nonzero(pick).flatten()
    dp.reset_wire()
    lines = dp.small_fetch(rows, N, packed)
    ok = bool(torch.equal(lines, truth[rows]))
    dp.small_scatter(rows, lines, packed, tensors)
    ok &= bool(torch.equal(packed[rows], truth[rows]))
    cat = torch.cat(tensors, dim=1)
    ok &= bool(torch.equal(cat[rows], truth[rows][:, :11]))
    rest = torch.ones(N, dtype=torch.bool)
    rest[rows] = False
    ok &= bool((cat[rest] == -1.0).all())
    rest[lo:hi] = False
    ok &= bool((packed[rest] == -1.0).all())
    w = dp.wire_bytes()
    ok &= w.get("all_to_all_small_ids", 0) == 8 * (world + rows.numel())
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        out.put(all(flags))
    dist.destroy_process_group()


def test_small_attribute_fetch_from_owners_gloo():
    """VERDICT r3 item 7 (first half): the small attributes live at the owner of a row range; step S fetches the current
    lines of a batch's candidate rows point to point (world 2, 3, 4)."""
    for world in (2, 3, 4):
        _run_world(_small_fetch_worker, world)
