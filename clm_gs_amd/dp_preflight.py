"""Camera-DP pre-flight (net-new: the reference is single GPU).

The locality exchange (dp.py) leans on collectives with data-dependent, uneven split sizes.  Before a multi-GPU run
commits to it, every rank starts ONE short-lived child process of this module; the children form their OWN process
group (file-store rendezvous, same backend as the run), and on it they run

  1. the raw collectives the exchange is built from -- all_reduce, all_to_all_single with uneven (and empty) splits on
     int64 ids and on [rows, 48] float tables, reduce_scatter_tensor, all_gather_into_tensor -- against analytic values;
  2. dp.border_plan / border_params_out / border_grads_home (in parts, as the engine issues them) on a seeded table
     whose expected contents every rank can compute by itself;
  3. two tiny batches of clm_offload_train_one_batch under the locality exchange (Z-ordered rows, dealt cameras, side
     stream overlap, sharded moments, small attributes at their owners) followed by flush_lazy_rows(): the replicas'
     checksums must be identical on every rank.

A child writes one JSON report {ok, error, seconds per stage}.  The PARENT rank waits for its child with a timeout and
kills it when it expires: a collective that hangs, aborts the process or faults the GPU context takes the child down,
not the run.  bench.py then agrees on min(ok) over the ranks and falls back to the plain all-reduce exchange
(north_star's camera-DP) when any rank's pre-flight failed -- `dp.fallback` in its JSON line says so.

CLMGS_PREFLIGHT_INJECT (test hook, only with CLMGS_TEST_HOOKS=1): "raise" = the locality collective raises on rank 0, "corrupt" = rank 0's border
parameter rows arrive wrong, "hang" = rank 0 never returns from it.
"""
import argparse
import json
import os
import subprocess
import sys
import time


def _check(cond, what):
    if not cond:
        raise AssertionError("pre-flight: " + what)


def _raw_collectives(dist, torch, dev, rank, G):
    i64 = dict(dtype=torch.int64, device=dev)
    t = torch.tensor([rank + 1.0], device=dev)
    dist.all_reduce(t)
    _check(float(t.item()) == G * (G + 1) / 2.0, "all_reduce sum")
    # uneven all_to_all: rank r sends (r + q) % 3 + (q != r) elements to q, valued r * 1000 + q; nothing to itself
    n_to = lambda r, q: 0 if r == q else ((r + q) % 3 + 1)
    send = torch.cat([torch.full((n_to(rank, q),), rank * 1000 + q, **i64) for q in range(G)])
    recv_n = [n_to(q, rank) for q in range(G)]
    recv = torch.empty((sum(recv_n),), **i64)
    dist.all_to_all_single(recv, send, output_split_sizes=recv_n, input_split_sizes=[n_to(rank, q) for q in range(G)])
    want = torch.cat([torch.full((n_to(q, rank),), q * 1000 + rank, **i64) for q in range(G)])
    _check(bool(torch.equal(recv, want)), "all_to_all_single int64, uneven splits")
    rows_to = lambda r, q: 0 if r == q else 257 * ((r + 2 * q) % 4)  # some pairs exchange nothing at all
    send = torch.cat([torch.full((rows_to(rank, q), 48), float(rank * 100 + q), device=dev) for q in range(G)])
    recv_n = [rows_to(q, rank) for q in range(G)]
    recv = torch.empty((sum(recv_n), 48), device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=recv_n, input_split_sizes=[rows_to(rank, q) for q in range(G)])
    want = torch.cat([torch.full((rows_to(q, rank), 48), float(q * 100 + rank), device=dev) for q in range(G)])
    _check(bool(torch.equal(recv, want)), "all_to_all_single [rows,48] float, uneven / empty splits")
    buf = torch.arange(G * 8, dtype=torch.float32, device=dev).reshape(G * 8, 1).repeat(1, 48) * (rank + 1)
    mine = torch.empty((8, 48), device=dev)
    dist.reduce_scatter_tensor(mine, buf)
    want = torch.arange(rank * 8, rank * 8 + 8, dtype=torch.float32, device=dev).reshape(8, 1).repeat(1, 48) * (G * (G + 1) / 2.0)
    _check(bool(torch.equal(mine, want)), "reduce_scatter_tensor")
    allg = torch.empty((G * 8, 48), device=dev)
    dist.all_gather_into_tensor(allg, mine)
    _check(bool(torch.equal(allg[:, 0], torch.arange(G * 8, dtype=torch.float32, device=dev) * (G * (G + 1) / 2.0))),
           "all_gather_into_tensor")


def _exchange_on_seeded_table(dist, torch, dev, rank, G):
    """border_plan + B (two parts) + D (two parts) on tables every rank can predict."""
    from . import dp
    n = 6000 + 37 * G
    gen = lambda q: torch.Generator().manual_seed(1000 + q)
    touched_of = [torch.sort(torch.randperm(n, generator=gen(q))[:n // 3]).values for q in range(G)]
    owner_of = torch.zeros((n,), dtype=torch.int64)
    for q in range(G):
        lo, hi = dp.owner_range(n, q, G)
        owner_of[lo:hi] = q
    true_rows = (torch.arange(n, dtype=torch.float32)[:, None] + 0.5).repeat(1, 48)
    lo, hi = dp.owner_range(n, rank, G)
    table = torch.full((n, 48), -1.0)
    table[lo:hi] = true_rows[lo:hi]
    table = table.to(dev)
    mine = touched_of[rank].to(dev)
    first, last = mine[: mine.numel() // 4].contiguous(), mine[mine.numel() // 2:].contiguous()
    pl = dp.border_plan(mine, n, first_rows=first, last_rows=last, publish_counts=False)
    dp.border_params_out(table, pl, "params0")
    dp.border_params_out(table, pl, "params1")
    _check(bool(torch.equal(table[mine], true_rows.to(dev)[mine])), "border_params_out: a rendered row is not its owner's")
    # gradient lines: rank q contributes (q + 1) * (row + 1) on its touched rows, stamped with step 7
    step = 7
    g48 = torch.zeros((n, 48), device=dev)
    g12 = torch.zeros((n, 12), device=dev)
    stamp = torch.zeros((n,), dtype=torch.int32, device=dev)
    val = (mine.to(torch.float32) + 1.0) * (rank + 1)
    g48[mine] = val[:, None].repeat(1, 48)
    g12[mine] = val[:, None].repeat(1, 12)
    stamp[mine] = step
    r0 = dp.border_grads_send([g48, g12], stamp, step, pl, "grads0")
    r1 = dp.border_grads_send([g48, g12], stamp, step, pl, "grads1")
    dp.border_grads_apply([g48, g12], stamp, step, pl, r0)
    dp.border_grads_apply([g48, g12], stamp, step, pl, r1)
    want = torch.zeros((n,), dtype=torch.float32)
    for q in range(G):
        want[touched_of[q]] += (touched_of[q].to(torch.float32) + 1.0) * (q + 1)
    own = torch.arange(lo, hi)
    _check(bool(torch.equal(g48[lo:hi, 0].cpu(), want[own])) and bool(torch.equal(g12[lo:hi, 11].cpu(), want[own])),
           "border_grads_home: the owner's summed gradient lines")


def _tiny_locality_training(dist, torch, dev, rank, G):
    from . import dp, utils
    from .strategies.clm_offload import GaussianModelCLMOffload, clm_offload_train_one_batch
    from .synthetic import nadir_cameras, synth_gaussians
    n, w, h, bsz, steps = 60000, 256, 192, 4, 2
    args = utils.default_args(bsz=bsz, sh_residency="hbm", dp_locality=True)
    args.clm_offload = True
    utils.set_args(args)
    utils.set_img_size(h, w)
    sc = synth_gaussians(n, seed=11, device=dev)
    order = utils.morton_order(sc["xyz"])
    for k in ("xyz", "scaling", "rotation", "opacity", "shs48"):
        sc[k] = utils.gather_rows(sc[k], order)
    cams = nadir_cameras(steps * bsz * G, n, w, h, 0.3, seed=11, device=dev)
    g = torch.Generator().manual_seed(5)
    for c in cams:
        c.original_image = (torch.rand(3, h, w, generator=g) * 255).to(torch.uint8).to(dev)
    m = GaussianModelCLMOffload(3)
    m.create_from_tensors(sc["xyz"], sc["shs48"], sc["scaling"], sc["rotation"], sc["opacity"], spatial_lr_scale=1.0)
    m.active_sh_degree = 3
    m.training_setup(args)
    ranks_of, _ = dp.deal_cameras(cams, m, G)
    pool = [c for c, q in zip(cams, ranks_of) if q == rank]
    _check(len(pool) >= steps * bsz, "camera deal left a rank short")

    class _Scene:
        cameras_extent = 30.0
    comm = torch.cuda.Stream()
    gen = torch.Generator(device=dev).manual_seed(1)
    it = 1
    for s in range(steps):
        utils.set_cur_iter(it)
        m.update_learning_rate(it)
        clm_offload_train_one_batch(m, _Scene, pool[s * bsz:(s + 1) * bsz], m.parameters_grad_buffer, None, None, comm, gen)
        it += bsz * G
    m.flush_lazy_rows()
    torch.cuda.synchronize()
    sums = torch.stack([t.detach().double().sum() for t in (m._xyz, m._opacity, m._scaling, m._rotation, m._parameters)])
    _check(bool(torch.isfinite(sums).all()), "non-finite parameters after two locality batches")
    allsums = torch.empty((G * sums.numel(),), dtype=torch.float64, device=dev)  # (flat: gloo takes no [G, k] output)
    dist.all_gather_into_tensor(allsums, sums)
    allsums = allsums.view(G, sums.numel())
    _check(bool((allsums == allsums[0:1]).all()), "replicas differ after two locality batches + flush")
    return [float(x) for x in sums.tolist()]


def child_main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--store", required=True)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--device", type=int, default=0, help="GPU index; -1 = CPU tensors (gloo), stages 1-2 only")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    rep = {"ok": False, "rank": a.rank, "stages_s": {}, "error": None}
    t_all = time.perf_counter()
    try:
        import datetime

        import torch
        import torch.distributed as dist
        on_gpu = a.device >= 0  # --device -1: CPU tensors over gloo (stages 1-2 only; the CPU test of this module)
        if on_gpu:
            torch.cuda.set_device(a.device)
        dev = torch.device("cuda", a.device) if on_gpu else torch.device("cpu")
        kw = dict(init_method="file://" + a.store, rank=a.rank, world_size=a.world,
                  timeout=datetime.timedelta(seconds=60))
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(a.backend, **kw)
        # fault injection for the fallback tests: honoured only when CLMGS_TEST_HOOKS=1 is set as well (the test suite does)
        inject = os.environ.get("CLMGS_PREFLIGHT_INJECT", "") if os.environ.get("CLMGS_TEST_HOOKS") == "1" else ""
        if inject and a.rank == 0:
            real = dist.all_to_all_single
            calls = {"n": 0}

            def broken(out, inp, *args_, **kw_):
                calls["n"] += 1
                if calls["n"] > 2 and out.dim() == 2:  # the raw checks pass; the exchange's first table transfer does not
                    if inject == "raise":
                        raise RuntimeError("injected failure of the locality collective (CLMGS_PREFLIGHT_INJECT=raise)")
                    if inject == "hang":
                        time.sleep(10 ** 6)
                    r_ = real(out, inp, *args_, **kw_)
                    if inject == "corrupt" and out.numel():
                        out.add_(1.0)
                    return r_
                return real(out, inp, *args_, **kw_)
            dist.all_to_all_single = broken
        stages = [("raw_collectives", _raw_collectives), ("exchange_on_seeded_table", _exchange_on_seeded_table)]
        if on_gpu:
            stages.append(("tiny_locality_training", _tiny_locality_training))
        for name, fn in stages:
            t0 = time.perf_counter()
            out = fn(dist, torch, dev, a.rank, a.world)
            if on_gpu:
                torch.cuda.synchronize()
            rep["stages_s"][name] = round(time.perf_counter() - t0, 3)
            if name == "tiny_locality_training":
                rep["replica_checksums"] = out
        dist.barrier()
        rep["ok"] = True
        try:
            dist.destroy_process_group()
        except Exception:
            pass
    except BaseException as e:  # noqa: BLE001 -- everything is a verdict here, including SystemExit of a watchdog
        rep["error"] = f"{type(e).__name__}: {e}"[:500]
    rep["seconds"] = round(time.perf_counter() - t_all, 2)
    with open(a.out + ".tmp", "w") as f:
        json.dump(rep, f)
    os.replace(a.out + ".tmp", a.out)
    # a failed stage leaves peers inside collectives: do not wait for interpreter teardown of a wedged backend
    sys.stdout.flush()
    os._exit(0 if rep["ok"] else 17)


def run(rank, world, backend, device, workdir, timeout_s=150.0):
    """Parent side (one call per rank, all ranks with the same `workdir`): start this rank's child, wait for it at most
    `timeout_s`, kill it when that expires.  -> the child's report (ok False + error on timeout / crash)."""
    out = os.path.join(workdir, f"report_{rank}.json")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID", "GROUP_RANK",
              "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
              "CLMGS_DP_FORCE"):
        env.pop(k, None)  # the children rendezvous through the file store only
    if world == 1:
        env["CLMGS_DP_FORCE"] = "1"  # a one-rank group runs every collective as the identity (tests: RCCL on a 1-GPU box)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "clm_gs_amd.dp_preflight", "--rank", str(rank), "--world", str(world), "--store",
           os.path.join(workdir, "store"), "--backend", backend, "--device", str(device), "--out", out]
    t0 = time.perf_counter()
    err_path = os.path.join(workdir, f"stderr_{rank}.txt")
    with open(err_path, "w") as ef:
        proc = subprocess.Popen(cmd, env=env, stdout=ef, stderr=ef, cwd=root, start_new_session=True)
        timed_out, rc, peer_failed = False, None, None
        while True:
            rc = proc.poll()
            if rc is not None:
                break
            if time.perf_counter() - t0 > timeout_s:
                timed_out = True
            else:
                # a peer's child has already reported a failure: this rank's child is then stuck in a collective the
                # failed one will never join -- no point in waiting for the timeout (one node: the directory is shared)
                for q in range(world):
                    pq = os.path.join(workdir, f"report_{q}.json")
                    if q != rank and os.path.exists(pq):
                        try:
                            rq = json.load(open(pq))
                        except Exception:  # noqa: BLE001
                            continue
                        if not rq.get("ok", False):
                            peer_failed = (q, rq.get("error"))
                            break
            if timed_out or peer_failed is not None:
                try:
                    os.killpg(proc.pid, 9)
                except OSError:
                    proc.kill()
                proc.wait()
                rc = None
                break
            time.sleep(0.1)
    rep = None
    if os.path.exists(out):
        try:
            rep = json.load(open(out))
        except Exception:  # noqa: BLE001
            rep = None
    if rep is None:
        tail = ""
        try:
            tail = open(err_path).read()[-400:]
        except OSError:
            pass
        rep = {"ok": False, "rank": rank, "stages_s": {},
               "error": (f"timeout after {timeout_s:.0f} s (child killed)" if timed_out
                         else f"rank {peer_failed[0]}'s pre-flight failed ({peer_failed[1]}); child killed" if peer_failed
                         else f"child exited with code {rc} without a report: ...{tail}")}
    elif timed_out:
        rep["ok"], rep["error"] = False, (rep.get("error") or f"timeout after {timeout_s:.0f} s")
    rep["wall_s"] = round(time.perf_counter() - t0, 2)
    return rep


if __name__ == "__main__":
    child_main()
