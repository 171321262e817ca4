"""-m gpu: camera-DP end to end on the device (2 ranks sharing cuda:0, gloo process group).

The exchange (clm_gs_amd/dp.py) is net-new relative to the single-GPU reference, so
parity is argued the way the reference argues its own strategies (strategy-vs-strategy agreement):
2 ranks x bsz B must equal 1 rank x bsz 2B on the same cameras, and the replicas must not drift.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, timeout=300, env=None):
    """Runs a launcher in its own process group; a hang (two ranks sharing one GPU over gloo is an
    artificial set-up: seen once to stall at start-up) kills the whole group -- the workers' watchdog
    (faulthandler, tests/dp_worker.py) has printed their stacks by then -- and is retried ONCE.
    Assertion failures are never retried."""
    import signal
    last = ""
    for attempt in range(2):
        proc = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                                start_new_session=True)
        try:
            out, err = proc.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            out, err = proc.communicate()
            last = "TIMEOUT after %ds (attempt %d)\n" % (timeout, attempt) + out[-2000:] + err[-4000:]
            continue
        assert proc.returncode == 0, out[-3000:] + err[-3000:]
        return out
    raise AssertionError(last)


def test_two_ranks_equal_one_rank_with_double_batch(dev):
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                os.path.join(ROOT, "tests", "dp_worker.py")])
    line = [l for l in out.splitlines() if l.startswith("DPRESULT ")][-1]
    res = json.loads(line[len("DPRESULT "):])
    assert res["replicas_equal"] is True
    # float atomics order differs between the two schedules: tolerance, not bit equality
    assert max(res["rel_l2_vs_single"]) < 2e-4, res


def test_owner_computes_exchange_equals_allreduce_and_single_rank(dev):
    """dp_owner_computes (rows owned by index range, all-gather of parameter rows before
    rendering, reduce-scatter of gradient rows after it, only the owner steps a row): after the flush the
    replicas are identical and equal the single-rank run on the doubled batch."""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                os.path.join(ROOT, "tests", "dp_worker.py"), "owner"])
    line = [l for l in out.splitlines() if l.startswith("DPRESULT ")][-1]
    res = json.loads(line[len("DPRESULT "):])
    assert res["replicas_equal"] is True
    assert max(res["rel_l2_vs_single"]) < 2e-4, res


def test_trainer_owner_computes_densify_keeps_replicas_identical(dev):
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                os.path.join(ROOT, "tests", "dp_worker.py"), "trainer_owner"])
    line = [l for l in out.splitlines() if l.startswith("DPRESULT ")][-1]
    res = json.loads(line[len("DPRESULT "):])
    assert res["replicas_equal"] is True, res
    assert res["n_after"] != res["n_before"] and res["split"], res


@pytest.mark.parametrize("opts", ["", "dp_small_refresh=2", "dp_small_owner=0"])
def test_locality_exchange_equals_single_rank_with_global_batch(dev, opts):
    """dp_locality (dp.py "locality exchange"): Z-ordered rows, cameras dealt to the rank owning most of their
    rows, only border rows travel (all_to_all: parameters out, gradient rows back).  The small attributes are stepped
    by the owner of a row range (default; foreign copies go stale inside Adam's step bound, the candidates of every
    visibility pass are fetched from their owners -- also with the all-gather refresh every 2 batches), or,
    dp_small_owner=0, the owners publish the summed small-attribute gradients to everybody (step F).  After
    flush_lazy_rows() the replicas are identical and equal the single-rank run on the global batches (union of the
    ranks' batches)."""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                os.path.join(ROOT, "tests", "dp_worker.py"), "locality"], env=dict(os.environ, CLMGS_DPW_OPTS=opts))
    line = [l for l in out.splitlines() if l.startswith("DPRESULT ")][-1]
    res = json.loads(line[len("DPRESULT "):])
    assert res["replicas_equal"] is True, res
    assert max(res["rel_l2_vs_single"]) < 2e-4, res
    assert res["local_share"] > 0.5, res                      # the deal keeps most touched rows at home
    assert res["wire"]["all_to_all_grads"] > 0, res
    if opts == "dp_small_owner=0":
        assert res["small_owner"] is False and res["wire"]["all_gather_small"] > 1000, res
    else:
        assert res["small_owner"] is True and res["wire"]["all_to_all_small"] > 0, res
        assert res["wire"].get("all_gather_small", 0) == 0, res   # nothing is published
    # row moments live at the owner of a row range only: half the table per rank at 2 ranks, and
    # the assembled tables equal the single-rank run's
    assert res["moments_sharded"] is True and res["moment_rows_held"] <= res["n"] // 2 + 2, res
    assert res["capture_restore_same"] is True, res           # capture (collective) -> restore keeps the sharded state
    assert max(res["moments_rel_l2_vs_single"]) < 1e-3, res


@pytest.mark.parametrize("ranks", [2, 4])
def test_locality_exchange_at_28m_two_ranks_share_the_gpu(dev, ranks):
    """config 4's camera-DP half at FULL SIZE as far as one GPU allows -- 28 M Gaussians, 4608x3456, two
    (and four) ranks of bsz 4 sharing the device (gloo; 2-4 x 37 GB of replicas), the locality exchange with the cameras
    dealt by row ownership and the exchange in parts behind the first / last camera: after two global batches and
    flush_lazy_rows() the replicas are bit-identical and equal the single-rank run on the global batches (bsz 8 / 16);
    border rows and the candidates' small-attribute lines really travelled."""
    import torch
    if ranks * 40e9 + 45e9 > torch.cuda.get_device_properties(0).total_memory:
        pytest.skip("needs %d replicas of 37 GB + the single-rank run on one device" % ranks)
    # CLMGS_DP_DEBUG: the engine asserts, batch by batch, that every foreign row the exact visibility pass selected
    # was a candidate of step S (the drift-dilated cull is a superset at full size too)
    env = dict(os.environ, CLMGS_DPW_SIZE="4608,3456,28000000,4,2,0.10", CLMGS_DP_DEBUG="1")
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                os.path.join(ROOT, "tests", "dp_worker.py"), "locality"], timeout=1500, env=env)
    line = [l for l in out.splitlines() if l.startswith("DPRESULT ")][-1]
    res = json.loads(line[len("DPRESULT "):])
    assert res["replicas_equal"] is True, res
    assert max(res["rel_l2_vs_single"]) < 2e-4, res
    assert res["wire"]["all_to_all_params"] > 1e8 and res["wire"]["all_to_all_grads"] > 1e8, res  # > 100 MB of border rows
    # the small attributes are stepped at their owners: the second batch fetched its candidates' current lines (step S)
    # (rank 0 asked for, or served, more than a megabyte of them; which of the two depends on whose cameras look
    # across the range border in the second batch)
    assert res["small_owner"] is True, res
    assert res["wire"]["all_to_all_small_ids"] + res["wire"]["all_to_all_small"] > 1e6, res


def test_locality_exchange_sparse_adam_equals_single_rank(dev):
    """dp_locality with sparse_adam (BigCity as the reference scripts it, bigcity.sh:54-85: SelectiveAdam for the
    small attributes, the SH rows of visible Gaussians only): border rows by all_to_all, clearing-policy gradient
    tables, the owners' published rows double as the global visibility set."""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                os.path.join(ROOT, "tests", "dp_worker.py"), "locality_sparse"])
    line = [l for l in out.splitlines() if l.startswith("DPRESULT ")][-1]
    res = json.loads(line[len("DPRESULT "):])
    assert res["replicas_equal"] is True, res
    assert max(res["rel_l2_vs_single"]) < 2e-4, res
    assert res["wire"]["all_to_all_grads"] > 0 and res["wire"].get("all_reduce", 0) == 0, res


def test_trainer_locality_densify_keeps_replicas_identical(dev):
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                os.path.join(ROOT, "tests", "dp_worker.py"), "trainer_locality"])
    line = [l for l in out.splitlines() if l.startswith("DPRESULT ")][-1]
    res = json.loads(line[len("DPRESULT "):])
    assert res["replicas_equal"] is True, res
    assert res["n_after"] != res["n_before"] and res["split"], res
    # row moments sharded by owner range (default) == the replicated tables, bit for bit, through clone / split /
    # prune / re-sort (the shards are rebuilt by an all_to_all at every structural change)
    assert res["sharded_equals_replicated_moments"] is True, res


def test_trainer_two_ranks_densify_keeps_replicas_identical(dev):
    """trainer.training under camera-DP: global-batch image stride, reduced densification
    statistics, shared split samples -> bit-identical replicas after clone / split / prune."""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                os.path.join(ROOT, "tests", "dp_worker.py"), "trainer"])
    line = [l for l in out.splitlines() if l.startswith("DPRESULT ")][-1]
    res = json.loads(line[len("DPRESULT "):])
    assert res["replicas_equal"] is True, res
    assert res["n_after"] != res["n_before"] and res["split"], res
    assert res["global_stride"], res


def test_bench_two_ranks_one_gpu(dev):
    """bench.py's N>1 branch (barrier, max over ranks, rank-0 JSON) with 2 ranks on one device."""
    env = dict(os.environ, CLMGS_DIST_BACKEND="gloo", CLMGS_SHARE_GPU="1")
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                "--config", "small", "--prime-seconds", "0"], env=env)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["global_batch"] == 2 * j["config"]["bsz_per_gpu"]


def test_bench_plain_python_starts_its_own_ranks(dev):
    """The driver's command form: `python bench.py --gpus 2` with NO RANK / WORLD_SIZE in the environment must start
    its own two ranks (bench._self_launch -> torch.distributed.run) and print ONE line with n_gpus == 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CLMGS_DIST_BACKEND="gloo", CLMGS_SHARE_GPU="1")
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                "--config", "small", "--prime-seconds", "0"], env=env)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["dp"]["mode"] == "locality" and j["dp"]["dp_exchange_bytes_per_step"] > 0
    # round 5: the pre-flight ran on the run's backend and passed (all three stages), the per-phase device times of the
    # exchange, the replicas' checksums after the completing flush and the plain all-reduce leg are in the line
    d = j["dp"]
    assert d["fallback"] is None and d["preflight"]["ok"], d["preflight"]
    assert set(d["preflight"]["stages_s_rank0"]) == {"raw_collectives", "exchange_on_seeded_table", "tiny_locality_training"}
    assert d["replicas_equal"] is True and d["completing_flush_ms"] is not None
    assert {"plan", "S", "B0", "B1", "D0", "D1", "tail_exchange"} <= set(d["phase_ms"]), d["phase_ms"]
    ar = d["allreduce_leg"]
    assert ar["mode"] == "allreduce" and ar["value"] > 0 and ar["replicas_equal"] is True, ar


def test_bench_falls_back_to_allreduce_when_the_preflight_fails(dev):
    """the first contact with a multi-GPU backend must not be able to lose the run.  The locality
    collective of the pre-flight is made to fail on rank 0 (CLMGS_PREFLIGHT_INJECT): the run must still print ONE line
    with n_gpus == 2, in the plain all-reduce mode, and say why."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CLMGS_DIST_BACKEND="gloo", CLMGS_SHARE_GPU="1", CLMGS_PREFLIGHT_INJECT="raise", CLMGS_TEST_HOOKS="1")
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                "--config", "small", "--prime-seconds", "0"], env=env)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0
    d = j["dp"]
    assert d["mode"] == "allreduce" and d["fallback"]["from"] == "locality" and d["fallback"]["to"] == "allreduce", d
    assert "injected failure" in json.dumps(d["fallback"]["reason"]), d["fallback"]
    assert d["preflight"]["ok"] is False and d["allreduce_leg"] is None


def test_bench_prints_its_line_when_the_allreduce_leg_does_not_finish(dev):
    """The all-reduce leg is an appendix of the multi-GPU line: a watchdog per rank (--allreduce-timeout, here 50 ms,
    i.e. it fires while the leg is still building its model) makes rank 0 print the ONE line without the leg and every
    rank leave the process -- exit code 0, the locality measurement intact."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CLMGS_DIST_BACKEND="gloo", CLMGS_SHARE_GPU="1")
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                "--config", "small", "--prime-seconds", "0", "--allreduce-timeout", "0.05"], env=env)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["dp"]["mode"] == "locality" and j["dp"]["replicas_equal"] is True
    assert j["dp"]["allreduce_leg"]["value"] is None and "watchdog" in j["dp"]["allreduce_leg"]["error"]


def test_bench_refuses_more_ranks_than_gpus(dev):
    """Without the sharing hook, --gpus N on a box with fewer than N devices fails loudly (exit 2), it does not
    silently run one rank."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CLMGS_SHARE_GPU")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                           "--config", "small"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert proc.returncode == 2 and "visible" in proc.stderr, (proc.returncode, proc.stderr[-500:])
    assert not [l for l in proc.stdout.splitlines() if l.startswith("{")]
