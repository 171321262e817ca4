"""-m gpu: the RCCL (`nccl`) backend itself, with the one GPU a test box has: world_size 1 under
torch.distributed.run.  Every collective of clm_gs_amd/dp.py and the engine's exchange branch run on a
real RCCL communicator; bench.py's torchrun branch (process-group init, barrier, MAX over ranks, rank-0
JSON) runs with --gpus 1."""
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


def _torchrun(script_args, timeout=400, env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_dp_collectives_and_engine_exchange_on_rccl_world1(dev):
    out = _torchrun([os.path.join(ROOT, "tests", "nccl_worker.py")])
    line = [l for l in out.splitlines() if l.startswith("NCCLRESULT ")][-1]
    res = json.loads(line[len("NCCLRESULT "):])
    assert all(res.values()), res


def test_dp_exchanges_with_more_than_2p26_row_ids_on_rccl_world1(dev):
    """The index paths of every camera-DP exchange at the size where raw advanced indexing breaks on this stack
    (> 2^26 indices, profiles/r03_index_defect.json): 68 M row ids of a 70 M-row model through the all-reduce,
    owner-computes and locality exchanges on a 1-rank RCCL group -- every table must come back bit for bit."""
    out = _torchrun([os.path.join(ROOT, "tests", "nccl_worker.py"), "big"], timeout=600)
    line = [l for l in out.splitlines() if l.startswith("NCCLRESULT ")][-1]
    res = json.loads(line[len("NCCLRESULT "):])
    assert res["n_rows"] > (1 << 26)
    assert all(v for k, v in res.items() if k != "n_rows"), res


def test_bench_under_torchrun_world1_uses_rccl(dev):
    env = dict(os.environ, CLMGS_DP_FORCE="1")
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                     "--config", "small", "--no-cpu-baseline", "--prime-seconds", "0"], env=env)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["dist_backend"] == "nccl"


def test_preflight_child_runs_on_rccl_with_one_rank(dev):
    """The camera-DP pre-flight (clm_gs_amd/dp_preflight.py) on the REAL backend: a one-rank RCCL group (file-store
    rendezvous, device binding, the collectives as identities on the backend's own stream, the side-stream overlap of the
    tiny locality batches) -- everything about the first multi-GPU contact that a one-GPU box can exercise."""
    import tempfile
    from clm_gs_amd import dp_preflight
    rep = dp_preflight.run(0, 1, "nccl", 0, tempfile.mkdtemp(prefix="clmgs_pf_nccl_"), 240.0)
    assert rep["ok"], rep
    assert set(rep["stages_s"]) == {"raw_collectives", "exchange_on_seeded_table", "tiny_locality_training"}, rep
