"""Camera-DP pre-flight (clm_gs_amd/dp_preflight.py) on the CPU: one child per rank over gloo (stages 1-2: the raw
collectives and the exchange on a seeded table; stage 3, the tiny locality training, needs a GPU and runs in
tests/test_gpu_dp.py through bench.py).  A failure, a wrong answer and a hang must each come back as ok == False on
EVERY rank within the timeout -- that verdict is what bench.py's fallback to the all-reduce exchange is built on."""
import os
import tempfile
import threading

import pytest

from clm_gs_amd import dp_preflight


def _run(world, inject=None, timeout=90.0):
    old = os.environ.get("CLMGS_PREFLIGHT_INJECT")
    os.environ["CLMGS_TEST_HOOKS"] = "1"  # (dp_preflight honours the injection only with this set)
    if inject:
        os.environ["CLMGS_PREFLIGHT_INJECT"] = inject
    else:
        os.environ.pop("CLMGS_PREFLIGHT_INJECT", None)
    try:
        wd = tempfile.mkdtemp(prefix="clmgs_pf_test_")
        reps = [None] * world

        def work(r):
            reps[r] = dp_preflight.run(r, world, "gloo", -1, wd, timeout)
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return reps
    finally:
        if old is None:
            os.environ.pop("CLMGS_PREFLIGHT_INJECT", None)
        else:
            os.environ["CLMGS_PREFLIGHT_INJECT"] = old


@pytest.mark.parametrize("world", [2, 3])
def test_preflight_passes_over_gloo(world):
    reps = _run(world)
    assert all(r["ok"] for r in reps), reps
    assert all(set(r["stages_s"]) == {"raw_collectives", "exchange_on_seeded_table"} for r in reps), reps


@pytest.mark.parametrize("inject,needle", [("raise", "injected failure"), ("corrupt", "not its owner's")])
def test_preflight_reports_a_failing_or_wrong_collective_on_every_rank(inject, needle):
    reps = _run(2, inject)
    assert not any(r["ok"] for r in reps), reps
    assert needle in reps[0]["error"], reps
    assert "rank 0's pre-flight failed" in reps[1]["error"] or reps[1]["error"], reps  # the peer does not wait for the timeout
    assert max(r["wall_s"] for r in reps) < 60


def test_preflight_hang_is_cut_by_the_timeout():
    reps = _run(2, "hang", timeout=15.0)
    assert not any(r["ok"] for r in reps), reps
    assert all("timeout" in (r["error"] or "") or "failed" in (r["error"] or "") for r in reps), reps
    assert max(r["wall_s"] for r in reps) < 40
