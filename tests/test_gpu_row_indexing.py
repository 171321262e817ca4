"""-m gpu: regression test for the row-selection helpers at > 2^25 rows.

torch 2.10 / ROCm 7: `t[idx]` / index_select with more than 2^26 indices into a table with 16-byte-multiple rows
writes only the first (len(idx) mod 2^26) output rows -- [102 231 360, 4]: the last 2^26 rows are garbage (one
64-thread workgroup per index against HIP's gridDim.x * blockDim.x <= 2^32; profiles/repro_index_defect.py is the
stand-alone repro, its output on MI355X is committed as profiles/r03_index_defect.json).  Every row permutation / selection / exchange of the package goes
through utils.take_rows / put_rows / gather_rows / select_rows (chunked); this test pins those helpers against
ANALYTIC tables at the failing size, and records -- without asserting -- whether the raw form is still broken,
so a fixed torch shows up as a message instead of a failure."""
import pytest
import torch

pytestmark = pytest.mark.gpu
M = 1 << 24


def _table(n, w):
    i = torch.arange(n, device="cuda", dtype=torch.int64)
    return torch.stack([((i * 7 + j) % M).to(torch.float32) for j in range(w)], dim=1).contiguous()


def _expected(idx, w):
    return torch.stack([((idx * 7 + j) % M).to(torch.float32) for j in range(w)], dim=1)


def _wrong(out, idx, w, chunk=1 << 24):
    return sum(int((out[a:a + chunk] != _expected(idx[a:a + chunk], w)).any(dim=1).sum())
               for a in range(0, idx.numel(), chunk))


@pytest.mark.parametrize("n,w", [(102_231_360, 4), (70_000_000, 12)])
def test_row_helpers_are_exact_beyond_2p25_rows(dev, n, w, record_property):
    from clm_gs_amd import utils
    t = _table(n, w)
    stride = 1_000_003
    idx = (torch.arange(n, device="cuda", dtype=torch.int64) * stride) % n   # every row once, scattered
    assert _wrong(utils.take_rows(t, idx), idx, w) == 0
    g = utils.gather_rows(t, idx)
    assert _wrong(g, idx, w) == 0
    back = torch.zeros_like(t)
    utils.put_rows(back, idx, g)
    assert torch.equal(back, t)
    del g, back
    mask = (torch.arange(n, device="cuda") % 5) < 3
    sel = utils.select_rows(t, mask)
    assert _wrong(sel, torch.nonzero(mask).flatten(), w) == 0
    del sel
    raw_wrong = _wrong(t[idx], idx, w)
    record_property("raw_advanced_indexing_wrong_rows", raw_wrong)
    print(f"[{n},{w}] raw t[idx]: {raw_wrong} wrong rows (0 = the stack is fixed; the helpers stay)")
