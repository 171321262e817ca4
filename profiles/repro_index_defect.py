#!/usr/bin/env python
"""Minimal repro of the torch 2.10 / ROCm 7 advanced-indexing defect the engines work around
(clm_gs_amd/utils.py: take_rows / put_rows / gather_rows / select_rows).

    python profiles/repro_index_defect.py [--out gpurun_out/index_defect.json]

Row tables with ANALYTIC contents (t[i, j] = ((i * 7 + j) mod 2^24), exact in fp32), so a gathered row is
checked against a closed form and no second gather is trusted.  For every (rows N, width w) and two index
sets (a permutation-like strided walk over all rows, and a sorted ~60 % subset) it counts wrong output rows of

    raw        t[idx]                       (what model code used before the work-around)
    isel       torch.index_select(t, 0, idx)
    take       clm_gs_amd.utils.take_rows   (chunks of 2^23 indices: the product path)
    put        clm_gs_amd.utils.put_rows    (the scatter direction, checked the same way)

No reference involvement; nothing here is imported by the package."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clm_gs_amd import utils  # noqa: E402

M = 1 << 24


def table(n, w, dev):
    i = torch.arange(n, device=dev, dtype=torch.int64)
    cols = [((i * 7 + j) % M).to(torch.float32) for j in range(w)]
    return torch.stack(cols, dim=1).contiguous()


def expected(idx, w):
    return torch.stack([((idx * 7 + j) % M).to(torch.float32) for j in range(w)], dim=1)


def wrong_rows(out, idx, w, chunk=1 << 24):
    bad, first = 0, None
    for a in range(0, idx.numel(), chunk):
        e = expected(idx[a:a + chunk], w)
        m = (out[a:a + chunk] != e).any(dim=1)
        k = int(m.sum())
        if k and first is None:
            first = a + int(torch.nonzero(m)[0])
        bad += k
    return bad, first


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true", help="only the shapes known to fail + one control")
    a = ap.parse_args()
    dev = "cuda"
    shapes = [(102_231_360, 4), (102_231_360, 3), (40_000_000, 4)] if a.quick else \
        [(n, w) for n in ((1 << 25) + 1, 40_000_000, 70_000_000, 102_231_360) for w in (1, 3, 4, 12, 48)]
    res = []
    for n, w in shapes:
        t = table(n, w, dev)
        for kind in ("strided_all", "sorted_subset"):
            if kind == "strided_all":   # every row once, in a scattered order (gcd(stride, n) = 1)
                stride = 1_000_003
                while n % stride == 0:
                    stride += 2
                idx = (torch.arange(n, device=dev, dtype=torch.int64) * stride) % n
            else:
                g = torch.Generator(device=dev).manual_seed(0)
                idx = torch.nonzero(torch.rand(n, device=dev, generator=g) < 0.6).flatten()
            row = dict(rows=n, width=w, index_set=kind, n_indices=int(idx.numel()))
            for name, fn in (("raw", lambda: t[idx]), ("isel", lambda: torch.index_select(t, 0, idx)),
                             ("take", lambda: utils.take_rows(t, idx))):
                out = fn()
                torch.cuda.synchronize()
                bad, first = wrong_rows(out, idx, w)
                row[name + "_wrong_rows"] = bad
                if bad:
                    row[name + "_first_wrong_output_row"] = first
                del out
            if kind == "strided_all" and w <= 12:
                dst = torch.zeros_like(t)
                utils.put_rows(dst, idx, expected(idx, w) if n <= 50_000_000 else utils.take_rows(t, idx))
                torch.cuda.synchronize()
                row["put_wrong_rows"] = int((dst != t).any(dim=1).sum())
                del dst
            print(json.dumps(row), flush=True)
            res.append(row)
            del idx
        del t
        torch.cuda.empty_cache()
    summary = dict(torch=torch.__version__, hip=torch.version.hip, device=torch.cuda.get_device_name(0), cases=res,
                   defect_seen=any(r.get("raw_wrong_rows", 0) or r.get("isel_wrong_rows", 0) for r in res),
                   product_path_wrong=sum(r.get("take_wrong_rows", 0) + r.get("put_wrong_rows", 0) for r in res))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(summary, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "cases"}))
    return 0 if summary["product_path_wrong"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
