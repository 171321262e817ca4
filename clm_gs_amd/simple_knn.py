"""simple_knn._C.distCUDA2 (scope row f3): mean squared distance to the 3 nearest neighbours,
used only to initialise log-scales (strategies/clm_offload/gaussian_model.py:60-63).  Host side:
bin the points into a uniform grid (~2 points per cell) and sort by cell with torch; the exact
shell search is the HIP kernel clmgs_knn3_mean_dist2."""
import torch

from . import _lib
from ._lib import check, dptr, stream


@torch.no_grad()
def distCUDA2(points):
    L = _lib.lib()
    pts = points.detach().float().contiguous()
    n = pts.shape[0]
    if n < 4:
        raise ValueError("distCUDA2 needs at least 4 points")
    lo, hi = pts.min(dim=0).values, pts.max(dim=0).values
    ext = torch.clamp(hi - lo, min=1e-6)
    vol = float(ext.prod())
    h = max((vol / n * 2.0) ** (1.0 / 3.0), float(ext.max()) / 1000.0)
    g = torch.clamp((ext / h).ceil().to(torch.int64), min=1)
    while int(g.prod()) > 64_000_000:  # keep the cell table bounded
        h *= 1.26
        g = torch.clamp((ext / h).ceil().to(torch.int64), min=1)
    gx, gy, gz = (int(v) for v in g)
    c = torch.clamp(((pts - lo) / h).to(torch.int64), min=torch.zeros(3, dtype=torch.int64, device=pts.device),
                    max=(g - 1).to(pts.device))
    cell = (c[:, 2] * gy + c[:, 1]) * gx + c[:, 0]
    cell_sorted, order = torch.sort(cell)
    pts_sorted = pts[order].contiguous()
    cell_start = torch.searchsorted(cell_sorted, torch.arange(gx * gy * gz + 1, device=pts.device)).to(torch.int32)
    out_sorted = torch.empty((n,), dtype=torch.float32, device=pts.device)
    check(L.clmgs_knn3_mean_dist2(stream(), n, dptr(pts_sorted, torch.float32), dptr(cell_start, torch.int32),
                                  float(lo[0]), float(lo[1]), float(lo[2]), float(h), gx, gy, gz,
                                  max(gx, gy, gz), dptr(out_sorted)))
    out = torch.empty_like(out_sorted)
    out[order] = out_sorted
    return out
