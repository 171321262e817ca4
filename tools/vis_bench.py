import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clm_gs_amd import gsplat as G, _lib
from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
N, W, H = 28_000_000, 4608, 3456
sc = synth_gaussians(N, seed=0, device="cuda")
cams = nadir_cameras(8, N, W, H, 0.1, seed=0, device="cuda")
for C in (1, 2, 4, 8):
    Ks = torch.stack([c.K for c in cams[:C]]); vms = torch.stack([c.world_view_transform.t() for c in cams[:C]])
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f, u = G.visibility_select(sc["xyz"], sc["rotation"], sc["scaling"], vms, Ks, W, H)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("select C", C, "ms", round(dt * 1e3, 3), "vis0", f[0].numel(), "union", u.numel(), flush=True)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = G.visibility_radii(sc["xyz"], sc["rotation"], sc["scaling"], vms, Ks, W, H, raw=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("radii  C", C, "ms", round(dt * 1e3, 3), flush=True)
