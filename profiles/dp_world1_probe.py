"""Device time of every step of the locality exchange on a 1-rank process group at the headline size (no byte moves:
what the bookkeeping costs).  torchrun --nproc-per-node 1 profiles/dp_world1_probe.py [n_rows] [touched_fraction]"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ["CLMGS_DP_FORCE"] = "1"
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from clm_gs_amd import dp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 28_000_000
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.38
g = torch.Generator(device="cuda").manual_seed(0)
# touched rows: runs of consecutive ids, like a batch's union in Z-order
starts = torch.randint(0, N // 512, (int(N * frac) // 512,), device="cuda", generator=g).unique() * 512
touched = (starts[:, None] + torch.arange(512, device="cuda")[None, :]).reshape(-1)
params = torch.randn((N, 48), device="cuda")
g_sh = torch.randn((N, 48), device="cuda")
g_small = torch.randn((N, 12), device="cuda")
stamp = torch.zeros((N,), dtype=torch.int32, device="cuda")
step = 5
stamp[touched] = step
res = {"n_rows": N, "touched": int(touched.numel())}


def timed(name, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    res[name + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    return out


pl = timed("A_border_plan", lambda: dp.border_plan(touched, N))
timed("B_border_params_out", lambda: dp.border_params_out(params, pl))
timed("D_border_grads_home", lambda: dp.border_grads_home([g_sh, g_small], stamp, step, pl))
timed("F_publish_small", lambda: dp.publish_small(g_small, stamp, step, N, pl))
# step S (round 4, dp_small_owner): the drift-dilated candidate pass over ALL rows of the bench scene for a batch of 4
# cameras (own range empty = every row evaluated, the cost on a rank of a large world), the same with every row owned
# (what a 1-rank group pays), the three all_to_alls of an empty request, and the unpack of a million fetched lines
del params, g_sh
from clm_gs_amd import gsplat as G
from clm_gs_amd.synthetic import nadir_cameras, synth_gaussians
sc = synth_gaussians(N, seed=0, device="cuda")
cams = nadir_cameras(4, N, 4608, 3456, 0.10, seed=0, device="cuda")
Ks = torch.stack([c.K for c in cams])
vms = torch.stack([c.world_view_transform.t() for c in cams])
cand = timed("S_candidates_all_rows", lambda: G.visibility_candidates(sc["xyz"], sc["scaling"], vms, Ks, 4608, 3456,
                                                                      pos_margin=0.02, scale_gain=1.3, own_lo=0, own_hi=0))
res["S_candidates"] = int(cand.numel())
timed("S_candidates_all_owned", lambda: G.visibility_candidates(sc["xyz"], sc["scaling"], vms, Ks, 4608, 3456,
                                                                pos_margin=0.02, scale_gain=1.3, own_lo=0, own_hi=N))
pk = torch.zeros((N, 12), device="cuda")
none = torch.empty((0,), dtype=torch.int64, device="cuda")
timed("S_small_fetch_empty", lambda: dp.small_fetch(none, N, pk))
rows = cand[:1_000_000].contiguous()
lines = torch.randn((rows.numel(), 12), device="cuda")
tens = [sc["xyz"], sc["opacity"].reshape(N, 1).contiguous(), sc["scaling"], sc["rotation"]]
timed("S_scatter_1M_lines", lambda: dp.small_scatter(rows, lines, pk, tens))
print("DPPROBE " + json.dumps(res))
dist.destroy_process_group()
