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
print("DPPROBE " + json.dumps(res))
dist.destroy_process_group()
