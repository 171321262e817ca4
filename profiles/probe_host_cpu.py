"""What limits the host thread pool on the GPU box: cgroup CPU quota, compute scaling, memory scaling."""
import ctypes, json, os, sys, time, subprocess
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clm_gs_amd import _lib
def sh(c):
    return subprocess.run(c, shell=True, capture_output=True, text=True).stdout.strip()
print(json.dumps({"cpu.max": sh("cat /sys/fs/cgroup/cpu.max 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
                  "nproc": sh("nproc"), "cpuset": sh("cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | head -c 200"),
                  "loadavg": sh("cat /proc/loadavg"), "governor": sh("cat /sys/devices/system/cpu/cpu0/cpufreq/scaling_governor 2>/dev/null"),
                  "mhz": sh("grep MHz /proc/cpuinfo | sort -k4 -n | sed -n '1p;$p' | tr '\\n' ' '"),
                  "dimms": sh("dmidecode -t memory 2>/dev/null | grep -c 'Size: [0-9]'")}))
L = _lib.lib(); P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
N, V = 12_000_000, 3_300_000
p, g, m, v = [torch.full((N, 48), 1e-3) for _ in range(4)]
stage = torch.empty(V, 48); last = torch.zeros(N, dtype=torch.int32); gs = torch.zeros(N, dtype=torch.int32)
rows = torch.randperm(N)[:V].sort().values.to(torch.int32).contiguous()
col_lr = torch.full((48,), 1e-3)
step = 0
for mode in ("copy_only", "one_step", "replay20"):
    for nt in (8, 16, 32, 64, 128):
        L.clmgs_host_pool_start(nt)
        best = 1e9
        for rep in range(2):
            if mode == "copy_only":
                to = step
            elif mode == "one_step":
                step += 1; gs[rows.long()] = step; to = step
            else:
                step += 20; to = step
            t0 = time.perf_counter()
            _lib.check(L.clmgs_host_rows_prepare(P(p), P(g), P(m), P(v), P(last), P(gs), P(rows), V, 48, P(col_lr), 0.9, 0.999,
                                                 1e-15, to, 0, 1, 0.25, 256, P(stage), 0))
            best = min(best, time.perf_counter() - t0)
        print(json.dumps({"mode": mode, "threads": nt, "ms": round(best * 1e3, 1), "Mrows_per_s": round(V / best / 1e6, 1)}))
