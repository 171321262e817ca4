"""Probe of the GPU box's host side (run on the GPU box): what the host<->HBM split of the clm_offload
strategy can count on.  Prints one JSON object.

  * host topology (cores, NUMA nodes, memory)
  * hipMemcpyAsync pinned H2D / D2H / both at once (SDMA engines), 1 GiB each
  * zero-copy row gather (GPU kernel reading 192 B rows of mapped pinned memory, random rows), row scatter
    (plain store) and row scatter-add (read-modify-write over the link)
  * host Adam over random 192 B rows of four pinned [N,48] tables with 8..256 threads
"""
import ctypes
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clm_gs_amd import _lib, clm_kernels  # noqa: E402
from clm_gs_amd.host import pinned_empty  # noqa: E402


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:
        return f"failed: {e}"


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    out = {"lscpu": sh("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA|^CPU\\(s\\)|L3'"),
           "numa": sh("numactl -H 2>/dev/null | head -20 || true"),
           "mem": sh("head -3 /proc/meminfo"),
           "affinity": len(os.sched_getaffinity(0)),
           "nodes": sh("ls -d /sys/devices/system/node/node* | wc -l")}
    dev = torch.device("cuda")
    N = int(os.environ.get("PROBE_ROWS", 12_000_000))
    V = 3_300_000
    tab = pinned_empty((N, 48))
    tab.zero_()
    GB = 1 << 30
    # ---- hipMemcpyAsync
    n1 = GB // 192
    d = torch.empty((n1, 48), device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d2 = torch.empty((n1, 48), device=dev)

    def h2d():
        with torch.cuda.stream(s1):
            d.copy_(tab[:n1], non_blocking=True)

    def d2h():
        with torch.cuda.stream(s2):
            tab[n1:2 * n1].copy_(d2, non_blocking=True)

    def both():
        h2d()
        d2h()
    out["memcpy_GBps"] = {"h2d": round(1.0 / timed(h2d) * 1.0737, 2), "d2h": round(1.0 / timed(d2h) * 1.0737, 2),
                          "both_each": round(1.0 / timed(both) * 1.0737, 2)}
    # chunked: 1 MB pieces (page-granular mirror)
    chunk = (1 << 20) // 192

    def h2d_chunks():
        with torch.cuda.stream(s1):
            for i in range(0, n1 - chunk, chunk):
                d[i:i + chunk].copy_(tab[i:i + chunk], non_blocking=True)
    t = timed(h2d_chunks, 2)
    out["memcpy_GBps"]["h2d_1MB_chunks"] = round(1.0737 / t, 2)
    out["memcpy_GBps"]["calls_per_s"] = round((n1 // chunk) / t, 0)
    # ---- zero-copy kernels
    g = torch.Generator(device=dev).manual_seed(0)
    rows = torch.randperm(N, generator=g, device=dev)[:V].sort().values
    buf = torch.empty((V, 48), device=dev)
    byts = V * 192 / 1e9
    zc = {}
    for grid in (32, 256, 2048, 0):
        zc[f"gather_grid{grid}"] = round(byts / timed(lambda: clm_kernels.send_shs2gpu_stream(buf, tab, rows, grid, 256)), 2)
    zc["scatter_store"] = round(byts / timed(lambda: clm_kernels.send_shs2cpu_grad_buffer_stream(buf, tab, rows, False, 0, 256)), 2)
    zc["scatter_add_rmw"] = round(byts / timed(lambda: clm_kernels.send_shs2cpu_grad_buffer_stream(buf, tab, rows, True, 0, 256)), 2)

    def gather_and_scatter():
        with torch.cuda.stream(s1):
            clm_kernels.send_shs2gpu_stream(buf, tab, rows, 0, 256)
        with torch.cuda.stream(s2):
            clm_kernels.send_shs2cpu_grad_buffer_stream(d2[:V] if V <= n1 else buf, tab, rows, False, 0, 256)
    zc["gather+store_concurrent_each"] = round(byts / timed(gather_and_scatter), 2)
    out["zero_copy_GBps"] = zc
    # ---- host Adam
    del d, d2, buf
    g_t, m_t, v_t = pinned_empty((N, 48)), pinned_empty((N, 48)), pinned_empty((N, 48))
    for t_ in (g_t, m_t, v_t):
        t_.zero_()
    g_t[:] = 1e-3
    rows_h = rows.to(torch.int32).cpu().contiguous()
    col_lr = torch.full((48,), 1e-3)
    L = _lib.lib()
    P = lambda t_: ctypes.c_void_p(t_.data_ptr())
    ad = {}
    for nt in (8, 32, 64, 128, 256):
        if nt > out["affinity"]:
            continue
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            _lib.check(L.clmgs_host_adam_rows(P(tab), P(g_t), P(m_t), P(v_t), P(rows_h), V, 48, P(col_lr), 0.9, 0.999,
                                              1e-15, 3, 1, 0.25, 1, None, nt))
            best = min(best, time.perf_counter() - t0)
        ad[f"threads{nt}"] = {"ms": round(best * 1e3, 1), "Mrows_per_s": round(V / best / 1e6, 1),
                              "GBps_dram": round(V * 192 * 8 / best / 1e9, 1)}
    out["host_adam_3.3Mrows"] = ad
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
