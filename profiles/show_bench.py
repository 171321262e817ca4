"""Pretty-print the JSON line of a bench.py log: python profiles/show_bench.py gpurun_out/b.log"""
import json
import sys

l = [x for x in open(sys.argv[1]) if x.startswith("{")]
d = json.loads(l[-1])
print(d["value"], "img/s", d["ms_per_step"], "ms/step; peak", round(d["peak_gpu_bytes"] / 1e9, 2), "GB", d["measured"])
tot = 0.0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"]):
    ms_step = v["share_of_step"] * d["ms_per_step"]
    tot += ms_step
    print(f"{k:28s} calls={v['calls']:3d} avg={v['avg_ms']:8.4f} ms  per-step={ms_step:6.2f} ms  {v.get('algo_GBps', '')}")
print("C-ABI kernels per step:", round(tot, 2), "ms of", d["ms_per_step"])
if "cpu_baseline" in d:
    print(d["cpu_baseline"])
