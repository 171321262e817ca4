"""CLMGS_BWD_DEBUG=3 python profiles/bwd_phases.py : phase breakdown of the backward tile kernel."""
import ctypes, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CLMGS_BWD_DEBUG"] = "3"
import torch
from clm_gs_amd import _lib
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-leg", "--no-trainer-leg", "--no-heavy-leg",
            "--gt", "resident", "--prime-seconds", "0", "--scene", os.environ.get("CLMGS_PHASES_SCENE", "slab")]
import bench
L = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
L.clmgs_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
bench.main()
L.clmgs_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
v = list(buf)
names = ["stage_cyc", "loop_cyc", "flush_cyc", "total_cyc", "entries", "entries_valid", "quadrant_passes", "rounds", "tiles", "list_len"]
d = dict(zip(names, v))
print(json.dumps(d))
tot = d["total_cyc"] or 1
print("share of wave time: stage %.1f%% loop %.1f%% flush %.1f%% other %.1f%%" % (
    100 * d["stage_cyc"] / tot, 100 * d["loop_cyc"] / tot, 100 * d["flush_cyc"] / tot,
    100 * (tot - d["stage_cyc"] - d["loop_cyc"] - d["flush_cyc"]) / tot))
print("entries/list %.3f  valid/entries %.3f  quad passes/entry %.2f  cyc/entry(loop) %.0f  rounds/tile %.2f" % (
    d["entries"] / max(d["list_len"], 1), d["entries_valid"] / max(d["entries"], 1),
    d["quadrant_passes"] / max(d["entries"], 1), d["loop_cyc"] / max(d["entries"], 1), d["rounds"] / max(d["tiles"], 1)))
