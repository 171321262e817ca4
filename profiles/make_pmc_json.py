"""Build profiles/pmc_traffic.json (read by bench.py) from the pmc_summary.py outputs:
python profiles/make_pmc_json.py <fetch_summary.txt> <write_summary.txt> [<sq_summary.txt> [<source label>]]
The optional SQ summary (a --pmc pass with SQ_INSTS_VALU) adds valu_insts = wave-level VALU instructions
per launch, the numerator of the compute-side roofline (peak: profiles/valu_calib.hip)."""
import ast
import json
import re
import sys

# C-ABI entry -> device kernels that make up one launch of it
GROUPS = {
    "clmgs_rasterize_bwd": ["rasterize_bwd_kernel"],  # engine path: the row sum lives in preprocess_bwd
    "clmgs_rasterize_fwd": ["rasterize_fwd_kernel"],
    "clmgs_l1_ssim_loss_fwd": ["loss_fwd_kernel"],
    "clmgs_l1_ssim_loss_bwd": ["loss_bwd_kernel"],
    "clmgs_preprocess_fwd": ["preprocess_fwd_kernel"],
    "clmgs_preprocess_bwd": ["preprocess_bwd_kernel"],
    "clmgs_adam_rows": ["adam_rows_kernel"],
    "clmgs_adam_catch_up": ["adam_catch_up"],
    "clmgs_adam_small_deferred": ["adam_small_deferred_kernel"],
    "clmgs_isect3_front": ["isect3_rows_kernel", "isect3_rows_finish_kernel"],
    "clmgs_isect3_bin": ["isect3_tile_scan_kernel", "isect3_scatter_kernel", "isect3_sort_"],
}


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"(.*?) (\{.*\}) launches (\d+)", line.strip())
        if m:
            out[m.group(1)] = ast.literal_eval(m.group(2))[counter] * 1024.0  # KB -> bytes, per launch
    return out


def read_raw(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"(.*?) (\{.*\}) launches (\d+)", line.strip())
        if m and counter in ast.literal_eval(m.group(2)):
            out[m.group(1)] = ast.literal_eval(m.group(2))[counter]
    return out


f, w = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
sq = read_raw(sys.argv[3], "SQ_INSTS_VALU") if len(sys.argv) > 3 else {}
label = sys.argv[4] if len(sys.argv) > 4 else "this round's build"
res = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), KB per launch "
                "averaged over the launches of `bench.py --steps 1 --warmup 1`; bytes = KB*1024, summed over "
                "the device kernels of one C-ABI call. MI355X guide: FETCH_SIZE reports 1/2 of wide coalesced "
                "reads on gfx950 (x2 applied in fetch_bytes_x2); other widths and WRITE_SIZE uncalibrated. "
                "traffic = fetch_bytes_x2 + write_bytes.",
       "rubble28m": {}}
for entry, kernels in GROUPS.items():
    fr = sum(v for k, v in f.items() if any(x in k for x in kernels) and "unsigned long" not in k)
    wr = sum(v for k, v in w.items() if any(x in k for x in kernels) and "unsigned long" not in k)
    res["rubble28m"][entry] = {"fetch_bytes_raw": fr, "fetch_bytes_x2": 2 * fr, "write_bytes": wr,
                               "traffic": 2 * fr + wr}
    vi = sum(v for k, v in sq.items() if any(x in k for x in kernels) and "unsigned long" not in k)
    if vi:
        res["rubble28m"][entry]["valu_insts"] = vi
res["_source"] = "rocprofv3 --pmc passes of " + label + " (profiles/collect_r05.sh)"
json.dump(res, open(__file__.replace("make_pmc_json.py", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res["rubble28m"], indent=1))
