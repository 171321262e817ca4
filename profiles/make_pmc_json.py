"""Build profiles/pmc_traffic.json (read by bench.py) from the pmc_summary.py outputs:
python profiles/make_pmc_json.py <fetch_summary.txt> <write_summary.txt> [<sq_summary.txt> [<source label> [<bench log>]]]
The optional SQ summary (a --pmc pass with SQ_INSTS_VALU) adds valu_insts = wave-level VALU instructions
per launch, the numerator of the compute-side roofline (peak: profiles/valu_calib.hip).  The optional bench log (the
JSON line of the same configuration, with its per-kernel table) adds algo_bytes and traffic_over_algo per entry, so
wasted traffic is one column.  Entries without a single counted byte (a kernel the configuration never launches) are left
out; clmgs_adam_catch_up takes the MEDIAN launch (the per-batch pass), not the mean that the one whole-table flush skews."""
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


MEDIAN_KERNELS = ("adam_catch_up",)  # launches of very different sizes: the typical (per-batch) launch is the median


def _lines(path):
    """-> (kernel name, counters per launch).  MEDIAN_KERNELS: the median launch of the instantiation with the MOST launches
    (adam_catch_up48_kernel<long> = the per-camera passes; <int> = the one whole-table flush at the end of the run)."""
    rows = []
    for line in open(path):
        m = re.match(r"(.*?) (\{.*?\}) launches (\d+)(?: median (\{.*\}))?", line.strip())
        if m:
            mean = ast.literal_eval(m.group(2))
            med = ast.literal_eval(m.group(4)) if m.group(4) else mean
            rows.append((m.group(1), mean, med, int(m.group(3))))
    for x in MEDIAN_KERNELS:
        fam = [r for r in rows if x in r[0]]
        if len(fam) > 1:
            keep = max(fam, key=lambda r: r[3])
            rows = [r for r in rows if r not in fam or r is keep]
    for name, mean, med, _n in rows:
        yield name, (med if any(x in name for x in MEDIAN_KERNELS) else mean)


def read(path, counter):
    return {k: d[counter] * 1024.0 for k, d in _lines(path) if counter in d}  # KB -> bytes, per launch


def read_raw(path, counter):
    return {k: d[counter] for k, d in _lines(path) if counter in d}


f, w = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
sq = read_raw(sys.argv[3], "SQ_INSTS_VALU") if len(sys.argv) > 3 else {}
label = sys.argv[4] if len(sys.argv) > 4 else "this round's build"
algo = {}
if len(sys.argv) > 5:  # bench log: per-kernel algorithmic bytes of the same configuration
    for line in open(sys.argv[5]):
        if line.startswith("{"):
            j = json.loads(line)
            for name, k in (j.get("kernels") or {}).items():
                if "algo_GBps" in k:
                    algo[name] = k["algo_GBps"] * 1e9 * k["avg_ms"] * 1e-3
            rf = j.get("roofline") or {}
            if rf.get("kernel") and rf.get("algo_bytes_per_launch"):
                algo[rf["kernel"]] = rf["algo_bytes_per_launch"]
res = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), KB per launch "
                "averaged over the launches of `bench.py --steps 1 --warmup 1`; bytes = KB*1024, summed over "
                "the device kernels of one C-ABI call. MI355X guide: FETCH_SIZE reports 1/2 of wide coalesced "
                "reads on gfx950 (x2 applied in fetch_bytes_x2); other widths and WRITE_SIZE uncalibrated. "
                "traffic = fetch_bytes_x2 + write_bytes.",
       "rubble28m": {}}
for entry, kernels in GROUPS.items():
    fr = sum(v for k, v in f.items() if any(x in k for x in kernels) and "unsigned long" not in k)
    wr = sum(v for k, v in w.items() if any(x in k for x in kernels) and "unsigned long" not in k)
    if fr + wr == 0:
        continue  # never launched in this configuration (e.g. clmgs_adam_rows under the deferred row optimizer)
    res["rubble28m"][entry] = {"fetch_bytes_raw": fr, "fetch_bytes_x2": 2 * fr, "write_bytes": wr,
                               "traffic": 2 * fr + wr}
    if entry in algo:
        res["rubble28m"][entry]["algo_bytes"] = round(algo[entry], 1)
        res["rubble28m"][entry]["traffic_over_algo"] = round((2 * fr + wr) / algo[entry], 3)
    if any(x in k for k in list(f) + list(w) for x in kernels if x in MEDIAN_KERNELS):
        res["rubble28m"][entry]["launch"] = "median launch of the per-camera passes (split_catch_up: four per batch; the whole-table flush excluded)"
    vi = sum(v for k, v in sq.items() if any(x in k for x in kernels) and "unsigned long" not in k)
    if vi:
        res["rubble28m"][entry]["valu_insts"] = vi
res["_source"] = "rocprofv3 --pmc passes of " + label + " (profiles/collect_r06.sh)"
json.dump(res, open(__file__.replace("make_pmc_json.py", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res["rubble28m"], indent=1))
