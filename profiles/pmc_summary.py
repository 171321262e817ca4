"""Summarise rocprofv3 --pmc csv output per kernel: python profiles/pmc_summary.py <dir>/..._counter_collection.csv
One line per kernel: `name {counter: mean per launch} launches n median {counter: median per launch}` -- the median is what
make_pmc_json.py uses for kernels whose launches are not alike (clmgs_adam_catch_up: the per-batch passes over the
touched rows and ONE whole-table flush at the end of the run)."""
import collections
import csv
import statistics
import sys

vals = collections.defaultdict(lambda: collections.defaultdict(list))
for x in csv.DictReader(open(sys.argv[1])):
    k = x["Kernel_Name"].split("(")[0][-48:]
    vals[k][x["Counter_Name"]].append(float(x["Counter_Value"]))
for k in sorted(vals):
    if "clmgs" in k or "rocprim" in k:
        print(k, {c: round(sum(v) / len(v), 1) for c, v in vals[k].items()}, "launches", max(len(v) for v in vals[k].values()),
              "median", {c: round(statistics.median(v), 1) for c, v in vals[k].items()})
