"""Summarise rocprofv3 --pmc csv output per kernel: python profiles/pmc_summary.py <dir>/..._counter_collection.csv"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for x in csv.DictReader(open(sys.argv[1])):
    k = x["Kernel_Name"].split("(")[0][-48:]
    agg[k][x["Counter_Name"]] += float(x["Counter_Value"])
    cnt[k][x["Counter_Name"]] += 1
for k in sorted(agg):
    if "clmgs" in k or "rocprim" in k:
        print(k, {c: round(v / cnt[k][c], 1) for c, v in agg[k].items()}, "launches", max(cnt[k].values()))
