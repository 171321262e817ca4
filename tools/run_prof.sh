cd /tmp && export TMPDIR=/tmp
R=/root/repo
cd $R && python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-300
cd /tmp
rm -rf /tmp/prof_s /tmp/pmcF /tmp/pmcW
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_s.log 2>&1
DB=$(find /tmp/prof_s -name "*.db" | head -1)
python - "$DB" > $R/gpurun_out/kernel_stats.csv <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
t_end = max(r[2] for r in rows)
# only the timed region: the last 5 steps (~170 ms), setup kernels (GT rendering, scene generation) excluded
win = 5 * 33.5e6
rows = [r for r in rows if r[1] >= t_end - win]
agg = {}
for n, s, e in rows:
    a = agg.setdefault(n.split("(")[0][-90:], [0, 0, 10**18, 0])
    a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
tot = sum(a[1] for a in agg.values())
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for n, (k, t, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'"{n}",{k},{t},{t/k:.1f},{lo},{hi},{100*t/tot:.3f}')
PY
python $R/profiles/timeline.py $DB 33 > $R/gpurun_out/timeline_default.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmcF.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcF -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_fetch.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmcW.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcW -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_write.txt 2>&1
cd $R && timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt overlap_cameras=false > gpurun_out/ab0.log 2>&1; python profiles/show_bench.py gpurun_out/ab0.log | tail -16
