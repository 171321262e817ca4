cd /tmp && export TMPDIR=/tmp
R=/root/repo
cd $R && python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-400
cd /tmp
rm -rf /tmp/prof_s /tmp/pmcF /tmp/pmcW
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_s.log 2>&1
DB=$(find /tmp/prof_s -name "*.db" | head -1)
python - "$DB" > $R/gpurun_out/kernel_stats.csv <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for n, k, t, a, lo, hi in rows[:40]:
    n = n.split("(")[0][-90:]
    print(f'"{n}",{k},{t},{a:.1f},{lo},{hi},{100*t/tot:.3f}')
PY
python $R/profiles/timeline.py $DB 36 > $R/gpurun_out/timeline_default.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmcF.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcF -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_fetch.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmcW.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcW -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_write.txt 2>&1
