cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/prof_a
rocprofv3 --kernel-trace -d /tmp/prof_a -o tr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --opt overlap_cameras=false > $R/gpurun_out/prof_a.log 2>&1
DB=$(find /tmp/prof_a -name "*.db" | head -1)
python - "$DB" > $R/gpurun_out/binning_kernels.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
t_end = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= t_end - 35.0e6]
agg = {}
for n, s, e in rows:
    k = n.split("(")[0][-70:]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t/1e6:8.3f} ms n={n:4d} avg {t/n/1e3:8.1f} us  {k}")
PY
