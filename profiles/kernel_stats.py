"""Per-kernel stats of the TIMED region of a rocprofv3 rocpd database (the last `window_ms` ms,
i.e. without scene generation and GT rendering): python profiles/kernel_stats.py results.db window_ms"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6
rows = list(c.execute("select name, start, end from kernels order by start"))
t_end = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= t_end - win]
agg = {}
for n, s, e in rows:
    a = agg.setdefault(n.split("(")[0][-90:], [0, 0, 10 ** 18, 0])
    a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
tot = sum(a[1] for a in agg.values())
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for n, (k, t, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'"{n}",{k},{t},{t / k:.1f},{lo},{hi},{100 * t / tot:.3f}')
