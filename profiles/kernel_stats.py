"""Per-kernel stats of the TIMED region of a rocprofv3 rocpd database (the last `window_ms` ms,
i.e. without scene generation and GT rendering): python profiles/kernel_stats.py results.db window_ms [anchor]
anchor (round 6): a kernel-name substring; the window then ENDS at the start of the last launch of that kernel instead of
at the end of the trace -- bench.py's timed steps are followed by the whole-table flush of the deferred row optimizer
("adam_catch_up48_kernel<int>") and by evidence passes with long host gaps, which a window counted from the end of the trace
would spend its milliseconds on."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6
rows = list(c.execute("select name, start, end from kernels order by start"))
t_end = max(r[2] for r in rows)
if len(sys.argv) > 3:
    hits = [r[1] for r in rows if sys.argv[3] in r[0]]
    if hits:
        t_end = max(hits)
rows = [r for r in rows if t_end - win <= r[1] < t_end] if len(sys.argv) > 3 else [r for r in rows if r[1] >= t_end - win]
agg = {}
for n, s, e in rows:
    a = agg.setdefault(n.split("(")[0][-90:], [0, 0, 10 ** 18, 0])
    a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
tot = sum(a[1] for a in agg.values())
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for n, (k, t, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'"{n}",{k},{t},{t / k:.1f},{lo},{hi},{100 * t / tot:.3f}')
