"""Non-clmgs kernels inside the last window of a rocprofv3 rocpd database:
python profiles/torch_ops.py results.db [window_ms]"""
import re
import sqlite3
import sys

db = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
t_end = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= t_end - win_ms * 1e6]
tot = {}
for n, s, e in rows:
    if "clmgs::" in n:
        continue
    k = re.sub(r"\s+", " ", n)
    m = re.search(r"(at::native::[A-Za-z_0-9:<>]*?(Functor|kernel|Kernel|Op)[A-Za-z_0-9]*)", k)
    k = (m.group(1) if m else k)[:110]
    a = tot.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e6
print(f"non-clmgs kernels in the last {win_ms} ms: {sum(v[1] for v in tot.values()):.2f} ms, {sum(v[0] for v in tot.values())} launches")
for k, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{ms:7.3f} ms n={n:3d}  {k}")
