"""Timeline analysis of a rocprofv3 rocpd database: python profiles/timeline.py results.db [n_last_ms | stepK]
Prints GPU busy fraction, per-kernel totals and the idle gaps inside the window.  `stepK` (e.g. step3):
the K-th batch counted from the END of the trace, delimited by two consecutive visibility_bits launches
(one per batch) -- use it with `bench.py --no-kernel-timing`, whose trace ends with the timed steps (the
default bench run ends with an instrumented pass and a single-stream batch)."""
import sqlite3
import sys

db = sys.argv[1]
arg = sys.argv[2] if len(sys.argv) > 2 else "150"
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))
if arg.startswith("step"):
    marks = [r[1] for r in rows if "visibility_bits" in r[0]]
    k = int(arg[4:] or 2)
    t0, t_end = marks[-k - 1], marks[-k]
    rows = [r for r in rows if t0 <= r[1] < t_end]
    print(f"{len(rows)} kernels in batch -{k} of the trace ({(t_end - t0) / 1e6:.2f} ms between two visibility_bits launches)")
else:
    win_ms = float(arg)
    t_end = max(r[2] for r in rows)
    t0 = t_end - win_ms * 1e6
    rows = [r for r in rows if r[1] >= t0]
    print(f"{len(rows)} kernels in the last {win_ms} ms")
iv = sorted((r[1], r[2]) for r in rows)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = (t_end - t0) if arg.startswith("step") else iv[-1][1] - iv[0][0]
ksum = sum(e - s for s, e in iv)
print(f"span {span/1e6:.2f} ms, GPU busy (union) {busy/1e6:.2f} ms = {100*busy/span:.1f}%, "
      f"sum of kernel durations {ksum/1e6:.2f} ms = {ksum/max(busy,1):.2f}x the busy time (concurrency)")
tot = {}
for n, s, e, st in rows:
    k = n.split("(")[0][-60:]
    a = tot.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e6
for k, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{ms:8.2f} ms n={n:4d}  {k}")
print("largest idle gaps:")
for g, at in sorted(gaps, reverse=True)[:14]:
    prev = [r for r in rows if r[2] <= at + 1][-1][0].split("(")[0][-50:]
    nxt = [r for r in rows if r[1] >= at + g - 1][0][0].split("(")[0][-50:]
    print(f"  {g/1e6:6.3f} ms at t-{(t_end-at)/1e6:7.2f} ms  after {prev}  before {nxt}")
