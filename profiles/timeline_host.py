"""Host-resident mode: how much of the host<->HBM traffic runs UNDER the rendering kernels.
python profiles/timeline_host.py results.db [window_ms | batchesK]
(`batches2` = the last two complete batches of the trace, from the third-last to the last
visibility_bits launch: the bench's final flush of the deferred row steps, host work only, stays out)
Reads a rocprofv3 rocpd database taken with --kernel-trace --memory-copy-trace.  Copies = the SDMA
transfers (hipMemcpyAsync: staging rows host->device, row lists device->host, GT images) PLUS the
zero-copy gradient scatter kernel (rows_move_f4_kernel writing pinned host memory).  "Render kernels" =
everything else on the device."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
arg = sys.argv[2] if len(sys.argv) > 2 else None
win = float(arg) * 1e6 if arg and not arg.startswith("batches") else None
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kern = list(c.execute("select name, start, end from kernels order by start"))
mc_view = next((n for n in names if n.lower() in ("memory_copies", "memory_copy")), None) or \
    next((n for n in names if "memory_cop" in n.lower() and "rocpd_" not in n.lower()), None) or \
    next((n for n in names if "memory_cop" in n.lower()), None)
copies = []
if mc_view:
    cols = [r[1] for r in c.execute(f"pragma table_info('{mc_view}')")]
    sc, ec = ("start" if "start" in cols else "start_timestamp"), ("end" if "end" in cols else "end_timestamp")
    size_c = next((x for x in cols if x in ("size", "bytes", "size_bytes")), None)
    name_c = next((x for x in cols if x in ("name", "kind", "direction")), None)
    q = f"select {sc}, {ec}, {size_c or 0}, {name_c or 0} from {mc_view} order by {sc}"
    copies = [(s, e, b, str(n)) for s, e, b, n in c.execute(q)]
t_end = max([k[2] for k in kern] + [x[1] for x in copies])
t0 = t_end - win if win else min(k[1] for k in kern)
if arg and arg.startswith("batches"):
    marks = [k[1] for k in kern if "visibility_bits" in k[0]]
    nb = int(arg[7:] or 2)
    t0, t_end = marks[-nb - 1], marks[-1]
kern = [k for k in kern if t0 <= k[1] < t_end]
copies = [x for x in copies if t0 <= x[0] < t_end]
scatter = [(s, e) for n, s, e in kern if "rows_move_f4_kernel" in n]
render = sorted((s, e) for n, s, e in kern if "rows_move_f4_kernel" not in n)


def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def overlap(a, b):  # total length of a covered by b (both unions)
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


ru = union(render)
span = t_end - t0
print(f"window {span / 1e6:.1f} ms; render kernels busy {sum(e - s for s, e in ru) / 1e6:.1f} ms")
print(f"copy view: {mc_view}; {len(copies)} SDMA copies, {len(scatter)} zero-copy scatter launches")
by = {}
for s, e, b, n in copies:
    a = by.setdefault(n, [0, 0, 0.0])
    a[0] += 1; a[1] += int(b or 0); a[2] += (e - s)
for n, (k, b, t) in by.items():
    print(f"  {n}: n={k} bytes={b / 1e9:.3f} GB busy={t / 1e6:.1f} ms -> {b / max(t, 1):.1f} GB/s while active")
for label, iv in (("SDMA copies", [(s, e) for s, e, _, _ in copies]), ("zero-copy gradient scatter", scatter),
                  ("all link traffic", [(s, e) for s, e, _, _ in copies] + scatter)):
    u = union(iv)
    tot = sum(e - s for s, e in u)
    if tot:
        print(f"{label}: active {tot / 1e6:.1f} ms ({100 * tot / span:.0f}% of the window), "
              f"{100 * overlap(u, ru) / tot:.0f}% of it under render kernels")
