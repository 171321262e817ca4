"""Per-queue timeline of the last training step of a rocprofv3 rocpd database:
python profiles/timeline_streams.py results.db [window_ms | stepK]
Every kernel of the window with its HSA queue, start offset and duration, then for each queue the
busy time and, for the queue that runs the tile kernels, its idle gaps and what the other queues ran
meanwhile (how well the kernel-type streams overlap)."""
import sqlite3
import sys

db = sys.argv[1]
arg = sys.argv[2] if len(sys.argv) > 2 else "31"
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, queue_id, grid_x, workgroup_x, lds_size from kernels order by start"))
if arg.startswith("step"):  # stepK: the K-th batch from the end, between two visibility_bits launches
    marks = [r[1] for r in rows if "visibility_bits" in r[0]]
    k = int(arg[4:] or 2)
    t0, t_end = marks[-k - 1], marks[-k]
    rows = [r for r in rows if t0 <= r[1] < t_end]
else:
    t_end = max(r[2] for r in rows)
    t0 = t_end - float(arg) * 1e6
    rows = [r for r in rows if r[1] >= t0]
KEYS = ("rasterize_bwd", "rasterize_fwd", "partials_sum", "preprocess_bwd", "preprocess_fwd", "loss_bwd",
        "loss_fwd", "adam_rows", "adam_small", "catch_up", "radix_scan_rows", "radix_scatter",
        "radix_onesweep", "radix_hist_all", "radix_hist", "radix_scan_digits", "isect2_keys", "isect2_count",
        "isect2_emit", "isect2_offsets", "scan_i64_blocks", "scan_i64_totals", "scan_i64_add",
        "visibility_bits", "visibility_emit", "copyBuffer", "fillBuffer", "index_put", "direct_copy")


def short(n):
    n = n.split("(")[0]
    for k in KEYS:
        if k in n:
            return k
    return n[-28:]


queues = sorted(set(r[3] for r in rows))
print("queues:", queues)
for n, s, e, q, gx, wx, lds in rows:
    print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e3:8.1f}us  q{queues.index(q)}  {short(n):18s} grid {gx // max(wx, 1)}x{wx} lds {lds}")
print()
raster_q = None
for q in queues:
    mine = [r for r in rows if r[3] == q]
    busy = sum(r[2] - r[1] for r in mine)
    n_r = sum(1 for r in mine if "rasterize" in r[0])
    print(f"q{queues.index(q)}: {len(mine)} kernels, busy {busy / 1e6:.2f} ms, tile kernels {n_r}")
    if n_r and (raster_q is None or n_r > raster_q[1]):
        raster_q = (q, n_r)
if raster_q:
    q = raster_q[0]
    mine = [r for r in rows if r[3] == q]
    print(f"\nidle gaps of q{queues.index(q)} (tile kernels) > 50 us and the other queues' kernels inside them:")
    for a, b in zip(mine[:-1], mine[1:]):
        gap = b[1] - a[2]
        if gap > 50e3:
            inside = [r for r in rows if r[3] != q and r[2] > a[2] and r[1] < b[1]]
            desc = ", ".join(f"{short(r[0])}:{(min(r[2], b[1]) - max(r[1], a[2])) / 1e3:.0f}" for r in inside)
            print(f"  {gap / 1e3:7.0f} us after {short(a[0])} at {(a[2] - t0) / 1e6:.2f} ms: {desc}")
