#!/bin/bash
# Host-resident mode (per-camera windows), peak GPU bytes and images/s against sh_hbm_budget_gb at 28 M rows (768 B per resident
# row: 21.5 GB = every row).  One bench process per budget on the same box; 12 timed host-resident batches each.
mkdir -p gpurun_out/r06
OUT=gpurun_out/r06/host_budget_curve.jsonl; : > $OUT
for B in 0 1.8 5.4 10.75 16.1 21.6; do
  timeout 600 python bench.py --steps 14 --warmup 2 --no-cpu-baseline --no-trainer-leg --no-heavy-leg --no-kernel-timing \
    --no-host-staging-pair --host-budget-leg-gb 0 --host-budget-gb $B --host-steps 12 > /tmp/hb.log 2>/tmp/hb.err
  python - "$B" >> $OUT <<'PY'
import json, sys
for l in open("/tmp/hb.log"):
    if l.startswith("{"):
        h = json.loads(l)["host_resident"]
        print(json.dumps({"sh_hbm_budget_gb": float(sys.argv[1]), **{k: h.get(k) for k in (
            "hbm_resident_rows", "value", "value_steady", "ms_per_step", "peak_gpu_bytes", "touched_rows_per_batch",
            "host_touched_rows_per_batch", "late_rows_per_batch", "host_pool_busy_fraction", "final_flush_ms", "host_ms_per_step",
            "loss_last", "error")}}))
PY
done
cat $OUT
