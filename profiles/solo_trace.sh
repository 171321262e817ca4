#!/bin/bash
# Per-kernel durations with NOTHING co-running: kernel trace of a few single-stream batches (overlap_cameras=false).
# usage (GPU box, repo root): bash profiles/solo_trace.sh <tag> [extra bench args]   (env is passed through)
R=$(pwd); TAG=$1; shift
O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o st -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline \
  --no-host-leg --no-trainer-leg --no-heavy-leg --no-kernel-timing --gt resident --prime-seconds 0 --opt overlap_cameras=false "$@" > $O/solo_trace_$TAG.log 2>&1
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 110 > $O/solo_kernel_stats_$TAG.csv
cd $R
