#!/bin/bash
# Round-3 recipe behind profiles/r03_* (run on the GPU box from the repo root; ~10 GPU-minutes).
# Parity / DP-bytes / index-defect reports come from the test suite and profiles/repro_index_defect.py.
set -x
R=$(pwd)
O=$R/gpurun_out/r03
mkdir -p $O
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_28m_final.log 2> $O/bench_28m_final.err
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt overlap_cameras=false --no-host-leg > $O/bench_28m_no_overlap.log 2>&1
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --opt device_side_counts=false > $O/bench_28m_exact_sizes.log 2>&1
timeout 300 python bench.py --config rubble10m --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_rubble10m_clm.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-kernel-timing --gt host > $O/bench_28m_gt_host.log 2>&1
timeout 300 python bench.py --residency host --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-timing --prime-seconds 0 --no-host-hint > $O/bench_28m_host_no_hint.log 2>&1
timeout 200 python bench.py --config bicycle6m --strategy no_offload --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_bicycle6m_no_offload.log 2>&1
timeout 200 python bench.py --config bicycle6m --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg > $O/bench_bicycle6m_clm.log 2>&1
timeout 300 python bench.py --config bigcity102m --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_bigcity102m_1gpu.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/pmcF /tmp/pmcW /tmp/pmcS
# kernel trace of the timed steps only (--no-kernel-timing: no instrumented pass, no single-stream batch after them)
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-leg --no-kernel-timing --gt resident > $O/prof_s.log 2>&1
DB=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 170 > $O/kernel_stats.csv
python $R/profiles/timeline.py $DB step3 > $O/timeline_step.txt 2>&1
python $R/profiles/timeline_streams.py $DB step3 > $O/timeline_streams.txt 2>&1
if [ -n "$SKIP_PMC" ]; then cd $R; ls -la $O; exit 0; fi
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -o f -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-leg --gt resident --prime-seconds 0 > $O/pmcF.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcF -name "*counter_collection.csv" | head -1) > $O/pmc_fetch_size.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -o w -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-leg --gt resident --prime-seconds 0 > $O/pmcW.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcW -name "*counter_collection.csv" | head -1) > $O/pmc_write_size.txt 2>&1
rm -f $O/pmc_sq_counters.txt
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmcS
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcS -o s -- python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-leg --prime-seconds 0 --gt resident --opt overlap_cameras=false > $O/pmcS.log 2>&1
  echo "== $SET" >> $O/pmc_sq_counters.txt
  python $R/profiles/pmc_summary.py $(find /tmp/pmcS -name "*counter_collection.csv" | head -1) >> $O/pmc_sq_counters.txt 2>&1
done
cd $R
ls -la $O
