#!/bin/bash
# Recipe behind the files in profiles/ (run on the GPU box from the repo root):
#   r01_bench_28m_latest.log   default `python bench.py` (JSON line incl. roofline + cpu_baseline)
#   r01_kernel_stats.csv       rocprofv3 --kernel-trace --stats of the same command, timed region only
#   r01_timeline_step.txt      busy fraction / per-kernel totals / idle gaps of one step
#   r01_timeline_streams.txt   the same step per HSA queue (front / memory / tile streams)
#   r01_pmc_*                  separate --pmc passes (FETCH_SIZE, WRITE_SIZE), --kernel-trace only
#   r01_bench_28m_no_overlap.log   solo kernel times (single stream)
set -x
R=$(pwd)
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_default.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/pmcF /tmp/pmcW
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_s.log 2>&1
DB=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 160 > $R/gpurun_out/kernel_stats.csv
python $R/profiles/timeline.py $DB 30 > $R/gpurun_out/timeline_default.txt 2>&1
python $R/profiles/timeline_streams.py $DB 30 > $R/gpurun_out/timeline_streams.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmcF.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcF -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_fetch.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmcW.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcW -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_write.txt 2>&1
cd $R
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt overlap_cameras=false > gpurun_out/ab0.log 2>&1
python profiles/show_bench.py gpurun_out/ab0.log | tail -18
# SQ counters of the final build, single stream (solo kernels): VALU issue utilisation of the tile kernels
cd /tmp
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmcS
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcS -o s -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --prime-seconds 0 --opt overlap_cameras=false > $R/gpurun_out/pmcS.log 2>&1
  echo "== $SET" >> $R/gpurun_out/pmc_sq.txt
  python $R/profiles/pmc_summary.py $(find /tmp/pmcS -name "*counter_collection.csv" | head -1) >> $R/gpurun_out/pmc_sq.txt 2>&1
done
cd $R
