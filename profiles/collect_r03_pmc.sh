#!/bin/bash
# The --pmc passes of round 3 alone (separate passes, --kernel-trace only; no priming renders so that every launch of
# a kernel in the summary is a training launch).  ~3 GPU-minutes.
set -x
R=$(pwd)
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcF /tmp/pmcW /tmp/pmcS
B="python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-leg --gt resident --prime-seconds 0"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -o f -- $B > $O/pmcF.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcF -name "*counter_collection.csv" | head -1) > $O/pmc_fetch_size.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -o w -- $B > $O/pmcW.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcW -name "*counter_collection.csv" | head -1) > $O/pmc_write_size.txt 2>&1
echo "== SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > $O/pmc_sq_counters.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmcS -o s -- $B --opt overlap_cameras=false > $O/pmcS.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcS -name "*counter_collection.csv" | head -1) >> $O/pmc_sq_counters.txt 2>&1
cd $R
