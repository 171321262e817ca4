#!/bin/bash
# Round-4 recipe behind profiles/r04_* (GPU box, repo root; ~12 GPU-minutes).  Parity / DP reports come from the test
# suite (gpurun_out/parity_fullsize.json, dp_bytes.json); probes: loss_microbench.py, densify_probe.py, dp_world1_probe.py.
set -x
R=$(pwd)
O=$R/gpurun_out/r04
mkdir -p $O
# .git does not travel to the GPU box: the label passed to make_pmc_json.py names the tree
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_28m_final.log 2> $O/bench_28m_final.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt overlap_cameras=false --no-host-leg --no-trainer-leg > $O/bench_28m_no_overlap.log 2>&1
CLMGS_BINNING=legacy timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-trainer-leg > $O/bench_28m_binning_legacy.log 2>&1
CLMGS_BINNING=lookback timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-trainer-leg > $O/bench_28m_binning_lookback.log 2>&1
timeout 400 python bench.py --scene heavy --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-trainer-leg > $O/bench_28m_heavy.log 2>&1
timeout 400 python bench.py --config rubble10m --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_rubble10m_clm.log 2>&1
timeout 300 python bench.py --config bicycle6m --strategy no_offload --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_bicycle6m_no_offload.log 2>&1
timeout 300 python bench.py --config bicycle6m --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-trainer-leg > $O/bench_bicycle6m_clm.log 2>&1
timeout 400 python bench.py --config bigcity102m --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_bigcity102m_1gpu.log 2>&1
timeout 200 python profiles/loss_microbench.py > $O/loss_microbench.txt 2>&1
timeout 300 python profiles/densify_probe.py > $O/densify_probe.json 2>/dev/null
timeout 200 python profiles/dp_world1_probe.py 2>&1 | grep DPPROBE > $O/dp_world1_probe.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/prof_h /tmp/pmcF /tmp/pmcW /tmp/pmcS
# kernel trace of the timed steps only (--no-kernel-timing: no instrumented pass, no single-stream batch after them)
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-leg --no-trainer-leg --no-kernel-timing --gt resident > $O/prof_s.log 2>&1
DB=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 190 > $O/kernel_stats.csv
python $R/profiles/timeline.py $DB step3 > $O/timeline_step.txt 2>&1
python $R/profiles/timeline_streams.py $DB step3 > $O/timeline_streams.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o st -- python $R/bench.py --scene heavy --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-trainer-leg --no-kernel-timing --gt resident > $O/prof_h.log 2>&1
DB=$(find /tmp/prof_h -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 240 > $O/kernel_stats_heavy.csv
bash $R/profiles/solo_trace.sh r04solo > /dev/null 2>&1; cp $R/gpurun_out/r4/solo_kernel_stats_r04solo.csv $O/kernel_stats_single_stream.csv
cd /tmp
B="python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-leg --no-trainer-leg --gt resident --prime-seconds 0"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -o f -- $B > $O/pmcF.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcF -name "*counter_collection.csv" | head -1) > $O/pmc_fetch_size.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -o w -- $B > $O/pmcW.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcW -name "*counter_collection.csv" | head -1) > $O/pmc_write_size.txt 2>&1
echo "== SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > $O/pmc_sq_counters.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmcS -o s -- $B --opt overlap_cameras=false > $O/pmcS.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcS -name "*counter_collection.csv" | head -1) >> $O/pmc_sq_counters.txt 2>&1
cd $R
ls -la $O
