#!/bin/bash
# Round-5 recipe behind profiles/r05_* (GPU box, repo root; ~17 GPU-minutes).  Parity / DP reports come from the test
# suite (gpurun_out/parity_fullsize.json, dp_bytes.json); probes: raster_microbench.py, densify_probe.py.
set -x
R=$(pwd)
O=$R/gpurun_out/r05; mkdir -p $O
N="--no-cpu-baseline --no-host-leg --no-trainer-leg --no-heavy-leg"
# 1. the driver's command (all legs: value, value_gt_streamed, value_heavy, host_resident, trainer, cpu_baseline)
timeout 700 python bench.py --steps 20 --warmup 5 > $O/bench_28m_final.log 2> $O/bench_28m_final.err
# 2. kernel trace of the timed steps only (in situ), one pipelined batch as a timeline
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/prof_h /tmp/pmcF /tmp/pmcW /tmp/pmcS
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --steps 8 --warmup 3 $N --no-kernel-timing --gt resident > $O/prof_s.log 2>&1
DB=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 165 > $O/kernel_stats.csv
python $R/profiles/timeline.py $DB step3 > $O/timeline_step.txt 2>&1
python $R/profiles/timeline_streams.py $DB step3 > $O/timeline_streams.txt 2>&1
cd $R
# 3. the same kernels with nothing co-running
bash $R/profiles/solo_trace.sh r05solo > /dev/null 2>&1; cp $R/gpurun_out/r4/solo_kernel_stats_r05solo.csv $O/kernel_stats_single_stream.csv
# 4. PMC passes (separate, --kernel-trace only)
cd /tmp
B="python $R/bench.py --steps 2 --warmup 2 $N --no-kernel-timing --gt resident --prime-seconds 0"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -o f -- $B > $O/pmcF.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcF -name "*counter_collection.csv" | head -1) > $O/pmc_fetch_size.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -o w -- $B > $O/pmcW.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcW -name "*counter_collection.csv" | head -1) > $O/pmc_write_size.txt 2>&1
echo "== SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > $O/pmc_sq_counters.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmcS -o s -- $B --opt overlap_cameras=false > $O/pmcS.log 2>&1
python $R/profiles/pmc_summary.py $(find /tmp/pmcS -name "*counter_collection.csv" | head -1) >> $O/pmc_sq_counters.txt 2>&1
cd $R
# 5. A/B legs of the round's changes (same box, 20 timed steps each)
timeout 300 python bench.py --steps 20 --warmup 5 $N --opt binning=sort > $O/bench_28m_binning_sort.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 $N --opt deferred_small_adam=false > $O/bench_28m_small_adam_eager.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 $N --opt split_catch_up=false > $O/bench_28m_catch_up_single_pass.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 $N --opt overlap_cameras=false > $O/bench_28m_no_overlap.log 2>&1
# 6. heavy scene: bench line + kernel trace
timeout 400 python bench.py --scene heavy --steps 10 --warmup 3 $N > $O/bench_28m_heavy.log 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o st -- python $R/bench.py --scene heavy --steps 6 --warmup 2 $N --no-kernel-timing --gt resident > $O/prof_h.log 2>&1
DB=$(find /tmp/prof_h -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 200 > $O/kernel_stats_heavy.csv
cd $R
# 7. the other BASELINE.json configurations
timeout 400 python bench.py --config rubble10m --steps 10 --warmup 3 $N > $O/bench_rubble10m_clm.log 2>&1
timeout 300 python bench.py --config bicycle6m --strategy no_offload --steps 10 --warmup 3 $N > $O/bench_bicycle6m_no_offload.log 2>&1
timeout 300 python bench.py --config bicycle6m --steps 10 --warmup 3 $N > $O/bench_bicycle6m_clm.log 2>&1
timeout 400 python bench.py --config bigcity102m --steps 6 --warmup 2 $N > $O/bench_bigcity102m_1gpu.log 2>&1
# 8. probes
timeout 200 python profiles/raster_microbench.py > $O/raster_microbench.txt 2>&1
timeout 300 python profiles/densify_probe.py > $O/densify_probe.json 2>/dev/null
# 9. camera-DP at full size with both ranks on the one GPU (gloo): bytes per collective, phases, pre-flight, replicas_equal
CLMGS_DIST_BACKEND=gloo CLMGS_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --allreduce-steps 2 > $O/bench_28m_dp2_one_gpu_gloo.log 2> $O/bench_28m_dp2_one_gpu_gloo.err
ls -la $O
