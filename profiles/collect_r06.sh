#!/bin/bash
# Round-6 recipe behind profiles/r06_* (GPU box, repo root; ~25 GPU-minutes).  Parity reports come from the test suite
# (gpurun_out/parity_fullsize.json, parity_report*.json); probes: valu_calib.hip, gather_probe.hip, raster_microbench.py.
set -x
R=$(pwd)
O=$R/gpurun_out/r06c; mkdir -p $O
N="--no-cpu-baseline --no-host-leg --no-trainer-leg --no-heavy-leg"
# 0. the FIRST process on the box: the gather ceiling + the deferred row pass alone (repeated at the very end: the first
#    process of a box has measured slower than later ones)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 profiles/gather_probe.hip -o /tmp/gather_probe 2>/dev/null
timeout 300 /tmp/gather_probe > $O/gather_probe_first_process.jsonl 2>&1
timeout 200 python profiles/catch_up_microbench.py > $O/catch_up_microbench_first_process.txt 2>&1
# 1. the driver's command (all legs: value, value_gt_streamed, value_heavy, host_resident, trainer, cpu_baseline)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_28m_final.log 2> $O/bench_28m_final.err
# 2. kernel trace of the timed steps only (in situ), one pipelined batch as a timeline
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/pmcF /tmp/pmcW /tmp/pmcS
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --steps 8 --warmup 3 $N --no-kernel-timing --gt resident > $O/prof_s.log 2>&1
DB=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 160 "adam_catch_up48_kernel<int>" > $O/kernel_stats.csv  # the 8 timed steps, up to the final flush
python $R/profiles/timeline.py $DB step3 > $O/timeline_step.txt 2>&1
python $R/profiles/timeline_streams.py $DB step3 > $O/timeline_streams.txt 2>&1
cd $R
# 3. the same kernels with nothing co-running
bash $R/profiles/solo_trace.sh r06solo > /dev/null 2>&1; cp $R/gpurun_out/r4/solo_kernel_stats_r06solo.csv $O/kernel_stats_single_stream.csv
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
timeout 300 python bench.py --steps 10 --warmup 3 $N > $O/bench_28m_for_pmc.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic_before.json
python profiles/make_pmc_json.py $O/pmc_fetch_size.txt $O/pmc_write_size.txt $O/pmc_sq_counters.txt "round 6 HEAD" $O/bench_28m_for_pmc.log > $O/pmc_traffic_summary.txt 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
# 5. A/B legs (same box)
timeout 300 python bench.py --steps 10 --warmup 3 $N --opt overlap_cameras=false > $O/bench_28m_no_overlap.log 2>&1
timeout 400 python bench.py --scene heavy --steps 10 --warmup 3 $N > $O/bench_28m_heavy.log 2>&1
# 6. the other BASELINE.json configurations
timeout 400 python bench.py --config rubble10m --steps 10 --warmup 3 $N > $O/bench_rubble10m_clm.log 2>&1
timeout 300 python bench.py --config bicycle6m --strategy no_offload --steps 10 --warmup 3 $N > $O/bench_bicycle6m_no_offload.log 2>&1
timeout 300 python bench.py --config bicycle6m --steps 10 --warmup 3 $N > $O/bench_bicycle6m_clm.log 2>&1
timeout 400 python bench.py --config bigcity102m --steps 6 --warmup 2 $N > $O/bench_bigcity102m_1gpu.log 2>&1
timeout 900 python bench.py --config bigcity102m --bsz 64 --steps 4 --warmup 2 $N > $O/bench_bigcity102m_bsz64_1gpu.log 2>&1
bash profiles/host_budget_curve.sh > /dev/null 2>&1; cp gpurun_out/r06/host_budget_curve.jsonl $O/host_budget_curve.jsonl
# 7. probes
timeout 200 python profiles/raster_microbench.py > $O/raster_microbench.txt 2>&1
timeout 200 python profiles/raster_microbench.py heavy 10 >> $O/raster_microbench.txt 2>&1
timeout 200 python profiles/catch_up_microbench.py > $O/catch_up_microbench.txt 2>&1
timeout 300 /tmp/gather_probe > $O/gather_probe_last_process.jsonl 2>&1
# 8. rocprof's duration and the event-timed duration of the SAME launches (20 back-to-back launches of each tile kernel)
cd /tmp
rm -rf /tmp/prof_mb
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_mb -o mb -- python $R/profiles/raster_microbench.py slab 20 > $O/raster_microbench_under_rocprof.txt 2>&1
DB=$(find /tmp/prof_mb -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 100000 | grep -i "rasterize\|Name" >> $O/raster_microbench_under_rocprof.txt
cd $R
# 8b. SQ counters of the two tile kernels on the microbench (one camera of the bench scene), four separate --pmc passes
cd /tmp
echo "# rocprofv3 --kernel-trace --pmc <4 counters per pass> -- python profiles/raster_microbench.py slab 5; per launch (mean, median)" > $O/pmc_sq_tile_kernels.txt
i=0
for SET in "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" \
           "SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1)); rm -rf /tmp/pmcT$i
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcT$i -o t -- python $R/profiles/raster_microbench.py slab 5 > $O/pmcT$i.log 2>&1
  python $R/profiles/pmc_summary.py $(find /tmp/pmcT$i -name "*counter_collection.csv" | head -1) | grep rasterize >> $O/pmc_sq_tile_kernels.txt 2>&1
done
cd $R
# 9. the GPU test suite
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu_final.log
ls -la $O
