#!/bin/bash
# Round 5, first GPU call: parity suite, tile-kernel A/B (round-4 rasterize.hip vs per-quadrant termination) on the slab and
# the heavy scene, quadrant-pass counters (profile builds), the full default bench line.
set -x
R=$(pwd); O=$R/gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-trainer-leg --no-heavy-leg --gt resident --prime-seconds 5"
for lib in r4 new; do
  L=$R/clm_gs_amd/libclmgs_hip.so; [ $lib = r4 ] && L=$R/clm_gs_amd/libclmgs_hip_r4.so
  CLMGS_LIB_PATH=$L timeout 300 python bench.py $B > $O/ab_slab_$lib.log 2>&1
  CLMGS_LIB_PATH=$L timeout 300 python bench.py $B --scene heavy > $O/ab_heavy_$lib.log 2>&1
done
for lib in r4 new; do
  L=$R/clm_gs_amd/libclmgs_hip_prof.so; [ $lib = r4 ] && L=$R/clm_gs_amd/libclmgs_hip_prof_r4.so
  CLMGS_LIB_PATH=$L CLMGS_BWD_DEBUG=3 timeout 300 python profiles/bwd_phases.py > $O/bwd_phases_slab_$lib.log 2>&1
  CLMGS_LIB_PATH=$L CLMGS_BWD_DEBUG=3 CLMGS_PHASES_SCENE=heavy timeout 300 python profiles/bwd_phases.py > $O/bwd_phases_heavy_$lib.log 2>&1
done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_28m_default.log 2> $O/bench_28m_default.err
python profiles/show_bench.py $O/bench_28m_default.log 2>/dev/null | head -60
ls -la $O
