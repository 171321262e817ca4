#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05w; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engines.py -q -x -k "split_catch or schedules or deferred_small or capture" > $O/pytest_b.log 2>&1; tail -2 $O/pytest_b.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-trainer-leg --no-heavy-leg"
P=$R/clm_gs_amd/libclmgs_hip_prof.so
timeout 300 $B > $O/front_1.log 2>&1
timeout 300 $B --opt split_catch_up=false > $O/single_1.log 2>&1
timeout 300 $B > $O/front_2.log 2>&1
CLMGS_LIB_PATH=$P timeout 300 $B > $O/prof_base.log 2>&1
CLMGS_LIB_PATH=$P CLMGS_BWD_LDS_PAD=3900 timeout 300 $B > $O/prof_b4.log 2>&1
CLMGS_LIB_PATH=$P CLMGS_BWD_LDS_PAD=6800 timeout 300 $B > $O/prof_b3.log 2>&1
CLMGS_LIB_PATH=$P CLMGS_BWD_LDS_PAD=3900 CLMGS_FWD_LDS_PAD=5300 timeout 300 $B > $O/prof_b4f5.log 2>&1
CLMGS_LIB_PATH=$P CLMGS_BWD_LDS_PAD=3900 CLMGS_FWD_LDS_PAD=7100 timeout 300 $B > $O/prof_b4f4.log 2>&1
for f in front_1 single_1 front_2 prof_base prof_b4 prof_b3 prof_b4f5 prof_b4f4; do python - $O/$f.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); k=d['kernels_solo_ms']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], k.get('clmgs_rasterize_bwd'), k.get('clmgs_rasterize_fwd'), d['roofline']['avg_launch_ms'])
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-leg --no-trainer-leg --no-heavy-leg --no-kernel-timing --gt resident > $O/prof_s.log 2>&1
DB=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/profiles/timeline.py $DB step3 > $O/timeline_step.txt 2>&1
python $R/profiles/timeline_streams.py $DB step3 > $O/timeline_streams.txt 2>&1
