#!/bin/bash
# Kernel trace of the locality exchange on a 1-rank RCCL group (CLMGS_DP_FORCE=1) next to the plain single-GPU run:
# what the exchange costs before any byte moves.  GPU box, repo root.
R=$(pwd); O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dp1 /tmp/prof_dp0
CLMGS_DP_FORCE=1 timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_dp1 -o st -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29533 $R/bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-host-leg --no-trainer-leg \
  --no-kernel-timing --gt resident --dp-mode locality > $O/dp_world1.log 2>&1
DB=$(find /tmp/prof_dp1 -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 205 > $O/kernel_stats_locality_world1.csv
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_dp0 -o st -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-leg --no-trainer-leg \
  --no-kernel-timing --gt resident > $O/dp_world0.log 2>&1
DB=$(find /tmp/prof_dp0 -name "*.db" | head -1)
python $R/profiles/kernel_stats.py "$DB" 195 > $O/kernel_stats_plain.csv
cd $R
grep -h '"value"' $O/dp_world1.log $O/dp_world0.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['value'], j['ms_per_step'], j.get('dp'))"
