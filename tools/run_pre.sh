set -x
cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
for v in 0 1 2; do
  CLMGS_PRE_VARIANT=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt overlap_cameras=false > gpurun_out/pre_v$v.log 2>&1
  python profiles/show_bench.py gpurun_out/pre_v$v.log 2>&1 | tail -20
done
