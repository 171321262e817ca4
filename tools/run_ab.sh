cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for lib in hip np; do for v in 0 2; do
echo "== lib $lib variant $v"
CLMGS_LIB_PATH=/root/repo/clm_gs_amd/libclmgs_$lib.so CLMGS_PRE_VARIANT=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt overlap_cameras=false > gpurun_out/ab0.log 2>&1
python profiles/show_bench.py gpurun_out/ab0.log 2>&1 | grep "img/s\|preprocess" | cut -c1-90
done; done
