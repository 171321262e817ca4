cd /root/repo
for ps in true false true false; do
echo "== packed_stats $ps"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt overlap_cameras=false --opt packed_stats=$ps > gpurun_out/ab0.log 2>&1
python profiles/show_bench.py gpurun_out/ab0.log 2>&1 | grep "img/s\|preprocess_bwd" | cut -c1-90
done
