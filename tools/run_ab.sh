cd /root/repo
for l in 2 4 8; do
echo "== lanes $l"
timeout 600 python bench.py --config bicycle6m --steps 5 --warmup 2 --no-cpu-baseline --opt overlap_lanes=$l > gpurun_out/bl_$l.log 2>&1; python profiles/show_bench.py gpurun_out/bl_$l.log | head -1
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --opt overlap_lanes=3 > gpurun_out/l3.log 2>&1; python profiles/show_bench.py gpurun_out/l3.log | head -1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/l2.log 2>&1; python profiles/show_bench.py gpurun_out/l2.log | head -1
