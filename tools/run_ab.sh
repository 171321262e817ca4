cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt overlap_cameras=false > gpurun_out/ab0.log 2>&1
python profiles/show_bench.py gpurun_out/ab0.log 2>&1 | grep "img/s\|loss\|C-ABI" | cut -c1-100
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab1.log 2>&1
python profiles/show_bench.py gpurun_out/ab1.log 2>&1 | head -1
