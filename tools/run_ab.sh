cd /root/repo
timeout 900 python bench.py --strategy naive_offload --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/naive_28m.log 2>&1; python profiles/show_bench.py gpurun_out/naive_28m.log | head -3
timeout 600 python bench.py --config bicycle6m --strategy naive_offload --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/naive_bicycle.log 2>&1; python profiles/show_bench.py gpurun_out/naive_bicycle.log | head -1
timeout 600 python bench.py --config bicycle6m --strategy no_offload --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/no_offload_bicycle.log 2>&1; python profiles/show_bench.py gpurun_out/no_offload_bicycle.log | head -8
timeout 600 python bench.py --config bicycle6m --strategy clm_offload --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/clm_bicycle.log 2>&1; python profiles/show_bench.py gpurun_out/clm_bicycle.log | head -1
timeout 600 python bench.py --config rubble10m --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/clm_10m.log 2>&1; python profiles/show_bench.py gpurun_out/clm_10m.log | head -1
