set -x
O=gpurun_out/r05m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "tile_major" 2>&1 | tail -8
bash profiles/solo_trace.sh r05m > /dev/null 2>&1; grep -E "isect3|fillBuffer" gpurun_out/r4/solo_kernel_stats_r05m.csv | cut -c1-200
bash profiles/solo_trace.sh r05m_heavy --scene heavy > /dev/null 2>&1; grep -E "isect3" gpurun_out/r4/solo_kernel_stats_r05m_heavy.csv | cut -c1-200
