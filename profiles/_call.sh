#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "nothing_on_the_image or morton or row_mover or tile_major" > $O/pytest_a.log 2>&1; tail -2 $O/pytest_a.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-host-leg --no-heavy-leg > $O/bench_short.log 2>&1; tail -c 400 $O/bench_short.log
