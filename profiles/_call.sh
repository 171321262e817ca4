#!/bin/bash
O=gpurun_out/r05t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "morton or row_mover" > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_engines.py tests/test_gpu_golden_engine.py -q -x > $O/pytest_b.log 2>&1; tail -3 $O/pytest_b.log
timeout 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-host-leg --no-heavy-leg --no-kernel-timing --trainer-trace > $O/trace.log 2>&1
timeout 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-host-leg --no-heavy-leg --no-kernel-timing > $O/plain.log 2>&1
tail -c 200 $O/plain.log
