#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_final.log 2>&1; tail -5 $O/pytest_gpu_final.log
