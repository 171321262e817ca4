set -x
timeout 900 python -m pytest tests/test_gpu_engines.py -q 2>&1 | tail -5
