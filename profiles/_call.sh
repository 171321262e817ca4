set -x
O=gpurun_out/r05b; mkdir -p $O
CLMGS_DIST_BACKEND=gloo CLMGS_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --config small --prime-seconds 0 > $O/dp2.log 2> $O/dp2.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r05b/dp2.log") if x.startswith("{")]
print(json.dumps(json.loads(l[-1])["dp"], indent=1)[:3000] if l else open("gpurun_out/r05b/dp2.err").read()[-3000:])
PY
tail -20 $O/dp2.err
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
