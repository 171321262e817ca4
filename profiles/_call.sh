set -x
O=gpurun_out/r05h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engines.py -q -x 2>&1 | tail -15
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-trainer-leg --no-heavy-leg --gt resident --prime-seconds 5"
timeout 300 python bench.py $B > $O/bench_slab.log 2>&1
timeout 300 python bench.py $B --opt deferred_small_adam=false > $O/bench_slab_eager.log 2>&1
python - <<'PY'
import json
for f in ("bench_slab","bench_slab_eager"):
    l=[x for x in open("gpurun_out/r05h/%s.log"%f) if x.startswith("{")]
    if not l: print(open("gpurun_out/r05h/%s.log"%f).read()[-1500:]); continue
    d=json.loads(l[-1]); print(f, d["value"], d["ms_per_step"], d["measured"]["loss_last"], {k:round(v,3) for k,v in d["kernels_solo_ms"].items()}); print({k:(v["calls"],v["avg_ms"]) for k,v in d["kernels"].items() if "adam" in k or "emit" in k})
PY
