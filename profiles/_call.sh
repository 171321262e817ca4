set -x
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engines.py tests/test_gpu_golden_engine.py tests/test_abi.py -q 2>&1 | tail -5
B="--steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-trainer-leg --no-heavy-leg --gt resident --prime-seconds 5"
for rep in 1 2; do
timeout 300 python bench.py $B 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['measured']['loss_last'], {k.replace('clmgs_',''):round(v,3) for k,v in d['kernels_solo_ms'].items() if 'adam' in k}, {k.replace('clmgs_',''):v['avg_ms'] for k,v in d['kernels'].items() if 'adam' in k})"
done
