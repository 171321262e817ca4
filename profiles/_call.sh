set -x
O=gpurun_out/r05p; mkdir -p $O
B="--steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-heavy-leg --no-kernel-timing --gt resident --prime-seconds 3"
for rep in 1 2; do for o in "binning=tile" "binning=sort" "deferred_small_adam=false"; do
timeout 300 python bench.py $B --opt $o 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['trainer']
print('$o', d['value'], 'trainer', t['trainer_img_s'], t['host_seconds_by_phase'], t['device_mallocs'], t['trainer_peak_gpu_bytes'])"
done; done
