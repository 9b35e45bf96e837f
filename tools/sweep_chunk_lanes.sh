# usage: bash tools/sweep_chunk_lanes.sh   -> device-resident prove and verify rates over (chunk, lanes) settings on one box
for cfg in ${SWEEP:-"32768 2" "22016 3" "16384 4" "32768 2"}; do
  set -- $cfg
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 --host-io 0 --roofline-steps 0 --verify-steps 2 --json-sample 0 --chunk $1 --lanes $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk $1 lanes $2', 'prove', round(d['value']), 'ms', d['ms_per_step'], 'verify', round(d['verify']['value']), 'hbm', d.get('hbm_used_gb'))"
done
