#!/bin/bash
# bench.py's device-resident prove / verify steps under GPU_MAX_HW_QUEUES = 4 / 6 / 8 / 12 (the library exports 8 unless the variable is set), same box, twice
F="--steps 4 --warmup 1 --verify-steps 5 --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline --roofline-steps 0"
for rep in 1 2; do
  for q in 8 4 6 12; do
    GPU_MAX_HW_QUEUES=$q python bench.py $F 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); v=d['verify']; print('queues $q: prove %.1f k/s (%.2f ms)  verify %.1f k/s (median %.2f ms)' % (d['value']/1e3, d['ms_per_step'], v['value']/1e3, v['median_ms']))"
  done
done
