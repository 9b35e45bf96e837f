# same-box experiment: chunk plans of the device-resident prover (uniform | staggered first chunks, ZK_DEVICE_STAGGER) and steps kept in flight (--device-stream)
F="--no-cpu-baseline --steps 3 --warmup 1 --host-io 0 --verify-steps 0 --json-sample 0 --latency 0 --roofline-steps 0 --check 0"
run() { python bench.py $F "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', round(d['value']), d['ms_per_step'], json.dumps(d.get('device_stream')) if d.get('device_stream') else '')"; }
for rep in 1 2; do
TAG="uniform 22016x3" run
TAG="stagger3 22016x3" ZK_DEVICE_STAGGER=3 run
TAG="uniform 16384x3" run --chunk 16384
TAG="stagger3 16384x3" ZK_DEVICE_STAGGER=3 run --chunk 16384
TAG="uniform 11008x3" run --chunk 11008
TAG="stagger3 11008x3" ZK_DEVICE_STAGGER=3 run --chunk 11008
TAG="device-stream3 22016x3" run --device-stream 3
done
