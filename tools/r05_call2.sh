#!/bin/bash
# Round 5, GPU call 2: the verifier's hand-written key grouping and the dense validation pass -- parity tests, then the bench line.
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_verify.py -x -q -m gpu > gpurun_out/r05/t_verify.log 2>&1; echo "verify tests rc=$?"; tail -5 gpurun_out/r05/t_verify.log
timeout 900 python -m pytest tests/test_gpu_mutants.py tests/test_wire_packed.py tests/test_gpu_small_batches.py -x -q -m gpu > gpurun_out/r05/t_mut.log 2>&1; echo "mutant tests rc=$?"; tail -5 gpurun_out/r05/t_mut.log
timeout 600 python bench.py --host-io 0 --latency 0 --json-sample 0 --steps 2 --warmup 1 > gpurun_out/r05/bench2.json 2> gpurun_out/r05/bench2.err; echo "bench rc=$?"
tail -3 gpurun_out/r05/bench2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench2.json').read().strip().splitlines()[-1])
print('prove', d['value'], d['ms_per_step'])
v=d['verify']; print('verify', v['value'], v['ms_per_step'], v['accepted'])
print(json.dumps(v['gpu_ms_by_family_per_step']))
print(json.dumps(v.get('roofline')))
print(json.dumps(d['cpu_baseline'].get('verify')))
print(d['config']['workload'], len(d['config']['workload']))
PY
