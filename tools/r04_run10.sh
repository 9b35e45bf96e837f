#!/bin/bash
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python -m pytest tests/test_napi_binding.py tests/test_gpu_scale.py -x -q -m gpu -k "javascript or 100001 or device_resident" 2>&1 | tail -6 > gpurun_out/r10_tests.log
make -s -C bindings/napi OUT=/tmp/zk.node && (cd bindings/napi && ZKATTEST_NODE=/tmp/zk.node node latency.js 7 > ../../gpurun_out/r04_facade_latency.json 2> ../../gpurun_out/r10_lat.err)
python bench.py --host-io 0 --json-sample 0 --no-cpu-baseline 2> gpurun_out/r10_bench.err | grep '"metric"' > gpurun_out/r10_bench.json
python3 - <<'PY' >> gpurun_out/r10_tests.log
import json
d=json.loads(open('gpurun_out/r10_bench.json').read())
print('value', d['value'], 'verify', d['verify']['value'], 'latency_ms_b1', d.get('latency_ms_b1'), d.get('verify_latency_ms_b1'))
print(json.dumps(d['latency'], indent=None))
print(json.dumps(d['roofline']['others'], indent=1))
print(open('gpurun_out/r04_facade_latency.json').read())
PY
cat gpurun_out/r10_tests.log; tail -3 gpurun_out/r10_lat.err
