#!/bin/bash
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_napi_binding.py tests/test_gpu_prove.py -x -q -m gpu -k "javascript or knobs or uniform" -s > gpurun_out/r05/t7.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05/t7.log | cut -c1-1500
BARGS="--no-cpu-baseline --host-io 0 --latency 0 --json-sample 0 --steps 1 --warmup 0 --roofline-steps 0 --verify-steps 2"
for cfg in "32768 2" "22016 3" "16384 4" "65536 1" "32768 3"; do
  set -- $cfg
  timeout 300 python bench.py $BARGS --verify-chunk $1 --verify-lanes $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['verify']; print('verify chunk', v['chunk'], 'lanes', v['lanes'], v['value'], v['ms_per_step'], 'hbm', d['hbm_used_gb'])"
done
