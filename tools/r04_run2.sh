#!/bin/bash
# round-4 GPU session 2: new tests, ring-100001 line, default line, pool first-call experiment
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "100001 or device_resident" 2>&1 | tail -15 > gpurun_out/r2_tests.log
python bench.py --ring 100001 --host-io 0 --json-sample 0 2> gpurun_out/r2_ring100001.err | grep '"metric"' > gpurun_out/r04_bench_line_ring100001.json
python bench.py --host-io 0 --json-sample 0 2> gpurun_out/r2_default.err | grep '"metric"' > gpurun_out/r2_bench_line_nohostio.json
for i in 1 2 3; do python tools/exp_pool_first_call.py --tag fresh$i 2>/dev/null | grep '^{' >> gpurun_out/r04_pool_first_call.jsonl; done
python bench.py --mode verify --batch 65536 --ring 1048576 --steps 1 --warmup 0 > /dev/null 2>&1
for i in 1 2 3; do python tools/exp_pool_first_call.py --tag after-ring2e20-$i 2>/dev/null | grep '^{' >> gpurun_out/r04_pool_first_call.jsonl; done
python tools/json_rate.py 2048 1 0 > /dev/null 2>&1
for m in register hostmalloc numauser nohuge; do
  ZKATTEST_POOL_ALLOC=$m python tools/exp_pool_first_call.py --tag after-json-$m 2>/dev/null | grep '^{' >> gpurun_out/r04_pool_first_call.jsonl
done
cat gpurun_out/r2_tests.log
