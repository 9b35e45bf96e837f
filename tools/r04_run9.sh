#!/bin/bash
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/ab
python -m pytest tests/test_gpu_prove.py tests/test_gpu_primitives.py tests/test_gpu_verify.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r9_tests.log
AB_ARGS="--host-io 0 --json-sample 0 --latency 0" bash tools/ab_variants.sh main norows > gpurun_out/r9_ab.log 2>&1
python tools/ab_compare.py gpurun_out/ab/main.json gpurun_out/ab/norows.json > gpurun_out/r04_ab_rows.txt
AB_ARGS="--host-io 0 --json-sample 0 --latency 0 --verify-steps 0 --roofline-steps 0" bash tools/ab_repeat.sh 2 main norows >> gpurun_out/r04_ab_rows.txt 2>&1
python bench.py --ring 100001 --batch 8192 --chunk 4096 --lanes 2 --verify-chunk 4096 --steps 1 --warmup 1 --cpu-sample 8 --host-io 0 --json-sample 0 > gpurun_out/r9_ring_small.json 2> gpurun_out/r9_ring_small.err
tail -5 gpurun_out/r9_ring_small.err >> gpurun_out/r9_tests.log
python3 -c "
import json
d=json.loads(open('gpurun_out/r9_ring_small.json').read().strip().splitlines()[-1])
print('ring small:', d['value'], d.get('key_table_proofs_last_chunk'), d['config']['workload'])
print(json.dumps(d.get('latency'), indent=0)[:3000])
" >> gpurun_out/r9_tests.log 2>&1
cat gpurun_out/r9_tests.log gpurun_out/r04_ab_rows.txt
