#!/bin/bash
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/ab
python -m pytest tests/test_gpu_prove.py tests/test_gpu_primitives.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r8_tests.log
AB_ARGS="--host-io 0 --json-sample 0 --latency 0" bash tools/ab_variants.sh main nopipe per32 > gpurun_out/r8_ab.log 2>&1
python tools/ab_compare.py gpurun_out/ab/main.json gpurun_out/ab/nopipe.json gpurun_out/ab/per32.json > gpurun_out/r04_ab_normalize.txt
AB_ARGS="--host-io 0 --json-sample 0 --latency 0 --verify-steps 0 --roofline-steps 0" bash tools/ab_repeat.sh 2 main nopipe per32 >> gpurun_out/r04_ab_normalize.txt 2>&1
python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "100001" 2>&1 | tail -4 >> gpurun_out/r8_tests.log
cat gpurun_out/r8_tests.log gpurun_out/r04_ab_normalize.txt
