#!/bin/bash
# Round 5, GPU call 3: tests of this round's verifier / pool / advisor changes, the bucket-order A/B, a kernel trace of the verify step, the pool line.
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r05
ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_wire_packed.py -x -q -m gpu > gpurun_out/r05/t3_verify.log 2>&1; echo "verify tests rc=$?"; tail -4 gpurun_out/r05/t3_verify.log
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "pool or rccl" > gpurun_out/r05/t3_pool.log 2>&1; echo "pool tests rc=$?"; tail -4 gpurun_out/r05/t3_pool.log
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/r05/t3_stream.log 2>&1; echo "stream tests rc=$?"; tail -4 gpurun_out/r05/t3_stream.log
BARGS="--no-cpu-baseline --host-io 0 --latency 0 --json-sample 0 --steps 1 --warmup 0 --roofline-steps 0 --verify-steps 2"
for o in global local; do
  ZKATTEST_MSM_ORDER=$o timeout 300 python bench.py $BARGS > gpurun_out/r05/v_$o.json 2> gpurun_out/r05/v_$o.err
  python - $o <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05/v_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
v=d['verify']; print(sys.argv[1], 'verify', v['value'], v['ms_per_step'], json.dumps(v['gpu_ms_by_family_per_step']))
PY
done
ZK_MSM_DEBUG=1 ZKATTEST_MSM_ORDER=global timeout 300 python bench.py $BARGS 2>&1 | grep "^msm:" | head -12
ZK_MSM_DEBUG=1 ZKATTEST_MSM_ORDER=local timeout 300 python bench.py $BARGS 2>&1 | grep "^msm:" | head -6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r05/trace_verify -o r -- python $ROOT/bench.py $BARGS > $ROOT/gpurun_out/r05/trace_verify.log 2>&1
cd $ROOT
python tools/rocpd_stats.py gpurun_out/r05/trace_verify/r_results.db | grep -E "k_msm|k_v_validate|Name" | head -30
timeout 600 python bench.py --pool --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --json-sample 0 --latency 0 2>gpurun_out/r05/pool.err | grep '"metric"' > gpurun_out/r05/pool.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/pool.json').read())
print('pool value', d['value'], 'device_resident_output', d.get('device_resident_output'))
PY
