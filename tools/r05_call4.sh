#!/bin/bash
# Round 5, GPU call 4: grouping passes v2 (group-major walk, 8-byte pairs, LDS-staged pass B), Exp challenge in three kernels at every chunk size.
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r05
ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_mutants.py -x -q -m gpu > gpurun_out/r05/t4_verify.log 2>&1; echo "verify tests rc=$?"; tail -4 gpurun_out/r05/t4_verify.log
timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_small_batches.py -x -q -m gpu > gpurun_out/r05/t4_prove.log 2>&1; echo "prove tests rc=$?"; tail -4 gpurun_out/r05/t4_prove.log
BARGS="--no-cpu-baseline --host-io 0 --latency 0 --json-sample 0 --steps 2 --warmup 1"
timeout 300 python bench.py $BARGS > gpurun_out/r05/b4.json 2> gpurun_out/r05/b4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/b4.json').read().strip().splitlines()[-1])
print('prove', d['value'], d['ms_per_step'], json.dumps(d['gpu_ms_by_family_per_step']))
v=d['verify']; print('verify', v['value'], v['ms_per_step'], json.dumps(v['gpu_ms_by_family_per_step']))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r05/trace4 -o r -- python $ROOT/bench.py $BARGS --lanes 1 --verify-lanes 1 > $ROOT/gpurun_out/r05/trace4.log 2>&1
cd $ROOT
python tools/rocpd_stats.py gpurun_out/r05/trace4/r_results.db | grep -E "k_msm|k_v_validate|k_exp|k_v_ch|k_v_exph|Name" | head -40
