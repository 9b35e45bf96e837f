#!/bin/bash
# Round 5: the "wide" table sums of small chunks (four / eight lanes per sum) -- parity tests, then latency A/B against the same library without them.
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_small_batches.py tests/test_gpu_prove.py -x -q -m gpu > gpurun_out/r05/t5_prove.log 2>&1; echo "prove tests rc=$?"; tail -4 gpurun_out/r05/t5_prove.log
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_mutants.py tests/test_wire_packed.py -x -q -m gpu > gpurun_out/r05/t5_verify.log 2>&1; echo "verify tests rc=$?"; tail -4 gpurun_out/r05/t5_verify.log
for rep in 1 2; do
  for v in main nowide; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so   # make -C zkp-ecdsa_amd/csrc BUILD=build_nowide LIB=../build_ab/lib_nowide.so EXTRA=-DZK_WIDE_MAX_UNITS=0u; [ "$v" = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib timeout 300 python tools/exp_latency.py 2>/dev/null | tail -1 > gpurun_out/r05/lat_${v}_$rep.json
    python - $v $rep <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05/lat_%s_%s.json'%(sys.argv[1],sys.argv[2])).read())
print(sys.argv[1], sys.argv[2], {k:d[k] for k in ('1','8','64')})
if sys.argv[2]=='1': print('   prove fam', d['prove_families_ms']); print('   verify fam', d['verify_families_ms'])
PY
  done
done
for v in main nowide; do
  lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so   # make -C zkp-ecdsa_amd/csrc BUILD=build_nowide LIB=../build_ab/lib_nowide.so EXTRA=-DZK_WIDE_MAX_UNITS=0u; [ "$v" = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
  ZKATTEST_LIB=$lib timeout 300 python tools/exp_latency_sizes.py 256 16 32 128 200 256 512 1024 2048 2>/dev/null | tail -1 | tee gpurun_out/r05/latsz_$v.json | cut -c1-600
done
