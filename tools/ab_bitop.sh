#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (6)): SHA-256's three-input boolean functions as v_bitop3_b32 (zkdev.h: zk_xor3, zk_maj) against the library built just before
# (build_ab/lib_prebitop.so: two v_xor_b32 per sigma, v_xor + v_bfi for Maj), same box, interleaved twice; the hash-bound families and one proof / verification per call.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06ab
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_prove.py -q -m gpu -k "sha or rng or golden or config2 or cooperative" > $O/bitop_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/bitop_tests.log
ARGS="--steps 4 --warmup 1 --verify-steps 5 --roofline-steps 1 --host-io 0 --json-sample 0 --latency 1 --no-cpu-baseline"
for rep in 1 2; do
  for v in prebitop main; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so; [ $v = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib timeout 600 python bench.py $ARGS 2>/dev/null | grep '"metric"' > $O/bitop_${v}_$rep.json
    python - $v $rep $O <<'PY'
import json,sys
d=json.loads(open('%s/bitop_%s_%s.json'%(sys.argv[3],sys.argv[1],sys.argv[2])).read())
f=d['gpu_ms_by_family_per_step']; v=d['verify']; g=v['gpu_ms_by_family_per_step']
print('bitop', sys.argv[1], sys.argv[2], 'prove %.1f k/s (%.2f ms)  verify %.1f k/s (median %.2f ms)  b1 %.2f / %.2f ms   hash %.2f rng_prepass %.2f  v_hash %.2f v_parse %.2f' % (
    d['value']/1e3, d['ms_per_step'], v['value']/1e3, v['median_ms'], d.get('latency_ms_b1',0), d.get('verify_latency_ms_b1',0), f['hash'], f['rng_prepass'], g['v_hash'], g['v_parse_validate']))
PY
  done
done
