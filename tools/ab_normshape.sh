#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (11)): points per thread of the batch normalisers now that a workgroup's inversion costs 38 us instead of 164 (engine.h: ZK_NORM_PER_MAX /
# ZK_NORM_MIN_THREADS; k_tom.hip and k_p256.hip rebuilt with -DZK_NORM_PER_MAX=32 / 16 / 8 and four / eight / sixteen times the threads -> build_ab/lib_norm<per>.so), same box, twice.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06ab
mkdir -p $O
ARGS="--steps 3 --warmup 1 --verify-steps 5 --roofline-steps 1 --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline"
for rep in 1 2; do
  for v in main norm32 norm16 norm8; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so; [ $v = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib timeout 600 python bench.py $ARGS 2>/dev/null | grep '"metric"' > $O/ns_${v}_$rep.json
    python - $v $rep $O <<'PY'
import json,sys
d=json.loads(open('%s/ns_%s_%s.json'%(sys.argv[3],sys.argv[1],sys.argv[2])).read())
f=d['gpu_ms_by_family_per_step']; v=d['verify']; g=v['gpu_ms_by_family_per_step']
print('normshape %-7s %s prove %.1f k/s (%.2f ms)  verify %.1f k/s (median %.2f)  tom_normalize %.2f p256_normalize %.2f tom_derived %.2f | v_tom_fixed %.2f v_p256_exp_points %.2f  failed %d accepted %d' % (
    sys.argv[1], sys.argv[2], d['value']/1e3, d['ms_per_step'], v['value']/1e3, v['median_ms'], f['tom_normalize'], f['p256_normalize'], f['tom_derived'], g['v_tom_fixed'], g['v_p256_exp_points'], d['failed_proofs'], v['accepted']))
PY
  done
done
