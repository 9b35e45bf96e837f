#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (8)): the lock-step products' order pinned by volatile asm (-DZK_MAD_VOLATILE=1) in k_p256.hip (lib_vol_p256.so) and in k_msm.hip
# (lib_vol_msm.so) against the library without it, same box, interleaved twice.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06ab
mkdir -p $O
ARGS="--steps 3 --warmup 1 --verify-steps 5 --roofline-steps 1 --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline"
for rep in 1 2; do
  for v in main vol_p256 vol_msm; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so; [ $v = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib timeout 600 python bench.py $ARGS 2>/dev/null | grep '"metric"' > $O/vol_${v}_$rep.json
    python - $v $rep $O <<'PY'
import json,sys
d=json.loads(open('%s/vol_%s_%s.json'%(sys.argv[3],sys.argv[1],sys.argv[2])).read())
f=d['gpu_ms_by_family_per_step']; v=d['verify']; g=v['gpu_ms_by_family_per_step']
print('vol %-9s %s prove %.1f k/s (%.2f ms)  verify %.1f k/s (median %.2f ms)  p256_exp_commit %.2f p256_front %.2f | v_msm_tom %.2f +bucket %.2f v_msm_p256 %.2f  failed %d accepted %d' % (
    sys.argv[1], sys.argv[2], d['value']/1e3, d['ms_per_step'], v['value']/1e3, v['median_ms'], f['p256_exp_commit'], f['p256_front'], g['v_msm_tom'], g['+v_msm_bucket'], g['v_msm_p256'], d['failed_proofs'], v['accepted']))
PY
  done
done
