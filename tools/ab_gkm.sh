#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (10)): the digit planes of the two matrix-core kernels in two halves, the free half prefetching the next chunk (GKM_DOUBLE_BUFFER=1, shipped)
# against the same library with k_gk_mfma.hip built -DGKM_DOUBLE_BUFFER=0 (build_ab/lib_gkm0.so), same box, interleaved twice; rings of 2^16, 2^17 and (verify only) 2^20 keys.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06ab
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_prove.py -q -m gpu -k "matrix or table_path or gk" > $O/gkm_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/gkm_tests.log
ARGS="--steps 3 --warmup 1 --verify-steps 5 --roofline-steps 1 --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline"
for rep in 1 2; do
  for v in gkm0 main; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so; [ $v = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    for ring in 65536 100001; do
      ZKATTEST_LIB=$lib timeout 600 python bench.py $ARGS --ring $ring 2>/dev/null | grep '"metric"' > $O/gkm_${v}_${ring}_$rep.json
      python - $v $rep $O $ring <<'PY'
import json,sys
d=json.loads(open('%s/gkm_%s_%s_%s.json'%(sys.argv[3],sys.argv[1],sys.argv[4],sys.argv[2])).read())
f=d['gpu_ms_by_family_per_step']; v=d['verify']; g=v['gpu_ms_by_family_per_step']
print('gkm %-5s ring %-6s %s prove %.1f k/s (%.2f ms)  verify %.1f k/s (median %.2f ms)  gk_fold %.2f  v_gk_total %.2f  failed %d accepted %d' % (
    sys.argv[1], sys.argv[4], sys.argv[2], d['value']/1e3, d['ms_per_step'], v['value']/1e3, v['median_ms'], f['gk_fold'], g['v_gk_total'], d['failed_proofs'], v['accepted']))
PY
    done
    ZKATTEST_LIB=$lib timeout 600 python bench.py --mode verify --batch 131072 --ring 1048576 --steps 1 --warmup 1 2>/dev/null | grep '"metric"' > $O/gkm_${v}_v20_$rep.json
    python - $v $rep $O <<'PY'
import json,sys
d=json.loads(open('%s/gkm_%s_v20_%s.json'%(sys.argv[3],sys.argv[1],sys.argv[2])).read())
print('gkm %-5s ring 2^20 verify %s %.1f k verifies/s  accepted %s of %s' % (sys.argv[1], sys.argv[2], d['value']/1e3, d.get('accepted'), d.get('of')))
PY
  done
done
