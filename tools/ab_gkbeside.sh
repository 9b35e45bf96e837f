#!/bin/bash
# Round 6 experiment (profiles/r06_ab_variants.txt (9)): stage 2's membership phase of every unsliced chunk on the lane's side stream beside the PointAdd phase
# (-DZK_EXP_GK_BESIDE in api.hip -> build_ab/lib_gkbeside.so) against the shipped library, same box, interleaved; bytes compared through a digest of the whole output.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06ab
mkdir -p $O
ARGS="--steps 6 --warmup 2 --verify-steps 5 --roofline-steps 0 --host-io 0 --json-sample 0 --latency 0 --cpu-sample 8"
for rep in 1 2 3; do
  for v in main gkbeside; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so; [ $v = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib timeout 600 python bench.py $ARGS 2>$O/gkb_${v}_$rep.err | grep '"metric"' > $O/gkb_${v}_$rep.json
    python - $v $rep $O <<'PY'
import json,sys
d=json.loads(open('%s/gkb_%s_%s.json'%(sys.argv[3],sys.argv[1],sys.argv[2])).read())
print('gkbeside %-9s %s %.1f k proofs/s  %.2f ms per step  failed %d  verify accepted %d  bit-exact vs oracle %s' % (sys.argv[1], sys.argv[2], d['value']/1e3, d['ms_per_step'], d['failed_proofs'], d['verify']['accepted'], d['cpu_baseline']['checked_bit_exact']))
PY
  done
done
