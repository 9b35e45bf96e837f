#!/bin/bash
# Round 6 experiment (profiles/r06_ab_variants.txt (5)): what would the prover gain if the host were out of the stage 1 -> stage 2 hand-over?  Upper bound by a library built with
# -DZK_EXP_CACHED_SCAN (build_ab/lib_cachedscan.so: a repeated identical call takes its chunk totals from a cache and never waits for a scan; the #ifdef lives in commit f4b3401's
# api.hip only -- check that commit out to rebuild it) against the shipped library, same box,
# interleaved.  The cached library is an experiment, not a product: its totals are only right because bench.py repeats one workload.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06ab
mkdir -p $O
ARGS="--steps 6 --warmup 2 --verify-steps 5 --roofline-steps 0 --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline"
for rep in 1 2 3; do
  for v in main cachedscan; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so; [ $v = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib timeout 600 python bench.py $ARGS 2>/dev/null | grep '"metric"' > $O/ho_${v}_$rep.json
    python - $v $rep $O <<'PY'
import json,sys
d=json.loads(open('%s/ho_%s_%s.json'%(sys.argv[3],sys.argv[1],sys.argv[2])).read())
print('handover', sys.argv[1], sys.argv[2], '%.1f k proofs/s  %.2f ms per step  failed %d  verify accepted %d of %d' % (d['value']/1e3, d['ms_per_step'], d['failed_proofs'], d['verify']['accepted'], d['verify']['of']))
PY
  done
done
for v in main cachedscan; do
  lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so; [ $v = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
  ZK_IO_DEBUG=2 ZKATTEST_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 2 --verify-steps 0 --roofline-steps 0 --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline > /dev/null 2> $O/ho_timeline_$v.txt
  echo "== $v: host timeline of the last call"; grep "^host" $O/ho_timeline_$v.txt | tail -12
done
