#!/bin/bash
# Round 6 same-box A/Bs (profiles/r06_ab_variants.txt): the dependent chains on cooperating waves (csrc/coop.h) and the divsteps inversion (field.h) against the round-5 forms.
#   main      : lib/libzkattest_hip.so
#   onelane   : the same library with ZKATTEST_ONE_LANE_CHAINS=1 (every dependent chain back in one lane)
#   r5        : build_ab/lib_fermat.so (make BUILD=build_fermat LIB=../build_ab/lib_fermat.so EXTRA=-DZK_INV_FERMAT=1) with ZKATTEST_ONE_LANE_CHAINS=1: round 5's arithmetic
# (1) one proof / one verification per call (tools/exp_latency.py), (2) the kernels themselves (rocprofv3 --kernel-trace of a one-proof call and of two verify steps),
# (3) the bench's prove / verify throughput, interleaved twice.  About 9 minutes of GPU time.
export GPU_MAX_HW_QUEUES=8
ROOT=$PWD
O=gpurun_out/r06ab
mkdir -p $O
MAIN=$ROOT/zkp-ecdsa_amd/lib/libzkattest_hip.so
R5=$ROOT/zkp-ecdsa_amd/build_ab/lib_fermat.so
run() {  # tag lib onelane cmd...
  tag=$1; lib=$2; one=$3; shift 3
  if [ "$one" = 1 ]; then ZKATTEST_LIB=$lib ZKATTEST_ONE_LANE_CHAINS=1 "$@"; else ZKATTEST_LIB=$lib "$@"; fi
}
for rep in 1 2; do
  for v in "main $MAIN 0" "onelane $MAIN 1" "r5 $R5 1"; do
    set -- $v
    run $1 $2 $3 timeout 300 python tools/exp_latency.py 2>/dev/null | tail -1 > $O/lat_$1_$rep.json
    python - $1 $rep $O <<'PY'
import json,sys
d=json.loads(open('%s/lat_%s_%s.json'%(sys.argv[3],sys.argv[1],sys.argv[2])).read())
print('latency', sys.argv[1], sys.argv[2], {k:d[k] for k in ('1','8','64')})
if sys.argv[2]=='1': print('   prove fam', d['prove_families_ms']); print('   verify fam', d['verify_families_ms'])
PY
  done
done
BARGS="--steps 2 --warmup 1 --verify-steps 5 --roofline-steps 1 --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline"
for rep in 1 2; do
  for v in "main $MAIN 0" "onelane $MAIN 1" "r5 $R5 1"; do
    set -- $v
    run $1 $2 $3 timeout 600 python bench.py $BARGS 2>/dev/null | grep '"metric"' > $O/bench_$1_$rep.json
    python - $1 $rep $O <<'PY'
import json,sys
d=json.loads(open('%s/bench_%s_%s.json'%(sys.argv[3],sys.argv[1],sys.argv[2])).read())
v=d['verify']
print('bench', sys.argv[1], sys.argv[2], 'prove %.1f k/s (%.2f ms)  verify %.1f k/s (min %.2f median %.2f max %.2f ms)' % (d['value']/1e3, d['ms_per_step'], v['value']/1e3, v['min_ms'], v['median_ms'], v['max_ms']))
f=d['gpu_ms_by_family_per_step']; g=v['gpu_ms_by_family_per_step']
print('   prove fam', {k:f[k] for k in ('tom_normalize','p256_normalize','p256_front','scalars') if k in f}, ' verify fam', {k:g[k] for k in ('v_msm_tom','v_msm_p256','v_p256_front_rtab') if k in g})
PY
  done
done
# the kernels themselves
cd /tmp && export TMPDIR=/tmp
for v in "main $MAIN 0" "onelane $MAIN 1"; do
  set -- $v
  run $1 $2 $3 timeout 600 rocprofv3 --kernel-trace -d $ROOT/$O/prof_b1_$1 -o r -- python $ROOT/tools/b1_timeline.py run > $ROOT/$O/b1_$1.log 2>&1
  python $ROOT/tools/b1_timeline.py parse $ROOT/$O/prof_b1_$1/r_results.db > $ROOT/$O/b1_timeline_$1.txt 2>&1
  run $1 $2 $3 timeout 600 rocprofv3 --kernel-trace -d $ROOT/$O/prof_v_$1 -o r -- python $ROOT/bench.py --steps 1 --warmup 0 --verify-steps 2 --roofline-steps 0 --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline > $ROOT/$O/v_$1.log 2>&1
  python $ROOT/tools/rocpd_stats.py $ROOT/$O/prof_v_$1/r_results.db | grep -E "Name|k_msm_red|k_pm_final|k_pm_reduce|k_msm_final" > $ROOT/$O/vkern_$1.csv
  rm -rf $ROOT/$O/prof_b1_$1 $ROOT/$O/prof_v_$1
done
cd $ROOT
for v in main onelane; do echo "== $v"; grep -E "k_rtab_base|k_v_straus|k_v_p256_straus|k_tom_normalize|k_p256_normalize|k_front|last (prove|verify)" $O/b1_timeline_$v.txt | grep -v "^k_.* x" ; cat $O/vkern_$v.csv; done
