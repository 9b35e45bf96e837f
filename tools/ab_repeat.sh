# usage: bash tools/ab_repeat.sh REPS name1 name2 ...   -> prove / verify rates of REPS interleaved runs per variant (no serial family pass needed)
reps=$1; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $reps); do
  for v in "$@"; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so
    [ "$v" = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 ${AB_ARGS:---host-io 0} 2> gpurun_out/ab/$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'run $r', 'prove', round(d['value']), 'ms', d['ms_per_step'], 'verify', round(d['verify']['value']) if d.get('verify') else None)"
  done
done
