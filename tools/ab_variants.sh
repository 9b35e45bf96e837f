# usage: bash tools/ab_variants.sh name1 name2 ...   (libraries zkp-ecdsa_amd/build_ab/lib_<name>.so; "base" here = lib_base.so, "main" = the built library)
# AB_ARGS: extra bench.py arguments (default: no host-buffer pass)
mkdir -p gpurun_out/ab
for v in "$@"; do
  lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so
  [ "$v" = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
  ZKATTEST_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 ${AB_ARGS:---host-io 0} > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err
  echo "$v rc=$?"
done
