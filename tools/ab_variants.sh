# usage: bash tools/ab_variants.sh name1 name2 ...   (libraries zkp-ecdsa_amd/lib_exp/lib_<name>.so; "base" = the built library)
mkdir -p gpurun_out/ab
for v in "$@"; do
  lib=$PWD/zkp-ecdsa_amd/lib_exp/lib_$v.so
  [ "$v" = base ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
  ZKATTEST_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err
  echo "$v rc=$?"
done
