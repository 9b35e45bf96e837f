# same-box A/B of the PCIe-inclusive one-call rates for library variants under zkp-ecdsa_amd/lib_exp (lib_noside.so: -DV_SIDE_MAXP=0; lib_sideall.so: auxiliary streams for every small chunk)
F="--steps 1 --warmup 0 --no-cpu-baseline --json-sample 0 --latency 0 --roofline-steps 0 --verify-steps 0 --host-io-stream 0 --host-io-packed 0"
for rep in 1 2; do
for lib in main noside sideall; do
  if [ $lib = main ]; then unset ZKATTEST_LIB; else export ZKATTEST_LIB=zkp-ecdsa_amd/build_ab/lib_$lib.so; fi
  python bench.py $F 2>/dev/null | grep '"metric"' | python -c "
import json,sys
l=json.loads(sys.stdin.read()); h=l['host_io']['pinned']
print('$lib', 'prove', h['proofs_per_s'], 'verify', h['verifies_per_s'], 'h2d', h['h2d_gbps'], 'd2h', h['d2h_gbps'])"
done; done
