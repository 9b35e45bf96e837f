#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (12)): the sampled repetitions of a small verify call parsed by the HEADER's challenge bits, the 16 KB hash that recomputes the
# challenge running beside R's table and the sampled points and only checked against the header in front of k_v_exp_status (k_verify.hip: k_v_sample / k_v_sample_check),
# against the library of the commit before (build_ab/lib_base.so), same box, interleaved three times.  Tests of the verifier's exception order first.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06ab
mkdir -p $O
mkdir -p $O; timeout 1500 python -m pytest tests/test_gpu_mutants.py tests/test_gpu_small_batches.py tests/test_gpu_verify.py tests/test_wire_packed.py -q -m gpu -x > $O/sample_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/sample_tests.log
for rep in 1 2 3; do
  for v in base main; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so; [ $v = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib timeout 300 python tools/lat_ab.py 65536 31 2>&1 | tail -1
  done
done
ZKATTEST_LIB=$PWD/zkp-ecdsa_amd/build_ab/lib_base.so timeout 300 python tools/lat_ab.py 1024 31 2>&1 | tail -1
timeout 300 python tools/lat_ab.py 1024 31 2>&1 | tail -1
