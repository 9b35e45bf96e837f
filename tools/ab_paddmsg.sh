#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (20)): the verifier's PointAdd challenges of a small call (<= 64 proofs) through the message / schedule / two-lane rounds
# kernels (k_v_padd_msg + k_hash.hip: launch_sha_msgs) instead of one lane per digest (k_v_padd_hash), against the library of the commit before
# (build_ab/lib_base.so), same box, interleaved three times.  The verifier's parity and exception-order tests first, then the kernel timeline of one call.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06ab
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mutants.py tests/test_gpu_small_batches.py tests/test_gpu_verify.py tests/test_wire_packed.py -q -m gpu -x > $O/paddmsg_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/paddmsg_tests.log
for rep in 1 2 3; do
  for v in base main; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so; [ $v = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib timeout 300 python tools/lat_ab.py 65536 31 2>&1 | tail -1
  done
done
ZKATTEST_LIB=$PWD/zkp-ecdsa_amd/build_ab/lib_base.so timeout 300 python tools/lat_ab.py 1024 31 2>&1 | tail -1
timeout 300 python tools/lat_ab.py 1024 31 2>&1 | tail -1
R=$PWD
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/b1tl && timeout 300 rocprofv3 --kernel-trace -d /tmp/b1tl -o r -- python $R/tools/b1_timeline.py run > /dev/null 2>&1; python $R/tools/b1_timeline.py parse $(find /tmp/b1tl -name 'r_results.db' | head -1) > $R/gpurun_out/r06_b1_timeline_v10.txt 2>&1)
grep -E "^== " gpurun_out/r06_b1_timeline_v10.txt
