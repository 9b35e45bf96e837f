#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (14)): the per-family HIP events of a blocking call (jobs.h: timed) off, for calls of a few proofs -- library built with
# -DZK_TIMED_DEFAULT=false (build_ab/lib_notime.so) against the shipped one, same box, interleaved three times.
export GPU_MAX_HW_QUEUES=8
for rep in 1 2 3; do
  for v in main notime; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_$v.so; [ $v = main ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib timeout 300 python tools/lat_ab.py 65536 31 2>&1 | tail -1
  done
done
