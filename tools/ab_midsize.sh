#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (21)): blocking verify calls of 257..8192 proofs with ONE auxiliary stream waiting for stage 1 (the P-256 relation's
# kernels in a row on it) against the library before (all four streams wait, the table walks on a stream of their own): build_ab/lib_base.so, same box, interleaved.
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests/test_gpu_mutants.py tests/test_gpu_small_batches.py tests/test_gpu_verify.py tests/test_gpu_stream.py -q -m gpu -x 2>&1 | tail -2
for B in 300 512 1024 2048 4096 8192; do
  for rep in 1 2; do
    echo "before: $(ZKATTEST_LIB=$PWD/zkp-ecdsa_amd/build_ab/lib_base.so timeout 300 python tools/lat_dist.py 65536 24 $B 2>&1 | tail -1 | sed 's/  */ /g')"
    echo "tree:   $(timeout 300 python tools/lat_dist.py 65536 24 $B 2>&1 | tail -1 | sed 's/  */ /g')"
  done
done
