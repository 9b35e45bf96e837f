#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (21)): a blocking verify call of a few proofs whose auxiliary streams are handed their stage-2 kernels by the host when stage 1
# is over (VerifyJob::host_release) against the same library with the streams waiting for stage 1's event (ZKATTEST_NO_HOST_RELEASE=1): same box, interleaved;
# tools/lat_dist.py (N calls on host pointers).  The verifier's parity / exception-order tests first.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06ab
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mutants.py tests/test_gpu_small_batches.py tests/test_gpu_verify.py tests/test_wire_packed.py -q -m gpu -x > $O/release_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/release_tests.log
for rep in 1 2 3; do
  echo "streams wait:  $(ZKATTEST_NO_HOST_RELEASE=1 timeout 300 python tools/lat_dist.py 65536 200 1 2>&1 | tail -1 | sed 's/  */ /g')"
  echo "host releases: $(timeout 300 python tools/lat_dist.py 65536 200 1 2>&1 | tail -1 | sed 's/  */ /g')"
done
for B in 4 16 64 200; do
  echo "streams wait:  $(ZKATTEST_NO_HOST_RELEASE=1 timeout 300 python tools/lat_dist.py 65536 60 $B 2>&1 | tail -1 | sed 's/  */ /g')"
  echo "host releases: $(timeout 300 python tools/lat_dist.py 65536 60 $B 2>&1 | tail -1 | sed 's/  */ /g')"
done
bash tools/tl1.sh
