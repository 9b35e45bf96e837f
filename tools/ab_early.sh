#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (27)): a small one-chunk prove call puts the second RNG prepass and the membership phase up to its challenge on the device BEFORE the
# host waits for the scan's totals, against the library before (build_ab/lib_base.so), same box, interleaved three times; tools/lat_dist.py.
export GPU_MAX_HW_QUEUES=8
timeout 1200 python -m pytest tests/test_gpu_small_batches.py tests/test_gpu_prove.py tests/test_gpu_stream.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2 3; do
  echo "before: $(ZKATTEST_LIB=$PWD/zkp-ecdsa_amd/build_ab/lib_base.so timeout 300 python tools/lat_dist.py 65536 200 1 2>&1 | tail -2 | head -1 | sed 's/  */ /g')"
  echo "tree:   $(timeout 300 python tools/lat_dist.py 65536 200 1 2>&1 | tail -2 | head -1 | sed 's/  */ /g')"
done
for B in 4 16 64 256 1024; do
  echo "before: $(ZKATTEST_LIB=$PWD/zkp-ecdsa_amd/build_ab/lib_base.so timeout 300 python tools/lat_dist.py 65536 60 $B 2>&1 | tail -2 | head -1 | sed 's/  */ /g')"
  echo "tree:   $(timeout 300 python tools/lat_dist.py 65536 60 $B 2>&1 | tail -2 | head -1 | sed 's/  */ /g')"
done
for i in 1 2 3; do timeout 120 python tools/lat_modes.py 2>&1 | tail -1 | cut -c1-60; done
