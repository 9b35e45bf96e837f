#!/bin/bash
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gpu_small_batches.py tests/test_gpu_verify.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2 3; do
  echo "base (HEAD~):  $(ZKATTEST_LIB=$PWD/zkp-ecdsa_amd/build_ab/lib_base.so timeout 300 python tools/lat_dist.py 65536 200 1 2>&1 | tail -1 | sed 's/  */ /g')"
  echo "tree:          $(timeout 300 python tools/lat_dist.py 65536 200 1 2>&1 | tail -1 | sed 's/  */ /g')"
done
bash tools/tl1.sh
