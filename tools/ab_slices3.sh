#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (22), last part): the slice rule of the tree (one chunk of <= 8192 proofs, or chunks of <= 4096 in a call of <= 32768) against
# the one-chunk rule (build_ab/lib_base.so), same box: bench.py's one host-pointer call of 65 536 proofs, the pool's, and calls of 6000..32768 proofs.
export GPU_MAX_HW_QUEUES=8
F="--steps 1 --warmup 1 --verify-steps 0 --json-sample 0 --latency 0 --no-cpu-baseline --roofline-steps 0 --host-io-stream 0 --host-io-packed 0"
for rep in 1 2; do
  for v in base tree; do
    lib=$PWD/zkp-ecdsa_amd/build_ab/lib_base.so; [ $v = tree ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
    ZKATTEST_LIB=$lib python bench.py $F 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_io']['pinned']; print('$v', '65536: one call %.1f k proofs/s (%.4f s)' % (h['proofs_per_s']/1e3, h['prove_s']))"
  done
done
for v in base tree; do
  lib=$PWD/zkp-ecdsa_amd/build_ab/lib_base.so; [ $v = tree ] && lib=$PWD/zkp-ecdsa_amd/lib/libzkattest_hip.so
  ZKATTEST_LIB=$lib python bench.py --pool --gpus 1 --steps 2 --warmup 1 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v pool', d['value'])"
done
for cfg in "8192 4096 2" "16384 4096 2" "32768 4096 2" "32768 8192 2" "6000 2048 2" "4096 4096 1"; do
  set -- $cfg
  echo "B=$1 chunk=$2 lanes=$3 base: $(ZKATTEST_LIB=$PWD/zkp-ecdsa_amd/build_ab/lib_base.so LAT_CHUNK=$2 LAT_LANES=$3 timeout 300 python tools/lat_dist.py 65536 8 $1 2>&1 | tail -2 | head -1 | sed 's/  */ /g')"
  echo "B=$1 chunk=$2 lanes=$3 tree: $(LAT_CHUNK=$2 LAT_LANES=$3 timeout 300 python tools/lat_dist.py 65536 8 $1 2>&1 | tail -2 | head -1 | sed 's/  */ /g')"
done
timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_prove.py tests/test_gpu_stream.py tests/test_wire_packed.py -q -m gpu -x 2>&1 | tail -2
