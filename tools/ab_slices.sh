#!/bin/bash
# Round 6 A/B (profiles/r06_ab_variants.txt (22)): prove calls of ONE chunk on page-locked host buffers with the automatic slice size (512 up to 2048 proofs, 1024 up to
# 8192) against the old default (LAT_SLICE=4096: one slice below 4096 proofs), same library, same box, interleaved; tools/lat_dist.py, 16 calls each.
export GPU_MAX_HW_QUEUES=8
timeout 900 python -m pytest tests/test_gpu_scale.py -q -m gpu -x -k "tapered or bench_configuration" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_stream.py tests/test_wire_packed.py -q -m gpu -x 2>&1 | tail -2
for B in 600 1024 2048 4096 8192; do
  for rep in 1 2; do
    echo "4096 per slice: $(LAT_SLICE=4096 timeout 300 python tools/lat_dist.py 65536 16 $B 2>&1 | tail -2 | head -1 | sed 's/  */ /g')"
    echo "automatic:      $(timeout 300 python tools/lat_dist.py 65536 16 $B 2>&1 | tail -2 | head -1 | sed 's/  */ /g')"
  done
done
