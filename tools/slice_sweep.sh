#!/bin/bash
# prove calls of 1024..8192 proofs on page-locked host buffers: proofs per PointAdd slice (the output of a slice travels while the next one is computed)
export GPU_MAX_HW_QUEUES=8
for B in 1024 2048 4096 8192; do
  for S in 0 256 512 1024 2048; do
    [ $S -ge $B ] && continue
    echo "B=$B slice=$S: $(LAT_SLICE=$S timeout 300 python tools/lat_dist.py 65536 12 $B 2>&1 | tail -2 | head -1 | sed 's/  */ /g')"
  done
done
