#!/bin/bash
# calls of 2048..8192 proofs on page-locked host buffers as ONE chunk (what bench.py's latency table does) or as several chunks on one or two lanes
export GPU_MAX_HW_QUEUES=8
for B in 2048 4096 8192; do
  for cfg in "$B 1" "$((B/2)) 1" "$((B/2)) 2" "$((B/4)) 2" "$((B/4)) 1"; do
    set -- $cfg
    echo "B=$B chunk=$1 lanes=$2: $(LAT_CHUNK=$1 LAT_LANES=$2 timeout 300 python tools/lat_dist.py 65536 12 $B 2>&1 | tail -2 | tr '\n' ' ' | sed 's/  */ /g' | sed 's/min [0-9.]* p25 [0-9.]* //g; s/p75.*B=/ B=/; s/p75.*//')"
  done
done
