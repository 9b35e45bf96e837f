#!/bin/bash
# One kernel timeline of a one-proof prove + verify call pair (tools/b1_timeline.py under rocprofv3 --kernel-trace) -> gpurun_out/tl_dbg.txt, and the lines of the
# verifier's PointAdd-challenge kernels and their neighbours.
export GPU_MAX_HW_QUEUES=8
R=$PWD
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/b1tl && timeout 300 rocprofv3 --kernel-trace -d /tmp/b1tl -o r -- python $R/tools/b1_timeline.py run > /dev/null 2>&1; python $R/tools/b1_timeline.py parse $(find /tmp/b1tl -name 'r_results.db' | head -1) > $R/gpurun_out/tl_dbg.txt 2>&1)
grep -E "^== " gpurun_out/tl_dbg.txt
awk '/last verify call/,/-- totals/' gpurun_out/tl_dbg.txt | grep -E "k_v_padd_msg|k_exph_sched|k_exph_rounds2<true>|normalize_each|slot_terms"
