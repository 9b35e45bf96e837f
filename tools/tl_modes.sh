#!/bin/bash
# the shortest and the longest one-proof prove call of ONE process, kernel by kernel (tools/b1_timeline.py modes) -> gpurun_out/tl_modes.txt
export GPU_MAX_HW_QUEUES=8
R=$PWD
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/b1tl && B1_RING=65536 B1_CALLS=40 timeout 300 rocprofv3 --kernel-trace -d /tmp/b1tl -o r -- python $R/tools/b1_timeline.py run > /tmp/b1run.log 2>&1; python $R/tools/b1_timeline.py modes $(find /tmp/b1tl -name 'r_results.db' | head -1) > $R/gpurun_out/tl_modes.txt 2>&1)
grep "prove" /tmp/b1run.log | awk '{print $4}' | sort -n | tr '\n' ' '; echo
head -70 gpurun_out/tl_modes.txt
