#!/bin/bash
# One proof / one verification per call (tools/lat_dist.py) under runtime environment knobs; ZKATTEST_LIB picks the library.
run() { echo "$1: $(env $1 timeout 120 python tools/lat_dist.py 65536 100 1 2>&1 | tail -2 | tr '\n' ' ' | sed 's/  */ /g')"; }
for rep in 1 2; do
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=12
run GPU_MAX_HW_QUEUES=16
run GPU_MAX_HW_QUEUES=24
run HSA_ENABLE_INTERRUPT=0
run HIP_FORCE_DEV_KERNARG=1
run ROC_ACTIVE_WAIT_TIMEOUT=2000
run HSA_ENABLE_SDMA=0
run "HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1"
done
