#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export ZKATTEST_HOST_ALLOC_PROBE=0
mkdir -p gpurun_out
O=gpurun_out/r04_runtime_ab.txt; rm -f $O
fmt() { python3 -c "import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print(d['tag'], [(c['proofs_per_s']) for c in d.get('calls',[])], [os.path.basename(x) for x in d.get('hip_runtime',[])] if False else d.get('hip_runtime'))"; }
echo "# allocation probing OFF everywhere.  system HIP runtime (/opt/rocm), 6 fresh processes back to back" >> $O
for i in 1 2 3 4 5 6; do python tools/exp_pool_first_call.py --tag sys$i --sync-calls 2 --stream 0 --probe 0 2>/dev/null | fmt >> $O; done
echo "# torch imported first (the wheel's bundled HIP runtime), 6 fresh processes back to back" >> $O
for i in 1 2 3 4 5 6; do python tools/exp_pool_first_call.py --tag torch$i --sync-calls 2 --stream 0 --probe 0 --torch-first 1 2>/dev/null | fmt >> $O; done
echo "# system runtime, HSA_ENABLE_SDMA_RECOMMENDED_ENG=0" >> $O
for i in 1 2 3 4 5 6; do HSA_ENABLE_SDMA_RECOMMENDED_ENG=0 python tools/exp_pool_first_call.py --tag rec0_$i --sync-calls 2 --stream 0 --probe 0 2>/dev/null | fmt >> $O; done
echo "# system runtime, HSA_ENABLE_SDMA_RECOMMENDED_ENG=1" >> $O
for i in 1 2 3 4 5 6; do HSA_ENABLE_SDMA_RECOMMENDED_ENG=1 python tools/exp_pool_first_call.py --tag rec1_$i --sync-calls 2 --stream 0 --probe 0 2>/dev/null | fmt >> $O; done
cat $O
