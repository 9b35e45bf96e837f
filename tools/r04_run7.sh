#!/bin/bash
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
O=gpurun_out/r04_pool_first_call_fixed.txt; rm -f $O
fmt() { python3 -c "import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print(d['tag'], d['alloc_s'], [(c['proofs_per_s']) for c in d.get('calls',[])])"; }
echo "# allocation probing ON (default), 10 fresh processes back to back" >> $O
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do ZK_ALLOC_DEBUG=1 python tools/exp_pool_first_call.py --tag on$i --sync-calls 2 --stream 0 --probe 0 2>gpurun_out/r7_err.txt | fmt >> $O; grep '^alloc:' gpurun_out/r7_err.txt | sed 's/^/      /' >> $O; done
echo "# allocation probing OFF, 6 fresh processes back to back" >> $O
for i in 1 2 3 4 5 6; do ZKATTEST_HOST_ALLOC_PROBE=0 python tools/exp_pool_first_call.py --tag off$i --sync-calls 2 --stream 0 --probe 0 2>/dev/null | fmt >> $O; done
cat $O
