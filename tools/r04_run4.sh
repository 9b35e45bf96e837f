#!/bin/bash
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
O=gpurun_out/r4_summary.txt; rm -f $O
fmt() { python3 -c "import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print(d['tag'], 'start', d.get('probe_at_start'), [(c['proofs_per_s']) for c in d.get('calls',[])], 'end', d.get('probe_at_end'))"; }
echo "# A: back to back, abrupt exit" >> $O
for i in 1 2 3 4 5 6; do python tools/exp_pool_first_call.py --tag A$i --sync-calls 2 --stream 0 2>/dev/null | fmt >> $O; done
echo "# B: 3 s pause between processes" >> $O
for i in 1 2 3 4 5 6; do sleep 3; python tools/exp_pool_first_call.py --tag B$i --sync-calls 2 --stream 0 2>/dev/null | fmt >> $O; done
echo "# C: clean exit (pool destroyed), back to back" >> $O
for i in 1 2 3 4 5 6; do python tools/exp_pool_first_call.py --tag C$i --sync-calls 2 --stream 0 --clean-exit 1 2>/dev/null | fmt >> $O; done
echo "# D: while another process holds a context" >> $O
python tools/exp_pool_first_call.py --tag holder --hold 45 2>/dev/null | fmt >> $O &
sleep 6
for i in 1 2 3; do python tools/exp_pool_first_call.py --tag D$i --sync-calls 2 --stream 0 2>/dev/null | fmt >> $O; done
wait
cat $O
