#!/bin/bash
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
O=gpurun_out/r5_summary.txt; rm -f $O
fmt() { python3 -c "import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); print(d['tag'], 'start', d.get('probe_at_start'), [(c['proofs_per_s']) for c in d.get('calls',[])], 'end', d.get('probe_at_end'))"; }
python tools/pcie_link_watch.py 200 0.1 > gpurun_out/r5_link_watch.txt 2>&1 &
W=$!
echo "# t=$(date +%s.%N) A: back to back" >> $O
for i in 1 2 3 4 5 6; do echo "t=$(date +%s.%N)" >> $O; python tools/exp_pool_first_call.py --tag A$i --sync-calls 2 --stream 0 2>/dev/null | fmt >> $O; done
echo "# rocm-smi" >> $O
rocm-smi --showpcieclk 2>&1 | tail -12 >> $O
rocm-smi --showperflevel 2>&1 | tail -6 >> $O
echo "# try: perf level high" >> $O
rocm-smi --setperflevel high >> $O 2>&1
rocm-smi --showperflevel 2>&1 | tail -4 >> $O
for i in 1 2 3 4 5 6; do echo "t=$(date +%s.%N)" >> $O; python tools/exp_pool_first_call.py --tag H$i --sync-calls 2 --stream 0 2>/dev/null | fmt >> $O; done
rocm-smi --setperflevel auto >> $O 2>&1
kill $W 2>/dev/null
date +%s.%N >> gpurun_out/r5_link_watch.txt
cat $O; echo ----; head -60 gpurun_out/r5_link_watch.txt
