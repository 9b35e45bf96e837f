#!/bin/bash
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
rm -f gpurun_out/r3_*.txt
for i in 1 2 3 4 5; do
  ZK_IO_DEBUG=1 python tools/exp_pool_first_call.py --tag io$i --sync-calls 2 --stream 0 > gpurun_out/r3_io$i.out 2> gpurun_out/r3_io$i.err
  grep '^{' gpurun_out/r3_io$i.out | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['tag'], [(c['proofs_per_s'],c['d2h_gbps']) for c in d['calls']])" >> gpurun_out/r3_summary.txt
  grep '^io:' gpurun_out/r3_io$i.err | awk '{print $5, $(NF-1)}' | tr -d '(' | sort | awk '{n[$1]++; s[$1]+=$2; if(min[$1]==""||$2<min[$1])min[$1]=$2; if($2>max[$1])max[$1]=$2} END{for(k in n) printf "   lane %s: %d copies, mean %.1f min %.1f max %.1f GB/s\n", k, n[k], s[k]/n[k], min[k], max[k]}' >> gpurun_out/r3_summary.txt
done
for i in 1 2 3 4; do
  ZKATTEST_COPY_STREAMS=1 python tools/exp_pool_first_call.py --tag onecopystream$i --sync-calls 3 --stream 0 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['tag'], [(c['proofs_per_s'],c['d2h_gbps']) for c in d['calls']])" >> gpurun_out/r3_summary.txt
done
for i in 1 2 3; do
  HSA_ENABLE_SDMA=0 python tools/exp_pool_first_call.py --tag nosdma$i --sync-calls 3 --stream 0 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['tag'], [(c['proofs_per_s'],c['d2h_gbps']) for c in d['calls']])" >> gpurun_out/r3_summary.txt
done
head -c 6000 gpurun_out/r3_io1.err > gpurun_out/r3_io1_head.txt
cat gpurun_out/r3_summary.txt
