#!/bin/bash
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
O=gpurun_out/r6_link_state.jsonl; rm -f $O
for i in 1 2 3 4 5 6; do python tools/exp_link_state.py idle2,probe4g,tables16,idle2 s$i 2>gpurun_out/r6_err.txt | grep '^{' >> $O; done
for i in 1 2 3; do taskset -c 0-7 python tools/exp_link_state.py probe4g,tables16 cpu0-7_$i 2>/dev/null | grep '^{' >> $O; done
lscpu | grep -i "numa\|socket" > gpurun_out/r6_lscpu.txt; nproc >> gpurun_out/r6_lscpu.txt; cat /sys/fs/cgroup/cpuset.cpus.effective >> gpurun_out/r6_lscpu.txt 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r6_link_state.jsonl'):
    d=json.loads(l)
    print(d['tag'], d.get('bus'), 'gpu node', d.get('gpu_numa_node'))
    for s in d['steps']:
        print('   %-9s t=%5.1f cpu/node=%s  default=%s node0=%s node1=%s  %s' % (s['after'], s['t'], s['cpu_node'], s['d2h_h2d']['default'], s['d2h_h2d']['node0'], s['d2h_h2d']['node1'], s['dev']))
PY
cat gpurun_out/r6_lscpu.txt; tail -3 gpurun_out/r6_err.txt
