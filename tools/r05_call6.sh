#!/bin/bash
# Round 5, GPU call 6: the whole GPU tier, then the default bench line.
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05/t6_all.log 2>&1; echo "gpu tier rc=$?"; tail -6 gpurun_out/r05/t6_all.log
timeout 900 python bench.py > gpurun_out/r05/b6.json 2> gpurun_out/r05/b6.err; echo "bench rc=$?"; tail -2 gpurun_out/r05/b6.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/b6.json').read().strip().splitlines()[-1])
print('prove', d['value'], d['ms_per_step'], 'latency b1', d.get('latency_ms_b1'), d.get('verify_latency_ms_b1'))
v=d['verify']; print('verify', v['value'], v['ms_per_step'], json.dumps(v['gpu_ms_by_family_per_step']))
print('cpu', json.dumps(d['cpu_baseline'])[:600])
print('roofline', d['roofline']['frac'], json.dumps(v.get('roofline'))[:500])
print('pcie', d.get('value_pcie_inclusive'), d.get('verify_pcie_inclusive'), d.get('value_pcie_inclusive_steady_packed'), d.get('verify_pcie_inclusive_steady_packed'))
print(json.dumps(d.get('latency'))[:1500])
PY
