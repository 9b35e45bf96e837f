import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['workload'][-26:], 'set_params_s', d.get('set_params_s'), d['gpu_ms_by_family_per_step'], 'failed', d['failed_proofs'], 'verify', (d.get('verify') or {}).get('value'), (d.get('verify') or {}).get('accepted'))
