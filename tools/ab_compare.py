#!/usr/bin/env python3
"""Compare bench.py JSON lines of library variants (tools/ab_variants.sh): per-family GPU ms of the serial pass."""
import json, sys, glob, os
files = sys.argv[1:] or sorted(glob.glob('gpurun_out/ab/*.json'))
rows = {}
names = []
for f in files:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e)
        continue
    n = os.path.basename(f)[:-5]
    names.append(n)
    rows.setdefault('PROVE proofs/s', {})[n] = d['value']
    rows.setdefault('PROVE ms/step', {})[n] = d['ms_per_step']
    for k, v in d['gpu_ms_by_family_per_step'].items():
        rows.setdefault('p:' + k, {})[n] = v
    if d.get('verify'):
        rows.setdefault('VERIFY /s', {})[n] = d['verify']['value']
        rows.setdefault('VERIFY ms', {})[n] = d['verify']['ms_per_step']
        rows.setdefault('VERIFY accepted', {})[n] = d['verify']['accepted']
        for k, v in d['verify']['gpu_ms_by_family_per_step'].items():
            rows.setdefault('v:' + k, {})[n] = v
    rows.setdefault('failed_proofs', {})[n] = d.get('failed_proofs')
print('%-24s' % '' + ''.join('%12s' % n for n in names))
for k, r in rows.items():
    print('%-24s' % k + ''.join('%12s' % (r.get(n, '')) for n in names))
