#!/usr/bin/env python3
"""Which kernel family is longer in a process whose one-proof prove calls are 'slow' (1.38-1.41 ms instead of 1.30-1.32)?  First 120 untimed-family calls
(median wall), then 40 calls with per-family events on (zk_ctx_set_timing 1): the median GPU ms of every family.  One line per process; run it in several."""
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkp_ecdsa_amd as Z  # noqa: E402

eng = Z.Engine(0)
eng.set_comb_bits(16)
eng.set_params(*eng.synth_params(2024), 80)
ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, 65536, 4)
eng.set_ring(ring, 65536)
eng.set_lanes(1)
eng.set_chunk(1)
pin = Z.PinnedBuffer(8 << 20)
a = (msg[:32], sig[:64], pk[:64], which[:1], seeds[:32])
w = []
for k in range(125):
    dt, hout, hoff, hst = eng.prove_batch_host_raw(*a, out=pin)
    if k >= 5:
        w.append(1e3 * dt)
w.sort()
eng.set_timing(1)
fam = {}
for k in range(40):
    eng.prove_batch_host_raw(*a, out=pin)
    for n, v in eng.last_timing()[1].items():
        fam.setdefault(n, []).append(v)
out = []
for n, v in sorted(fam.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2]):
    v.sort()
    out.append('%s %.3f' % (n, v[len(v) // 2]))
print('one proof %.3f ms (p25 %.3f p75 %.3f) | %s' % (w[len(w) // 2], w[len(w) // 4], w[3 * len(w) // 4], '  '.join(out[:12])))
