#!/usr/bin/env python3
"""Median wall time of small verify calls (zk_verify_batch on host pointers, one lane), for same-box A/Bs of the small-call schedule: ZKATTEST_LIB picks the library.
    python tools/lat_ab.py [ring keys = 65536] [calls per size = 31]     ->  one line: B:ms pairs for B = 1 4 16 64 200 (and the prove call of B = 1)"""
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkp_ecdsa_amd as Z  # noqa: E402

nk = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 31
eng = Z.Engine(0)
eng.set_comb_bits(16)
eng.set_params(*eng.synth_params(2024), 80)
ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, nk, 256)
eng.set_ring(ring, nk)
eng.set_lanes(1)
pin = Z.PinnedBuffer(64 << 20)
out = []
for B in (1, 4, 16, 64, 200):
    eng.set_chunk(B)
    a = (msg[:32 * B], sig[:64 * B], pk[:64 * B], which[:B], seeds[:32 * B])
    tp, tv = [], []
    for k in range(reps + 2):
        dt, hout, hoff, hst = eng.prove_batch_host_raw(*a, out=pin)
        assert not any(hst)
        vdt, vok, vst = eng.verify_batch_host_raw(a[0], hout, hoff, B)
        assert sum(vok) == B
        if k >= 2:
            tp.append(dt), tv.append(vdt)
    tp.sort(), tv.sort()
    out.append('B=%d verify %.3f ms (min %.3f)%s' % (B, 1e3 * tv[len(tv) // 2], 1e3 * tv[0], ' prove %.3f ms' % (1e3 * tp[len(tp) // 2]) if B == 1 else ''))
print(os.path.basename(os.environ.get('ZKATTEST_LIB', 'main')), ' | '.join(out))
