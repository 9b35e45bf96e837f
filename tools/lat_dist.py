#!/usr/bin/env python3
"""Distribution of the wall time of small calls (zk_prove_batch / zk_verify_batch on host pointers, one lane): min, quartiles, 90th percentile, max over N calls.
ZKATTEST_LIB picks the library.      python tools/lat_dist.py [ring keys = 65536] [calls = 200] [B = 1]"""
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkp_ecdsa_amd as Z  # noqa: E402

nk = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
eng = Z.Engine(0)
eng.set_comb_bits(16)
eng.set_params(*eng.synth_params(2024), 80)
ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, nk, max(256, B))
eng.set_ring(ring, nk)
eng.set_lanes(int(os.environ.get('LAT_LANES', '1')))
eng.set_chunk(int(os.environ.get('LAT_CHUNK', B)))   # LAT_CHUNK / LAT_LANES: the call as several chunks on several lanes
if os.environ.get('LAT_SLICE'):
    eng.set_slice(int(os.environ['LAT_SLICE']))   # proofs per PointAdd slice of the prover (default: 4096 with a page-locked sink)
pin = Z.PinnedBuffer(max(64 << 20, B * 180000))
a = (msg[:32 * B], sig[:64 * B], pk[:64 * B], which[:B], seeds[:32 * B])
tp, tv = [], []
for k in range(reps + 5):
    dt, hout, hoff, hst = eng.prove_batch_host_raw(*a, out=pin)
    assert not any(hst)
    vdt, vok, vst = eng.verify_batch_host_raw(a[0], hout, hoff, B)
    assert sum(vok) == B
    if k >= 5:
        tp.append(1e3 * dt), tv.append(1e3 * vdt)


def q(v):
    v = sorted(v)
    n = len(v)
    return 'min %.3f  p25 %.3f  p50 %.3f  p75 %.3f  p90 %.3f  max %.3f' % (v[0], v[n // 4], v[n // 2], v[3 * n // 4], v[9 * n // 10], v[-1])


print('%-22s B=%d  prove  %s' % (os.path.basename(os.environ.get('ZKATTEST_LIB', 'main')), B, q(tp)))
print('%-22s B=%d  verify %s' % ('', B, q(tv)))
