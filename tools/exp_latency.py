#!/usr/bin/env python3
"""Small-batch latency of zk_prove_batch / zk_verify_batch for one library build (ZKATTEST_LIB selects it): B = 1, 8, 64 proofs per call on a ring
of 1024 keys, 16-bit combs, median of 9 calls; per-family GPU milliseconds of the last B = 1 call.
  ZKATTEST_LIB=zkp-ecdsa_amd/build_ab/lib_x.so python tools/exp_latency.py"""
import json
import os
import sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkp_ecdsa_amd as Z

eng = Z.Engine(0)
eng.set_timing(1)
eng.set_comb_bits(16)
eng.set_params(*eng.synth_params(2024), 80)
ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, 1024, 64)
eng.set_ring(ring, 1024)
eng.set_lanes(1)
pin = Z.PinnedBuffer(64 << 20)
out = {'lib': os.environ.get('ZKATTEST_LIB', 'main')}
for B in (1, 8, 64):
    eng.set_chunk(B)
    a = (msg[:32 * B], sig[:64 * B], pk[:64 * B], which[:B], seeds[:32 * B])
    tp, tv = [], []
    for k in range(10):
        dt, hout, hoff, hst = eng.prove_batch_host_raw(*a, out=pin)
        fam_p = eng.last_timing()
        vdt, vok, vst = eng.verify_batch_host_raw(a[0], hout, hoff, B)
        fam_v = eng.last_timing()
        assert sum(vok) == B
        if k:
            tp.append(dt), tv.append(vdt)
    tp.sort(), tv.sort()
    out[str(B)] = {'prove_ms': round(1e3 * tp[4], 2), 'verify_ms': round(1e3 * tv[4], 2)}
    if B == 1:
        out['prove_gpu_ms'] = round(fam_p[0], 2)
        out['prove_families_ms'] = {k: round(v, 2) for k, v in sorted(fam_p[1].items(), key=lambda kv: -kv[1])}
        out['verify_gpu_ms'] = round(fam_v[0], 2)
        out['verify_families_ms'] = {k: round(v, 2) for k, v in sorted(fam_v[1].items(), key=lambda kv: -kv[1])}
print(json.dumps(out))
