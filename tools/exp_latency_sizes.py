#!/usr/bin/env python3
"""Latency of zk_verify_batch (and zk_prove_batch) against the batch size for one library build (ZKATTEST_LIB) and one threshold of the batched Tom-256
check (zk_ctx_set_batch_verify): where the per-proof sums of a small batch stop paying.
  ZKATTEST_LIB=... python tools/exp_latency_sizes.py <batch_verify_min> <B> [<B> ...]"""
import json
import os
import sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkp_ecdsa_amd as Z

bmin = int(sys.argv[1])
sizes = [int(a) for a in sys.argv[2:]]
nkeys = max(1024, max(sizes))   # (the synthetic workload plants one key per proof: no more proofs than keys)
eng = Z.Engine(0)
eng.set_comb_bits(16)
eng.set_params(*eng.synth_params(2024), 80)
ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, nkeys, max(sizes))
eng.set_ring(ring, nkeys)
lanes = int(os.environ.get('ZK_LAT_LANES', '1'))   # ZK_LAT_LANES=3: the call split into that many chunks on as many lanes
eng.set_lanes(lanes)
eng.set_batch_verify(bmin)
pin = Z.PinnedBuffer(max(sizes) * 180000 + (1 << 20))
out = {'lib': os.environ.get('ZKATTEST_LIB', 'main'), 'batch_verify_min': bmin, 'lanes': lanes}
for B in sizes:
    eng.set_chunk((B + lanes - 1) // lanes)
    a = (msg[:32 * B], sig[:64 * B], pk[:64 * B], which[:B], seeds[:32 * B])
    tp, tv = [], []
    for k in range(6):
        dt, hout, hoff, hst = eng.prove_batch_host_raw(*a, out=pin)
        vdt, vok, vst = eng.verify_batch_host_raw(a[0], hout, hoff, B)
        assert sum(vok) == B
        if k:
            tp.append(dt), tv.append(vdt)
    tp.sort(), tv.sort()
    out[str(B)] = {'prove_ms': round(1e3 * tp[2], 2), 'verify_ms': round(1e3 * tv[2], 2)}
print(json.dumps(out))
