#!/usr/bin/env python3
"""Times zk_ctx_set_ring for a ring of N keys with and without the per-key tables: python tools/exp_set_ring.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for kt in ('1', '0'):
    os.environ['ZKATTEST_KEYTAB'] = kt
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    eng.set_comb_bits(16)
    eng.set_params(*eng.synth_params(1), 80)
    ring = eng.synth_workload(1, N, 1)[0]
    ts = []
    for _ in range(3):
        t0 = time.time()
        eng.set_ring(ring, N)
        ts.append(time.time() - t0)
    print('ZKATTEST_KEYTAB=%s  ring of %d keys: zk_ctx_set_ring %s s' % (kt, N, ' / '.join('%.3f' % t for t in ts)))
    eng.close()
