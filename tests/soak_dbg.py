#!/usr/bin/env python3
"""The soak test of tests/test_gpu_small_batches.py as a loop that reports every mismatch with the settings it happened under (chunk, lanes, bucket threshold,
call size) instead of stopping at the first: python tests/soak_dbg.py [rounds = 5] [verify only = 0]"""
import hashlib
import os
import random
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
sys.path.insert(0, os.path.join(R, 'oracle'))   # test infrastructure: the oracle is the checker here, as in the tests
import coracle as CO  # noqa: E402
import zkp_ecdsa_amd as Z  # noqa: E402
from test_gpu_small_batches import _forge  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
vonly = int(sys.argv[2]) if len(sys.argv) > 2 else 0
S, nkeys, NP = 6300, 64, 24
eng = Z.Engine(0)
nh, tg, th = eng.synth_params(S)
eng.set_comb_bits(16)
eng.set_params(nh, tg, th, 80)
ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, NP)
eng.set_ring(ring, nkeys)
octx = CO.OracleCtx(nh, tg, th, 80)
octx.set_ring(ring, nkeys)
honest, st = octx.prove_batch(msg, sig, pk, which, seeds=seeds, nthreads=16)
pool = []
for i, p in enumerate(honest):
    m = msg[32 * i:32 * i + 32]
    pool.append((m, p, hashlib.sha256(b'soak-h' + bytes([i])).digest()))
    pool.append((m, _forge(p, i % 5), hashlib.sha256(b'soak-f' + bytes([i])).digest()))
ok, vst = octx.verify_batch(b''.join(e[0] for e in pool), [e[1] for e in pool], nthreads=16, vseeds=b''.join(e[2] for e in pool))
cut = lambda buf, w, ids: b''.join(buf[w * i:w * i + w] for i in ids)
bad = 0
for rd in range(rounds):
    rnd = random.Random(4 + rd)
    cfg = None
    for call in range(300):
        if call % 25 == 0:
            cfg = (rnd.choice((4096, 16, 5)), rnd.choice((1, 2, 3)), rnd.choice((256, 0, 8)))
            eng.set_chunk(cfg[0]), eng.set_lanes(cfg[1]), eng.set_batch_verify(cfg[2])
        if call % 3 == 0 and not vonly:
            ids = [rnd.randrange(NP) for _ in range(rnd.randint(1, 6))]
            got, gst = eng.prove_batch(cut(msg, 32, ids), cut(sig, 64, ids), cut(pk, 64, ids), [which[i] for i in ids], seeds=cut(seeds, 32, ids))
            if gst != [0] * len(ids) or got != [honest[i] for i in ids]:
                bad += 1
                print('round %d call %d PROVE mismatch cfg %s n %d' % (rd, call, cfg, len(ids)), flush=True)
        else:
            ids = [rnd.randrange(len(pool)) for _ in range(rnd.randint(1, 40))]
            g = eng.verify_batch(b''.join(pool[i][0] for i in ids), [pool[i][1] for i in ids], vseeds=b''.join(pool[i][2] for i in ids))
            w = ([ok[i] for i in ids], [vst[i] for i in ids])
            if g != w:
                bad += 1
                d = [(k, ids[k], g[0][k], w[0][k], g[1][k], w[1][k]) for k in range(len(ids)) if (g[0][k], g[1][k]) != (w[0][k], w[1][k])]
                print('round %d call %d VERIFY mismatch cfg (chunk, lanes, bmin) %s n %d: (pos, pool id, ok got/want, status got/want) %s' % (rd, call, cfg, len(ids), d), flush=True)
print('mismatches: %d' % bad)
