"""-m gpu: the paths small calls take since round 4: one term per lane in the verifier's per-proof sums, the independent phases of
a one-chunk call on auxiliary streams, the prover's membership phase beside its PointAdd phase.  The bytes, verdicts and exact status codes must be those
of the oracle, and of the engine's own chunked paths, whatever path the sizes select: B = 1 (the reference's only shape, zkpAttestList.ts:104-190), a
few proofs, one chunk below and above the bucket pass's threshold, several chunks, the bucket pass switched off."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu


def _vseeds(n, tag):
    return b''.join(hashlib.sha256(tag + i.to_bytes(4, 'big')).digest() for i in range(n))


def _forge(p, kind):
    b = bytearray(p)
    if kind == 0:
        b[-9] ^= 0x10                       # a membership response: always caught
    elif kind == 1:
        b[96 + 40] ^= 0x01                  # comS1 (Clambda): off the curve -> deserialisation error
    else:
        b[304 + 336 * (kind % 7) + 100] ^= 0x04   # inside a repetition: caught when the verifier samples it
    return bytes(b)


@pytest.mark.parametrize('B', [1, 2, 5])
def test_a_few_proofs_per_call_bytes_and_verdicts_are_the_oracles(B):
    import coracle as CO
    import zkp_ecdsa_amd as Z
    S, nkeys = 6100 + B, 16
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(S)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, B)
    eng.set_ring(ring, nkeys)
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, nkeys)
    gp, gst = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    op, ost = octx.prove_batch(msg, sig, pk, which, seeds=seeds, nthreads=B)
    assert gst == ost == [0] * B and gp == op
    for tag in (b's0', b's1'):
        vs = _vseeds(B, tag)
        assert eng.verify_batch(msg, gp, vseeds=vs) == octx.verify_batch(msg, gp, nthreads=B, vseeds=vs) == ([1] * B, [0] * B)
        for kind in (0, 1, 2, 3):
            bad = [(_forge(p, kind) if i == B - 1 else p) for i, p in enumerate(gp)]
            g = eng.verify_batch(msg, bad, vseeds=vs)
            assert g == octx.verify_batch(msg, bad, nthreads=B, vseeds=vs), (tag, kind, g)
            assert g[0][:B - 1] == [1] * (B - 1)
    eng.close()


def test_every_size_class_of_the_verifier_gives_the_same_verdicts():
    import coracle as CO
    import zkp_ecdsa_amd as Z
    S, nkeys, B = 6200, 1024, 300
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(S)
    eng.set_comb_bits(16)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, B)
    eng.set_ring(ring, nkeys)
    proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    forged = {3: 0, 17: 2, 44: 1, 130: 3, 255: 0, 256: 4, 299: 0}
    plist = [(_forge(p, forged[i]) if i in forged else p) for i, p in enumerate(proofs)]
    vs = _vseeds(B, b'sz')
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, nkeys)
    want = octx.verify_batch(msg, plist, nthreads=16, vseeds=vs)
    assert want[0].count(0) >= 4 and 10 in want[1]
    # (chunk, zk_ctx_set_batch_verify): one chunk through the bucket pass with its failing groups re-checked one term per lane; 256 + 44; three chunks of
    # per-proof sums; a chunk of one proof at the end; the bucket pass off: one chunk of 300 (four lanes per slot), 150 + 150 (one term per lane)
    for chunk, bmin in ((300, 256), (256, 256), (100, 256), (299, 256), (300, 0), (150, 0), (64, 64)):
        eng.set_chunk(chunk)
        eng.set_batch_verify(bmin)
        got = eng.verify_batch(msg, plist, vseeds=vs)
        assert got == want, (chunk, bmin, [i for i in range(B) if (got[0][i], got[1][i]) != (want[0][i], want[1][i])])
    eng.close()


def test_soak_random_small_calls_against_precomputed_oracle_results():
    """300 calls of random size and composition on ONE context, provers and verifiers interleaved, chunk size / lanes / bucket threshold changed in between:
    the side streams of consecutive calls reuse the same events, accumulators and workspaces, so a missing join shows up as a wrong byte or verdict here.
    A proof's verdict depends on its bytes and its verifier seed only (the engine's private randomisers change with its position in the batch, the outcome
    must not): the oracle judges every (proof, seed) pair of the pool once."""
    import random
    import coracle as CO
    import zkp_ecdsa_amd as Z
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
    assert st == [0] * NP
    pool = []   # (message, proof bytes, verifier seed)
    for i, p in enumerate(honest):
        m = msg[32 * i:32 * i + 32]
        pool.append((m, p, hashlib.sha256(b'soak-h' + bytes([i])).digest()))
        pool.append((m, _forge(p, i % 5), hashlib.sha256(b'soak-f' + bytes([i])).digest()))
    ok, vst = octx.verify_batch(b''.join(e[0] for e in pool), [e[1] for e in pool], nthreads=16, vseeds=b''.join(e[2] for e in pool))
    assert ok[0::2] == [1] * NP and ok[1::2].count(0) >= NP // 2
    rnd = random.Random(4)
    cut = lambda buf, w, ids: b''.join(buf[w * i:w * i + w] for i in ids)
    for call in range(300):
        if call % 25 == 0:
            eng.set_chunk(rnd.choice((4096, 16, 5)))
            eng.set_lanes(rnd.choice((1, 2, 3)))
            eng.set_batch_verify(rnd.choice((256, 0, 8)))
        if call % 3 == 0:
            ids = [rnd.randrange(NP) for _ in range(rnd.randint(1, 6))]
            got, gst = eng.prove_batch(cut(msg, 32, ids), cut(sig, 64, ids), cut(pk, 64, ids), [which[i] for i in ids], seeds=cut(seeds, 32, ids))
            assert gst == [0] * len(ids) and got == [honest[i] for i in ids], call
        else:
            ids = [rnd.randrange(len(pool)) for _ in range(rnd.randint(1, 40))]
            g = eng.verify_batch(b''.join(pool[i][0] for i in ids), [pool[i][1] for i in ids], vseeds=b''.join(pool[i][2] for i in ids))
            assert g == ([ok[i] for i in ids], [vst[i] for i in ids]), (call, ids)
    eng.close()
