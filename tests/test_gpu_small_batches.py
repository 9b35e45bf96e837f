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


def test_one_chunk_through_the_bucket_pass_with_several_failing_ranges():
    """A small one-chunk call that takes the bucket pass (threshold 8) and fails it in several separate runs of groups: every run's per-proof sums reuse the
    same per-term accumulators, so a run has to start behind the one before it.  (Round 6's one-fork-per-chunk change got that wrong for the second run; the
    soak test above found it in 1 of 4 runs -- this one places the failing groups so that the regions overlap and fails on that version every time.)"""
    import coracle as CO
    import zkp_ecdsa_amd as Z
    S, nkeys, B = 6400, 64, 40
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(S)
    eng.set_comb_bits(16)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, B)
    eng.set_ring(ring, nkeys)
    proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    # groups of 5 proofs (8 groups): the runs {0, 1}, {4}, {7} fail -- a LONGER run first, so that the next run's membership accumulators (they lie behind the
    # run's own slot accumulators in the shared array) land inside the region the first run's slot sums are still using
    plist = [(_forge(p, 0) if i in (2, 8, 22, 37, 38) else p) for i, p in enumerate(proofs)]
    vs = _vseeds(B, b'rg')
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, nkeys)
    want = octx.verify_batch(msg, plist, nthreads=16, vseeds=vs)
    assert want[0].count(0) == 5
    eng.set_chunk(4096)
    eng.set_batch_verify(8)
    for lanes in (1, 2):
        eng.set_lanes(lanes)
        for rep in range(12):
            c0 = eng.test_counter(0)
            got = eng.verify_batch(msg, plist, vseeds=vs)
            assert got == want, (lanes, rep, [i for i in range(B) if (got[0][i], got[1][i]) != (want[0][i], want[1][i])])
            assert eng.test_counter(0) - c0 == 20   # groups 0, 1, 4, 7 of five proofs each were re-checked proof by proof, the other four passed the bucket pass
    eng.close()


def test_per_family_events_only_when_asked_for_in_small_calls():
    """zk_ctx_set_timing (include/zkattest.h): AUTO records the per-family events only in calls of more than 8 192 proofs, ON in every blocking call, OFF never;
    the bytes and verdicts do not depend on it."""
    import zkp_ecdsa_amd as Z
    S, nkeys, B = 6500, 64, 3
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(S)
    eng.set_comb_bits(16)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, B)
    eng.set_ring(ring, nkeys)
    p0, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B and eng.last_timing() == (0.0, {})
    vs = _vseeds(B, b'tm')
    assert eng.verify_batch(msg, p0, vseeds=vs) == ([1] * B, [0] * B) and eng.last_timing() == (0.0, {})
    eng.set_timing(1)
    p1, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    tot, fam = eng.last_timing()
    assert p1 == p0 and tot > 0 and 'tom_commit' in fam and 'hash' in fam
    assert eng.verify_batch(msg, p0, vseeds=vs) == ([1] * B, [0] * B)
    tot, fam = eng.last_timing()
    assert tot > 0 and 'v_hash' in fam and eng.last_wall_ms() > 0
    eng.set_timing(0)
    p2, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert p2 == p0 and eng.last_timing() == (0.0, {})
    with pytest.raises(Exception):
        eng.set_timing(3)
    eng.close()
