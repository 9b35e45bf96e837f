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
