"""-m gpu: proveSignatureList on the HIP engine vs the oracle, byte for byte (ZKA1), under the RNG contract."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu


def _setup(S, nkeys, B, sec=80):
    import coracle as CO
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(S)
    eng.set_params(nh, tg, th, sec)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, B)
    eng.set_ring(ring, nkeys)
    octx = CO.OracleCtx(nh, tg, th, sec)
    octx.set_ring(ring, nkeys)
    return eng, octx, (msg, sig, pk, which, seeds)


@pytest.mark.parametrize('nkeys,B', [(6, 3), (8, 5), (2, 1), (37, 4)])
def test_prove_matches_oracle_small(nkeys, B):
    eng, octx, (msg, sig, pk, which, seeds) = _setup(1000 + nkeys, nkeys, B)
    got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    exp, est = octx.prove_batch(msg, sig, pk, which, seeds=seeds, nthreads=8)
    assert st == est == [0] * B
    for b in range(B):
        assert got[b] == exp[b], 'proof %d differs (first diff at byte %d)' % (
            b, next(i for i in range(min(len(got[b]), len(exp[b]))) if got[b][i] != exp[b][i]))
    ok, vst = octx.verify_batch(msg, got, nthreads=8)
    assert ok == [1] * B and vst == [0] * B
    eng.close()


def test_prove_ring_1024_sample():
    """BASELINE config 2 shape at reduced batch: ring 2^10, 64 proofs, all diffed against the oracle."""
    eng, octx, (msg, sig, pk, which, seeds) = _setup(7, 1024, 64)
    eng.set_chunk(48)  # two chunks, the second one ragged
    got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    exp, est = octx.prove_batch(msg, sig, pk, which, seeds=seeds, nthreads=64)
    assert st == est == [0] * 64
    assert [hashlib.sha256(g).hexdigest() for g in got] == [hashlib.sha256(e).hexdigest() for e in exp]
    eng.close()
