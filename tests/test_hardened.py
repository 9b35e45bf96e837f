"""Hardened mode (include/zkattest.h: zk_hardened_h, zk_ctx_set_mode): the two TODOs of the reference as an opt-in that is
deliberately NOT byte-compatible with it -- h without a known discrete logarithm (src/commit/pedersen.ts:62) and the statement
hashed into the membership challenge (src/proofGK/gk.ts:178).  The oracle of this mode is the Python restatement of the
engine's own specification (oracle/zkattest_ref.py: hardened_h, ring_digest, gk_statement)."""
import hashlib

import pytest


def test_nums_generators_match_the_restatement_and_lie_in_the_groups():
    import zkattest_ref as R
    import zkp_ecdsa_amd as Z
    for tag in (b'', b'deployment-1', bytes(range(200))):
        nh, th = Z.hardened_h(tag)
        (hx, hy), (tx, ty) = R.hardened_h(tag)
        assert nh == hx.to_bytes(32, 'big') + hy.to_bytes(32, 'big')
        assert th == tx.to_bytes(36, 'big') + ty.to_bytes(36, 'big')
        assert hy % 2 == 0
        P = R.WeierstrassPoint(R.p256, hx, hy, 1)
        T = R.TEdwardsPoint(R.tomEdwards256, tx, ty)
        assert R.p256.isOnGroup(P) and R.tomEdwards256.isOnGroup(T)
        # prime order: (order - 1) * T + T = identity, and T is not the identity
        assert T.mul(R.tomEdwards256.newScalar(R.tomEdwards256.order - 1)).add(T).isIdentity() and not T.isIdentity()
    assert Z.hardened_h(b'a') != Z.hardened_h(b'b')


def test_ring_digest_definition():
    import zkattest_ref as R
    vals = [int.from_bytes(hashlib.sha256(bytes([i % 256, i // 256])).digest(), 'big') % R.p256.p for i in range(600)]
    padded = [v.k for v in R.pad(vals, R.tomEdwards256)]
    assert len(padded) == 1024
    leaves = b''.join(hashlib.sha256(b''.join(v.to_bytes(32, 'big') for v in padded[i:i + 256])).digest() for i in range(0, 1024, 256))
    assert R.ring_digest(padded) == hashlib.sha256(b'ZKAttest-ring-v1' + (1024).to_bytes(8, 'big') + leaves).digest()
    assert R.ring_digest(padded[:8]) == hashlib.sha256(b'ZKAttest-ring-v1' + (8).to_bytes(8, 'big') + hashlib.sha256(b''.join(v.to_bytes(32, 'big') for v in padded[:8])).digest()).digest()


@pytest.mark.gpu
@pytest.mark.parametrize('nkeys,sec', [(6, 20), (600, 20)])
def test_hardened_prove_and_verify_match_the_restatement(nkeys, sec):
    import zkattest_ref as R
    import zkp_ecdsa_amd as Z
    S, B = 31, 2
    eng = Z.Engine(0)
    nh, th = Z.hardened_h(b'test')
    _, tg, _ = eng.synth_params(S)
    eng.set_params(nh, tg, th, sec)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, B)
    eng.set_ring(ring, nkeys)
    keys = [int.from_bytes(ring[32 * i:32 * i + 32], 'big') for i in range(nkeys)]
    padded = [v.k for v in R.pad(keys, R.tomEdwards256)]
    assert eng.ring_digest() == R.ring_digest(padded)
    ref, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)            # reference mode
    eng.set_mode(Z.MODE_HARDENED)
    hard, st2 = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == st2 == [0] * B
    # the Python restatement with the same parameters, seeds and mode
    g = R.tomEdwards256.generator()
    params = R.SystemParametersList(
        R.PedersenParams(R.p256, R.p256.generator(), R.WeierstrassPoint(R.p256, int.from_bytes(nh[:32], 'big'), int.from_bytes(nh[32:], 'big'), 1)),
        R.PedersenParams(R.tomEdwards256, g, R.TEdwardsPoint(R.tomEdwards256, int.from_bytes(th[:36], 'big'), int.from_bytes(th[36:], 'big'))), sec)
    for b in range(B):
        args = (msg[32 * b:32 * b + 32], sig[64 * b:64 * b + 64], b'\x04' + pk[64 * b:64 * b + 64], which[b], keys)
        want = R.proof_to_bytes(R.proveSignatureList(params, *args, R.SeedRng(seeds[32 * b:32 * b + 32]), hardened=True))
        assert hard[b] == want
        assert ref[b] == R.proof_to_bytes(R.proveSignatureList(params, *args, R.SeedRng(seeds[32 * b:32 * b + 32])))
        # only the membership part differs (same commitments: same randomness), and it differs
        n = max(1, (nkeys - 1).bit_length())
        gk = n * (4 * 72 + 96) + 32
        assert hard[b][:-gk] == ref[b][:-gk] and hard[b][-gk:-gk + 4 * 72 * n] == ref[b][-gk:-gk + 4 * 72 * n] and hard[b][-32:] != ref[b][-32:]
        proof = R.proof_from_bytes(hard[b])
        assert R.verifySignatureList(params, args[0], keys, proof, hardened=True) and not R.verifySignatureList(params, args[0], keys, proof)
    vs = b''.join(hashlib.sha256(b'hv%d' % i).digest() for i in range(B))
    assert eng.verify_batch(msg, hard, vseeds=vs) == ([1] * B, [0] * B)
    assert eng.verify_batch(msg, ref, vseeds=vs) == ([0] * B, [0] * B)        # a reference-mode proof does not verify in hardened mode
    # the statement is bound: another message, or the same proof against a ring that differs in an unrelated entry
    other = bytes(32) + msg[32:]
    assert eng.verify_batch(other, hard, vseeds=vs)[0] == [0, 1]
    ring2 = bytearray(ring)
    ring2[32 * (nkeys - 1) + 31] ^= 1
    eng.set_ring(bytes(ring2), nkeys)
    assert eng.verify_batch(msg, hard, vseeds=vs)[0] == [0] * B
    eng.set_mode(Z.MODE_REFERENCE)
    eng.set_ring(ring, nkeys)
    assert eng.verify_batch(msg, ref, vseeds=vs) == ([1] * B, [0] * B) and eng.verify_batch(msg, hard, vseeds=vs)[0] == [0] * B
    eng.close()
