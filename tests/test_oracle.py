"""not-gpu: pins the oracle.  Reference KATs and structural tests (test/bignum/big.test.ts, test/proofGK/interpolate.test.ts,
test/curves/ec.test.ts, test/curves/multimult.test.ts, test/commit/*.test.ts, test/exp/*.test.ts, test/proofGK/gk.test.ts)
re-stated against oracle/zkattest_ref.py, public vectors (RFC 6979 A.2.5, FIPS 180-4), and the committed golden
fixtures against both restatements."""
import hashlib
import json
import os
import random

import pytest

import coracle as CO
import zkattest_ref as R

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'golden.json')))


def test_reference_kats():
    assert R.invMod(3, 5) == 2 and R.invMod(7, 41) == 6               # test/bignum/big.test.ts:19-21
    assert R.interpolate([1, 2, 3], [1, 2, 3], 401) == [0, 1, 0]      # test/proofGK/interpolate.test.ts:19-26
    for a, m, r in GOLD['kats']['invMod']:
        assert R.invMod(a, m) == r
    x, y, m, c = GOLD['kats']['interpolate']
    assert R.interpolate(x, y, m) == c
    assert R.invMod(0, 41) == 0


def test_public_vectors():
    d = 0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721
    h = hashlib.sha256(b'sample').digest()
    assert R.rfc6979_k(d, h) == 0xA6E3C57DD01ABE90086538398355DD4C3B17AA873382B0F24D6129493D8AAD60
    sig = R.ecdsa_sign(d, h)
    assert sig.hex().upper() == ('EFD48B2AACB6A8FD1140DD9CD45E81D69D2C877B56AAF991C34D0EA84EAF3716'
                                 'F7CB1C942D657C41D436C7A1B6E29F65F3E900DBB9AFF4064DC4AB2F843ACDA8')
    pk = R.ecdsa_pubkey(d)
    assert pk.hex().upper()[2:] == ('60FED4BA255A9D31C961EB74C6356D68C049B8923B61FA6CE669622E60F29FB6'
                                    '7903FE1008B8BC99A41AE9E95628BC64F2F1B20C2D7E9F5177A3C294D4462299')
    assert R.ecdsa_verify(pk, h, sig)
    assert CO.sha256(b'abc').hex() == 'ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad'
    assert CO.sha256(b'abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq').hex() == \
        '248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1'
    assert CO.p256_mul(d) == pk[1:]


@pytest.mark.parametrize('group', [R.p256, R.tomEdwards256], ids=['p256', 'tom256'])
def test_ec_structure(group):  # test/curves/ec.test.ts:21-90
    rng = R.OsRng(b'ec')
    G = group.generator()
    assert G.mul(group.newScalar(group.order - 1)).add(G).isIdentity()
    P = G.mul(group.randomScalar(rng))
    for _ in range(5):
        P = P.mul(group.randomScalar(rng))
        assert group.isOnGroup(P)
    Q = P.mul(group.newScalar(group.order - 1))
    assert P.add(Q).isIdentity()
    k1, k2 = group.randomScalar(rng), group.randomScalar(rng)
    assert P.dblmul(k1, Q, k2).eq(P.mul(k1).add(Q.mul(k2)))
    I = group.identity()
    assert group.deserializePoint(I.toBytes()).eq(I)
    for _ in range(5):
        pt = G.mul(group.randomScalar(rng))
        assert group.deserializePoint(pt.toBytes()).eq(pt)
    assert len(G.toBytes()) == (65 if group is R.p256 else 67)


def test_c_oracle_field_and_curves():
    rnd = random.Random(1)
    for w, m in enumerate([R.p256.p, R.p256.order, R.tomEdwards256.p]):
        for _ in range(100):
            a, b = rnd.randrange(m), rnd.randrange(m)
            assert CO.field_op(w, 0, a, b) == a * b % m
            assert CO.field_op(w, 1, a, b) == (a + b) % m
            assert CO.field_op(w, 2, a, b) == (a - b) % m
        assert CO.field_op(w, 3, 0) == 0
        a = rnd.randrange(1, m)
        assert CO.field_op(w, 3, a) == pow(a, -1, m)
    for _ in range(3):
        k = rnd.randrange(R.p256.order)
        x, y = R.p256.generator().mul(R.p256.newScalar(k)).toAffine()
        assert CO.p256_mul(k) == x.to_bytes(32, 'big') + y.to_bytes(32, 'big')
        k = rnd.randrange(R.tomEdwards256.order)
        x, y = R.tomEdwards256.generator().mul(R.tomEdwards256.newScalar(k)).toAffine()
        assert CO.tom_mul(k) == x.to_bytes(36, 'big') + y.to_bytes(36, 'big')


def test_multimult_matches_naive():  # test/curves/multimult.test.ts:21-55
    g = R.tomEdwards256
    rng = R.OsRng(b'mm')
    known = g.generator().mul(g.randomScalar(rng))
    mm = R.MultiMult(g)
    mm.addKnown(known)
    naive = g.identity()
    for i in range(20):
        pt = known if i % 5 == 0 else g.generator().mul(g.randomScalar(rng))
        s = g.randomScalar(rng)
        mm.insert(pt, s)
        naive = naive.add(pt.mul(s))
    assert mm.evaluate().eq(naive)
    rel = R.Relation(g)
    P = g.generator().mul(g.randomScalar(rng))
    s = g.randomScalar(rng)
    rel.insertM([P, P.mul(s)], [s, g.newScalar(1).neg()])
    m2 = R.MultiMult(g)
    rel.drain(m2, rng)
    assert m2.evaluate().isIdentity()


def _pedersen(rng):
    return R.generatePedersenParams(R.tomEdwards256, rng)


def test_equality_mult_roundtrip():  # test/commit/equality.test.ts, mult.test.ts
    rng, v = R.OsRng(b'eqm'), R.OsRng(b'v')
    pp = _pedersen(rng)
    q = pp.c.order
    x = R.rnd(q, rng)
    C1, C2 = pp.commit(x, rng), pp.commit(x, rng)
    pi = R.proveEquality(pp, x, C1, C2, rng)
    mm = R.MultiMult(pp.c)
    assert R.aggregateEquality(pp, C1.p, C2.p, pi, mm, v) and mm.evaluate().isIdentity()
    y = R.rnd(q, rng)
    z = x * y % q
    Cx, Cy, Cz = pp.commit(x, rng), pp.commit(y, rng), pp.commit(z, rng)
    pm = R.proveMult(pp, x, y, z, Cx, Cy, Cz, rng)
    mm = R.MultiMult(pp.c)
    assert R.aggregateMult(pp, Cx.p, Cy.p, Cz.p, pm, mm, v) and mm.evaluate().isIdentity()
    mm = R.MultiMult(pp.c)
    R.aggregateMult(pp, Cx.p, Cy.p, pp.commit(z + 1, rng).p, pm, mm, v)
    assert not mm.evaluate().isIdentity()


def test_pointadd_and_exp_roundtrip():  # test/exp/pointAdd.test.ts, exp.test.ts (secLevel reduced for CPU time)
    rng, v = R.OsRng(b'pa'), R.OsRng(b'v2')
    pw = _pedersen(rng)
    ec = R.p256
    P, Q = ec.generator().mul(ec.randomScalar(rng)), ec.generator().mul(ec.randomScalar(rng))
    Rr = P.add(Q)
    cs = [pw.commit(c, rng) for pt in (P, Q, Rr) for c in pt.toAffine()]
    pi = R.provePointAdd(pw, P, Q, Rr, *cs, rng)
    mm = R.MultiMult(pw.c)
    assert R.aggregatePointAdd(pw, *[c.p for c in cs], pi, mm, v) and mm.evaluate().isIdentity()
    with pytest.raises(ValueError):
        R.provePointAdd(pw, P, Q, P, *cs, rng)  # "Points don't add up!"
    pn = R.generatePedersenParams(ec, rng)
    s = R.rnd(ec.order, rng)
    Pt = pn.g.mul(ec.newScalar(s))
    px, py = Pt.toAffine()
    Cs = pn.commit(s, rng)
    Px, Py = pw.commit(px, rng), pw.commit(py, rng)
    proofs = R.proveExp(pn, pw, s, Cs, Pt, Px, Py, 6, rng)
    assert R.verifyExp(pn, pw, Cs.p, Px.p, Py.p, proofs, 6, v)
    with pytest.raises(ValueError):
        R.verifyExp(pn, pw, Cs.p, Px.p, Py.p, proofs, 7, v)  # 'security level not achieved'


def test_gk_roundtrip_and_fold_form():  # test/proofGK/gk.test.ts:22-30
    rng, v = R.OsRng(b'gk'), R.OsRng(b'v3')
    pp = _pedersen(rng)
    vec = [3, 5, 7, 11, 13]
    com = pp.commit(11, rng)
    proof = R.proveMembership(pp, com, 3, vec, rng)
    assert R.verifyMembership(pp, com.p, vec, proof, v)
    assert not R.verifyMembership(pp, pp.commit(12, rng).p, vec, proof, v)
    q = pp.c.order
    x = 123456789
    fk = [s.k for s in proof.f]
    padded = [s.k for s in R.pad(vec, pp.c)]
    layer = padded[:]
    for j in range(len(fk)):
        layer = [((x - fk[j]) * layer[2 * i] + fk[j] * layer[2 * i + 1]) % q for i in range(len(layer) // 2)]
    assert layer[0] == R.gk_total_naive(padded, fk, x, q)  # the fold used by the oracle == gk.ts:239-250 as written


def _ctx(case):
    c = CO.OracleCtx(bytes.fromhex(case['nist_h']), bytes.fromhex(case['tom_g']), bytes.fromhex(case['tom_h']), case['sec'])
    ring = b''.join(int(v, 16).to_bytes(32, 'big') for v in case['ring'])
    c.set_ring(ring, case['nkeys'])
    return c


def _stream(rec):
    seed = bytes.fromhex(rec['stream_seed'])
    blocks = [hashlib.sha256(seed + k.to_bytes(8, 'big')).digest() for k in range(rec['stream_blocks'])]
    for idx, val in rec['plant']:
        blocks[idx] = int(val, 16).to_bytes(32, 'big')
    return blocks


@pytest.mark.parametrize('name', ['small_full', 'ring6_sec80', 'ring37_sec80', 'rejection_stream'])
def test_c_oracle_matches_golden(name):
    case = GOLD[name]
    c = _ctx(case)
    for rec in case['proofs']:
        args = (bytes.fromhex(rec['msg']), bytes.fromhex(rec['sig']), bytes.fromhex(rec['pk']), [rec['which']])
        if 'seed' in rec:
            proofs, st = c.prove_batch(*args, seeds=bytes.fromhex(rec['seed']))
        else:
            blocks = _stream(rec)
            proofs, st = c.prove_batch(*args, streams=b''.join(blocks), stream_blocks=len(blocks))
        assert st == [0]
        assert len(proofs[0]) == rec['len'] and hashlib.sha256(proofs[0]).hexdigest() == rec['sha256']
        if 'proof' in rec:
            assert proofs[0].hex() == rec['proof']
        ok, vst = c.verify_batch(args[0], proofs)
        assert ok == [1] and vst == [0]
        bad = bytearray(proofs[0])
        bad[-5] ^= 1                      # tamper zd
        assert c.verify_batch(args[0], [bytes(bad)])[0] == [0]
        bad = bytearray(proofs[0])
        bad[40] ^= 1                      # R.x off the curve -> deserialisation error
        assert c.verify_batch(args[0], [bytes(bad)])[1] != [0]
        ok, _ = c.verify_batch(bytes(32), proofs)   # wrong message
        assert ok == [0]


def test_python_restatement_matches_golden_and_layout():
    case = GOLD['small_full']
    rec = case['proofs'][0]
    raw = bytes.fromhex(rec['proof'])
    proof = R.proof_from_bytes(raw)
    assert R.proof_to_bytes(proof) == raw
    params = R.synth_params(case['S'], case['sec'])
    ring = [int(v, 16) for v in case['ring']]
    assert R.verifySignatureList(params, bytes.fromhex(rec['msg']), ring, proof)
    again = R.proveSignatureList(params, bytes.fromhex(rec['msg']), bytes.fromhex(rec['sig']), b'\x04' + bytes.fromhex(rec['pk']),
                                 rec['which'], ring, R.SeedRng(bytes.fromhex(rec['seed'])))
    assert R.proof_to_bytes(again) == raw
    # size formula used by the engine (include/zkattest.h)
    sec, n = case['sec'], len(proof.membershipProof.cl)
    z = sum(1 for e in proof.expProof if e.alpha is None)
    assert len(raw) == 304 + 336 * sec + 3392 * z + n * (4 * 72 + 3 * 32) + 32
    # draw accounting of SURVEY section 8 row a-0: 3 + 4 sec + 40 z + 5 n fills (no rejection in this vector)
    assert rec['fills_consumed'] == 3 + 4 * sec + 40 * z + 5 * n


def test_error_statuses_match_reference_throws():
    case = GOLD['small_full']
    c = _ctx(case)
    rec = case['proofs'][0]
    msg, sig, pk = bytes.fromhex(rec['msg']), bytes.fromhex(rec['sig']), bytes.fromhex(rec['pk'])
    seed = bytes.fromhex(rec['seed'])
    badpk = bytearray(pk)
    badpk[63] ^= 1
    _, st = c.prove_batch(msg, sig, bytes(badpk), [0], seeds=seed)
    assert st == [1]                                   # 'point not in group' (weier.ts:83)
    _, st = c.prove_batch(msg, sig[:32] + bytes(32), pk, [0], seeds=seed)
    assert st == [3]                                   # s = 0 -> R at infinity -> 'T[i] is at infinity' (exp.ts:151)
    with pytest.raises(ValueError):
        R.proveSignatureList(R.synth_params(case['S'], case['sec']), msg, sig[:32] + bytes(32), b'\x04' + pk, 0,
                             [int(v, 16) for v in case['ring']], R.SeedRng(seed))
    # r = 0 (and r = n): invMod(0) = 0 (big.ts:113-119) makes s1 = z1 = 0, T1 = T_i, and provePointAdd throws at the first zero-bit repetition
    n_be = (0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551).to_bytes(32, 'big')
    for r_bytes in (bytes(32), n_be):
        _, st = c.prove_batch(msg, r_bytes + sig[32:], pk, [0], seeds=seed)
        assert st == [6]                               # "Points don't add up!" (pointAdd.ts:105)
        with pytest.raises(ValueError, match="Points don't add up"):
            R.proveSignatureList(R.synth_params(case['S'], case['sec']), msg, r_bytes + sig[32:], b'\x04' + pk, 0,
                                 [int(v, 16) for v in case['ring']], R.SeedRng(seed))
