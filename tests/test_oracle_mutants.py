"""CPU tier: the mutant generator of the GPU-tier differential sweep (tests/zka1_mutants.py) run through the ORACLE alone -- the C
restatement's verifier must reach every outcome the sweep claims to cover, the Python restatement must agree with it on a sample
(both are the checker: this pins the two against each other on malformed and adversarial inputs, not only on honest proofs), and the
prover's planted-fill / out-of-range cases must give the statuses the reference's throws map to."""
import hashlib

import coracle as CO
import zkattest_ref as R
from zka1_mutants import P256_N, Layout, mutants


def _xy(pt, w):
    x, y = pt.toAffine()
    return x.to_bytes(w, 'big') + y.to_bytes(w, 'big')


def _case(S, nkeys, B, sec=80):
    params = R.synth_params(S, sec)
    ring = R.synth_ring_fast(S, nkeys)
    ins = [R.synth_proof_input(S, b, nkeys) for b in range(B)]
    for m, s, p, w, d, seed in ins:
        ring[w] = R.keyToInt(p)
    c = CO.OracleCtx(_xy(params.NistGroup.h, 32), _xy(params.ProofGroup.g, 36), _xy(params.ProofGroup.h, 36), sec)
    c.set_ring(b''.join(v.to_bytes(32, 'big') for v in ring), nkeys)
    msg = b''.join(i[0] for i in ins)
    sig = b''.join(i[1] for i in ins)
    pk = b''.join(i[2][1:] for i in ins)
    which = [i[3] for i in ins]
    seeds = b''.join(i[5] for i in ins)
    return params, ring, c, msg, sig, pk, which, seeds


def _vseeds(n, tag):
    return b''.join(hashlib.sha256(tag + i.to_bytes(4, 'big')).digest() for i in range(n))


class _VerifierRng:
    """verifier-RNG contract (include/zkattest.h): fill k = SHA-256(seed || be64(k)); randomScalar takes the 32 bytes, rnd(small) the
    first byte of a fill"""

    def __init__(self, seed):
        self.seed, self.k = bytes(seed), 0

    def fill(self, nbytes):
        out = hashlib.sha256(self.seed + self.k.to_bytes(8, 'big')).digest()
        self.k += 1
        return out[:nbytes]


def _py_verify(params, ring, msg32, raw, vseed):
    """(ok, status) of the Python restatement under the verifier-RNG contract, exceptions mapped like include/zkattest.h."""
    try:
        proof = R.proof_from_bytes(raw)
    except Exception:
        return 0, 10
    try:
        return (1 if R.verifySignatureList(params, msg32, ring, proof, vrng=_VerifierRng(vseed)) else 0), 0
    except ValueError as e:
        text = str(e)
        for code, t in ((3, 'T is at infinity'), (4, 'T1 is at infinity'), (8, 'params not found'), (9, 'security level'), (7, 'R is at infinity')):
            if t in text:
                return 0, code
        raise


def test_the_sweep_reaches_every_verifier_outcome_and_both_restatements_agree():
    S, nkeys = 9001, 8
    params, ring, c, msg, sig, pk, which, seeds = _case(S, nkeys, 2)
    proofs, st = c.prove_batch(msg, sig, pk, which, seeds=seeds, nthreads=2)
    assert st == [0, 0]
    muts = mutants(proofs, 3, S, S)
    assert len(muts) >= 400
    names = [m[0] for m in muts]
    assert len(set(names)) > 380                     # (two strata may pick the same field twice)
    msgs = b''.join(msg[32 * m[1]:32 * m[1] + 32] for m in muts)
    plist = [m[2] for m in muts]
    seen = {}
    res = {}
    for tag in (b'm0', b'm1', b'm2'):
        vs = _vseeds(len(plist), tag)
        ok, vst = c.verify_batch(msgs, plist, nthreads=16, vseeds=vs)
        res[tag] = (ok, vst, vs)
        for i in range(len(plist)):
            seen.setdefault((ok[i], vst[i]), []).append(names[i])
    assert {(1, 0), (0, 0), (0, 3), (0, 4), (0, 8), (0, 10)} <= set(seen), sorted(seen)
    assert all(k[0] == 0 for k in seen if k[1] != 0)
    assert seen[(1, 0)].count('honest/p0') == 3 and seen[(1, 0)].count('honest/p1') == 3
    # mutants that may legitimately still verify touch one repetition only and pass when the verifier does not sample it (or, for
    # R.y + p, are another encoding of the same point, weier.ts:74-89): everything accepted besides the honest proofs is of that kind
    for nm in set(seen[(1, 0)]):
        assert nm.startswith(('honest', 'rep', 'alpha=', 'z=', 'T1inf', 'swap-padd', 'A-negated', 'Tx', 'swap-A-A', 'swap-Tx-Ty', 'R-y+p')), nm
    # the Python restatement on one mutant of every outcome and stratum head (full verifications take seconds each: a sample)
    ok, vst, vs = res[b'm0']
    picked, strata = [], set()
    for i, nm in enumerate(names):
        key = (nm.split('/')[0].rstrip('0123456789').split('-rep')[0].split('@')[0], ok[i], vst[i])
        if key not in strata and (vst[i] != 0 or len([p for p in picked if vst[p] == 0]) < 6):
            strata.add(key)
            picked.append(i)
    assert len(picked) >= 30
    for i in picked:
        got = _py_verify(params, ring, msgs[32 * i:32 * i + 32], plist[i], vs[32 * i:32 * i + 32])
        assert got == (ok[i], vst[i]), (names[i], got, (ok[i], vst[i]))


def test_planted_fills_and_which_out_of_range_in_the_oracle():
    """alpha_i = 0 -> 'T[i] is at infinity' (exp.ts:151); alpha_i = d / k -> T_i = pk -> 'T1 is at infinity' at the first zero-bit
    repetition (exp.ts:193); which >= N -> TypeError at gk.ts:162 (status 14), which in the padding proves for keys[0]."""
    S, nkeys = 777, 5
    params, ring, c, msg, sig, pk, which, seeds = _case(S, nkeys, 1)
    assert which == [0]
    nblk = 3 + 44 * 80 + 5 * 3 + 8
    d = R.fromBytes(R.synth_tag(b'sk', S, 0)) % (P256_N - 1) + 1
    k = R.fromBytes(R.synth_tag(b'nonce', S, 0)) % (P256_N - 1) + 1

    def stream(planted):
        blocks = [hashlib.sha256(seeds + j.to_bytes(8, 'big')).digest() for j in range(nblk)]
        for j, v in planted.items():
            blocks[j] = v.to_bytes(32, 'big')
        return b''.join(blocks)
    a = d * pow(k, -1, P256_N) % P256_N
    plans = [{}, {3 + 4 * 17: 0}, {3 + 4 * i: a for i in range(80)}, {3 + 4 * i: P256_N - a for i in range(80)}]
    B = len(plans)
    proofs, st = c.prove_batch(msg * B, sig * B, pk * B, [0] * B, streams=b''.join(stream(p) for p in plans), stream_blocks=nblk, nthreads=B)
    assert st[0] == 0 and st[1] == 3 and sorted(st[2:]) == [0, 4], st
    honest, _ = c.prove_batch(msg, sig, pk, [0], seeds=seeds)
    assert proofs[0] == honest[0]                    # an unplanted stream of the seed's fills IS the seed contract
    # the Python restatement throws the same two errors
    import pytest
    for plan, text in ((plans[1], 'T[i] is at infinity'), (plans[2 + st[2:].index(4)], 'T1 is at infinity')):
        blocks = stream(plan)
        with pytest.raises(ValueError, match=text.replace('[', r'\[').replace(']', r'\]')):
            R.proveSignatureList(params, msg, sig, b'\x04' + pk, 0, ring, R.StreamRng([blocks[32 * j:32 * j + 32] for j in range(nblk)]))
    # which: padding entries are keys[0]; past the padded ring the reference dies on undefined
    w = [0, 5, 7, 8, 0xffffffff]
    proofs, st = c.prove_batch(msg * 5, sig * 5, pk * 5, w, seeds=seeds * 5, nthreads=5)
    assert st == [0, 0, 0, 14, 14]
    assert c.verify_batch(msg * 3, proofs[:3], nthreads=3, vseeds=_vseeds(3, b'w')) == ([1] * 3, [0] * 3)
    assert len({bytes(p) for p in proofs[:3]}) == 3
    with pytest.raises(IndexError):
        R.proveSignatureList(params, msg, sig, b'\x04' + pk, 8, ring, R.SeedRng(seeds))
    p5 = R.proof_to_bytes(R.proveSignatureList(params, msg, sig, b'\x04' + pk, 5, ring, R.SeedRng(seeds)))
    assert p5 == proofs[1]


def test_layout_helper_matches_the_size_formula():
    S = 9001
    params, ring, c, msg, sig, pk, which, seeds = _case(S, 8, 1)
    proofs, st = c.prove_batch(msg, sig, pk, which, seeds=seeds)
    lay = Layout(proofs[0], 3)
    assert len(lay.zero_reps) + len(lay.one_reps) == 80
    assert len(lay.padd_points(lay.zero_reps[0])) == 32 and len(lay.padd_scalars(lay.zero_reps[0])) == 34
    assert lay.gk_scalars()[-1] + 32 == len(proofs[0])
