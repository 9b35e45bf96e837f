"""-m gpu: differential sweep of verifySignatureList on the HIP engine against the oracle's verifier -- seeded mutants of honest
ZKA1 proofs, stratified over every region of the layout, three verifier seeds each, rings of 8 and 1024 keys: engine
(ok, status) == oracle (ok, status), EXACT status codes (the facade re-throws the reference's error texts from them,
INTEGRATION.md section 4).  Plus the prover's planted-fill and degenerate-argument cases: alpha_i = 0 ('T[i] is at infinity',
exp.ts:151), alpha_i = d / k ('T1 is at infinity', exp.ts:193), `which` in the padding and past it, a one-key ring."""
import hashlib
import random

import pytest

pytestmark = pytest.mark.gpu

from zka1_mutants import P256_N, mutants


def _vseeds(n, tag):
    return b''.join(hashlib.sha256(tag + i.to_bytes(4, 'big')).digest() for i in range(n))


def _setup(S, nkeys, B):
    import coracle as CO
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(S)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, B)
    eng.set_ring(ring, nkeys)
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, nkeys)
    proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    return eng, octx, msg, proofs


@pytest.mark.parametrize('nkeys,S', [(8, 9001), (1024, 9002)])
def test_mutation_sweep_engine_equals_oracle_with_exact_status_codes(nkeys, S):
    eng, octx, msg, proofs = _setup(S, nkeys, 2)
    n = (nkeys - 1).bit_length()
    muts = mutants(proofs, n, S, S)
    assert len(muts) >= 400, len(muts)
    names = [m[0] for m in muts]
    msgs = b''.join(msg[32 * m[1]:32 * m[1] + 32] for m in muts)
    plist = [m[2] for m in muts]
    assert all(len(p) % 4 == 0 for p in plist)
    seen = {}
    for tag in (b'm0', b'm1', b'm2'):
        vs = _vseeds(len(plist), tag)
        g = eng.verify_batch(msgs, plist, vseeds=vs)
        o = octx.verify_batch(msgs, plist, nthreads=16, vseeds=vs)
        bad = [(names[i], (g[0][i], g[1][i]), (o[0][i], o[1][i])) for i in range(len(plist)) if (g[0][i], g[1][i]) != (o[0][i], o[1][i])]
        assert not bad, (tag, len(bad), bad[:12])
        for i in range(len(plist)):
            seen.setdefault((g[0][i], g[1][i]), []).append(names[i])
    # the sweep reaches every verifier outcome the ZKA1 form can produce: accepted, rejected, 'T is at infinity' (3), 'T1 is at
    # infinity' (4), 'params not found' (8), deserialisation failure (10)
    assert {(1, 0), (0, 0), (0, 3), (0, 4), (0, 8), (0, 10)} <= set(seen), sorted(seen)
    assert all(k[0] == 0 for k in seen if k[1] != 0)
    assert seen[(1, 0)].count('honest/p0') == 3 and seen[(1, 0)].count('honest/p1') == 3
    # an odd-length proof (its successor would be misaligned, so it goes last): both refuse it
    for cut in (1, 2, 3, 7):
        pl = [proofs[0], proofs[1][:-cut]]
        vs = _vseeds(2, b'odd')
        g = eng.verify_batch(msg[:64], pl, vseeds=vs)
        o = octx.verify_batch(msg[:64], pl, nthreads=2, vseeds=vs)
        assert g == o == ([1, 0], [0, 10])
    eng.close()


def test_a_proof_of_another_seclevel_is_refused_as_documented():
    """include/zkattest.h: a batch is homogeneous in secLevel -- a structurally perfect proof with another repetition count gets
    ZK_E_BAD_ENCODING from the engine (the reference would verify it with its own count, exp.ts:243-260: the oracle does)."""
    import coracle as CO
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(31)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(31, 8, 1)
    eng.set_params(nh, tg, th, 40)
    eng.set_ring(ring, 8)
    p40, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0]
    eng.set_params(nh, tg, th, 80)
    eng.set_ring(ring, 8)
    assert eng.verify_batch(msg, p40, vseeds=_vseeds(1, b's')) == ([0], [10])
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, 8)
    assert octx.verify_batch(msg, p40, vseeds=_vseeds(1, b's')) == ([1], [0])
    eng.close()


def _stream(seed, nblocks, planted):
    blocks = [hashlib.sha256(seed + k.to_bytes(8, 'big')).digest() for k in range(nblocks)]
    for k, v in planted.items():
        blocks[k] = v.to_bytes(32, 'big')
    return b''.join(blocks)


def test_planted_fills_reach_the_provers_exceptions_like_the_oracle():
    """ZK_RNG_STREAM with planted draws.  alpha_i = 0 on a repetition i > 0: T_i = 0 * R is the identity, 'T[i] is at infinity'
    (exp.ts:151, status 3).  alpha_i = d / k for every i: T_i = pk, so T1 = T_i - pk is the identity at the first zero-bit
    repetition, 'T1 is at infinity' (exp.ts:193, status 4).  Status 5 (pointAdd.ts:117-125) cannot be reached: P = T1 and R = T_i were just
    checked and Q = pk is a valid key; status 6 (pointAdd.ts:105) needs r = 0 mod n (tests/test_gpu_prove.py::test_error_statuses)."""
    import coracle as CO
    import zkattest_ref as R
    import zkp_ecdsa_amd as Z
    S, nkeys = 777, 8
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(S)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, 4)
    eng.set_ring(ring, nkeys)
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, nkeys)
    nblk = 3 + 44 * 80 + 5 * 3 + 8
    d = [R.fromBytes(R.synth_tag(b'sk', S, b)) % (P256_N - 1) + 1 for b in range(4)]
    k = [R.fromBytes(R.synth_tag(b'nonce', S, b)) % (P256_N - 1) + 1 for b in range(4)]
    # R = u1 G + u2 pk = +-k G (the synthetic signer may have normalised s): try both signs for proof 2 / 3
    a2 = d[2] * pow(k[2], -1, P256_N) % P256_N
    a3 = (P256_N - d[3] * pow(k[3], -1, P256_N)) % P256_N
    plans = [{},                                             # honest
             {3 + 4 * 17: 0},                                # alpha_17 = 0
             {3 + 4 * i: a2 for i in range(80)},             # alpha_i = d / k
             {3 + 4 * i: a3 for i in range(80)}]             # alpha_i = -d / k
    streams = b''.join(_stream(seeds[32 * b:32 * b + 32], nblk, plans[b]) for b in range(4))
    gp, gst = eng.prove_batch(msg, sig, pk, which, streams=streams, stream_blocks=nblk)
    op, ost = octx.prove_batch(msg, sig, pk, which, streams=streams, stream_blocks=nblk, nthreads=4)
    assert gst == ost, (gst, ost)
    assert gst[0] == 0 and gst[1] == 3 and sorted(gst[2:]) == [0, 4], gst
    assert gp == op
    # the same fills through the seed contract's honest stream: nothing planted, nothing thrown
    eng.close()


def test_which_in_the_padding_and_past_the_ring():
    """`which` in [n_keys, N) names a padding entry, i.e. keys[0] (gk.ts:75-86): the owner of keys[0] can prove with it and the bytes
    are the oracle's.  `which` >= N reads values[index].k of undefined (gk.ts:162): ZK_E_ARG from both, after the Exp phase."""
    import coracle as CO
    import zkp_ecdsa_amd as Z
    S, nkeys = 4321, 5                      # N = 8: indices 5, 6, 7 are copies of keys[0]
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(S)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, 1)   # proof 0 owns keys[0]
    eng.set_ring(ring, nkeys)
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, nkeys)
    B = 6
    w = [0, 5, 7, 8, 9, 0xffffffff]
    args = (msg * B, sig * B, pk * B, w)
    gp, gst = eng.prove_batch(*args, seeds=seeds * B)
    op, ost = octx.prove_batch(*args, seeds=seeds * B, nthreads=B)
    assert gst == ost == [0, 0, 0, 14, 14, 14]
    assert gp == op
    assert gp[0] != gp[1] and gp[1] != gp[2]          # the index bits enter the proof
    ok = eng.verify_batch(msg * 3, gp[:3], vseeds=_vseeds(3, b'w'))
    assert ok == ([1, 1, 1], [0, 0, 0]) == octx.verify_batch(msg * 3, gp[:3], nthreads=3, vseeds=_vseeds(3, b'w'))
    eng.close()


def test_a_one_key_ring_is_refused():
    """The reference cannot prove over one key either: n = 0 and interpolate([], []) evaluates -x[0] % m with x[0] undefined, a
    TypeError (interpolate.ts:40).  The engine refuses the ring itself (ZK_E_ARG), documented in include/zkattest.h."""
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    eng.set_params(*eng.synth_params(5), 80)
    ring, *_ = eng.synth_workload(5, 1, 0)
    with pytest.raises(Z.ZkError) as e:
        eng.set_ring(ring, 1)
    assert e.value.status == 14
    eng.close()


def test_mutants_in_small_calls_take_the_cooperating_wave_kernels_and_keep_their_statuses():
    """The sweep above verifies its mutants in ONE call of several hundred proofs: the batched checks first, then one-lane per-proof sums.  A call of a few proofs
    takes another path -- every term's Straus chain, both curves, and the doublings of R's table on cooperating waves (csrc/k_coop.hip; one proof per call is the
    reference's only shape, zkpAttestList.ts:147-184).  Every fourth mutant, eight per call and one per call, against the oracle: exact (ok, status) pairs, and
    zk_test_counter(4) proves the cooperative kernels were the ones that ran."""
    nkeys, S = 1024, 9002
    eng, octx, msg, proofs = _setup(S, nkeys, 2)
    muts = mutants(proofs, (nkeys - 1).bit_length(), S, S)[::4]
    assert len(muts) >= 100
    names = [m[0] for m in muts]
    msgs = [msg[32 * m[1]:32 * m[1] + 32] for m in muts]
    plist = [m[2] for m in muts]
    vs = _vseeds(len(plist), b'sc')
    o = octx.verify_batch(b''.join(msgs), plist, nthreads=16, vseeds=vs)
    c4 = eng.test_counter(4)
    got_ok, got_st = [], []
    for a in range(0, len(plist), 8):
        ok, st = eng.verify_batch(b''.join(msgs[a:a + 8]), plist[a:a + 8], vseeds=vs[32 * a:32 * a + 256])
        got_ok += ok
        got_st += st
    assert eng.test_counter(4) > c4, 'the small calls did not reach the cooperative kernels'
    bad = [(names[i], (got_ok[i], got_st[i]), (o[0][i], o[1][i])) for i in range(len(plist)) if (got_ok[i], got_st[i]) != (o[0][i], o[1][i])]
    assert not bad, (len(bad), bad[:12])
    assert {(1, 0), (0, 0)} <= set(zip(got_ok, got_st))
    for i in range(0, len(plist), 5):   # one proof per call
        assert eng.verify_batch(msgs[i], [plist[i]], vseeds=vs[32 * i:32 * i + 32]) == ([o[0][i]], [o[1][i]]), names[i]
    eng.close()
