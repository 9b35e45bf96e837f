"""-m gpu: verifySignatureList on the HIP engine vs the oracle's verifier (same verdict and status for the same
verifier seed), on honest, tampered, partially tampered and malformed proofs."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu


def _setup(S, nkeys, B, sec=80):
    import coracle as CO
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    eng.set_timing(1)   # the tests below read which kernel families ran off zk_last_timing
    nh, tg, th = eng.synth_params(S)
    eng.set_params(nh, tg, th, sec)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, B)
    eng.set_ring(ring, nkeys)
    octx = CO.OracleCtx(nh, tg, th, sec)
    octx.set_ring(ring, nkeys)
    proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    return eng, octx, msg, proofs


def _vseeds(n, tag=b'v'):
    return b''.join(hashlib.sha256(tag + bytes([i & 255, i >> 8])).digest() for i in range(n))


def _both(eng, octx, msg, proofs, vseeds):
    g = eng.verify_batch(msg, proofs, vseeds=vseeds)
    o = octx.verify_batch(msg, proofs, nthreads=16, vseeds=vseeds)
    return g, o


@pytest.mark.parametrize('nkeys,B,sec', [(8, 4, 80), (37, 3, 80), (2, 2, 80), (5, 2, 20), (1024, 6, 80)])
def test_honest_proofs_verify(nkeys, B, sec):
    eng, octx, msg, proofs = _setup(500 + nkeys, nkeys, B, sec)
    vs = _vseeds(B)
    g, o = _both(eng, octx, msg, proofs, vs)
    assert g == o == ([1] * B, [0] * B)
    g2 = eng.verify_batch(msg, proofs)  # default seeds
    assert g2 == ([1] * B, [0] * B)
    eng.close()


def test_tampered_proofs_rejected_like_the_oracle():
    eng, octx, msg, proofs = _setup(77, 8, 1)
    base = proofs[0]
    n, sec = 3, 80
    gk_off = len(base) - (n * (4 * 72 + 96) + 32)
    cases = {}

    def flip(name, pos, bit=1):
        b = bytearray(base)
        b[pos] ^= bit
        cases[name] = bytes(b)
    flip('zd', len(base) - 1)
    flip('gk_f0', gk_off + 4 * 72 * n + 31)
    flip('gk_za1', gk_off + 4 * 72 * n + 32 * (n + 1) + 31)
    flip('comS1_offcurve', 96 + 5)
    flip('R_offcurve', 32 + 40)
    flip('keyX_offcurve', 160 + 60)
    flip('header_len', 7)
    flip('header_bits', 31)
    flip('rep0_scalar', 304 + 208 + 31)
    # swap two valid Tom points (cl_0 <-> ca_0): still on the curve, wrong statement
    b = bytearray(base)
    b[gk_off:gk_off + 72], b[gk_off + 72 * n:gk_off + 72 * n + 72] = b[gk_off + 72 * n:gk_off + 72 * n + 72], b[gk_off:gk_off + 72]
    cases['swap_cl_ca'] = bytes(b)
    # keyXcom <-> keyYcom
    b = bytearray(base)
    b[160:232], b[232:304] = b[232:304], b[160:232]
    cases['swap_kx_ky'] = bytes(b)
    cases['truncated'] = base[:-4]
    names = sorted(cases)
    plist = [cases[k] for k in names]
    msgs = msg[:32] * len(plist)
    # 'truncated' is not a multiple-of-4 problem (still aligned), keep it last so that packing stays aligned
    plist.append(plist.pop(names.index('truncated')))
    names.append(names.pop(names.index('truncated')))
    for tag in (b'a', b'b', b'c'):
        vs = _vseeds(len(plist), tag)
        g, o = _both(eng, octx, msgs, plist, vs)
        assert g[0] == o[0], dict(zip(names, zip(g[0], o[0])))
        # statuses: the exact codes (the facade re-throws the reference's error text from them); tests/test_gpu_mutants.py sweeps this
        assert g[1] == o[1], dict(zip(names, zip(g[1], o[1])))
        assert sum(g[0]) <= 1  # only a flipped scalar in an unchecked rep may still pass
    # wrong message
    g, o = _both(eng, octx, bytes(32), [base], _vseeds(1))
    assert g == o == ([0], [0])
    eng.close()


def test_partial_tampering_follows_the_sampled_subset():
    """One bad rep out of 80: the reference accepts iff that rep is not among the 20 sampled ones (exp.ts:95-109).
    Engine and oracle must agree for every verifier seed."""
    eng, octx, msg, proofs = _setup(99, 6, 1)
    base = bytearray(proofs[0])
    # corrupt the first response scalar of rep 5 (offset found by walking the header bits)
    bits = int.from_bytes(base[16:32], 'big')
    off = 304
    for i in range(5):
        off += 336 + (0 if (bits >> i) & 1 else 3392)
    base[off + 208 + 31] ^= 1
    bad = bytes(base)
    nseeds = 24
    vs = _vseeds(nseeds, b'sub')
    g, o = _both(eng, octx, msg[:32] * nseeds, [bad] * nseeds, vs)
    assert g == o
    assert 0 < sum(g[0]) < nseeds  # some seeds sample rep 5, some do not
    eng.close()


def test_gk_length_mismatch_is_false_not_an_error():
    eng8, octx8, msg, proofs8 = _setup(123, 8, 1)     # n = 3
    eng16, octx16, msg16, proofs16 = _setup(123, 16, 1)  # n = 4
    g = eng16.verify_batch(msg, proofs8, vseeds=_vseeds(1))
    o = octx16.verify_batch(msg, proofs8, vseeds=_vseeds(1))
    assert g == o == ([0], [0])  # gk.ts:208-218 returns false
    eng8.close()
    eng16.close()


def test_json_wire_round_trip_then_verify():
    """test/zkpAttestList.test.ts:55-60: prove -> writeJson -> readJson -> verify (here through the C-ABI converters)."""
    import json
    import zkattest_ref as R
    import zkp_ecdsa_amd as Z
    eng, octx, msg, proofs = _setup(4242, 6, 3)
    texts = [Z.write_json(p) for p in proofs]
    assert texts[0] == R.proof_to_json(R.proof_from_bytes(proofs[0]))
    back = [Z.read_json(t) for t in texts]
    assert back == proofs
    assert eng.verify_batch(msg, back, vseeds=_vseeds(3)) == ([1] * 3, [0] * 3)
    # a proof edited on the wire parses but does not verify; an off-curve point parses and is refused at validation
    t = json.loads(texts[1])
    t['membershipProof']['zd']['k'] = hex(int(t['membershipProof']['zd']['k'], 16) ^ 1)
    forged = Z.read_json(json.dumps(t))
    t = json.loads(texts[2])
    t['R']['x'] = hex(int(t['R']['x'], 16) ^ 2)
    offcurve = Z.read_json(json.dumps(t))
    g, o = _both(eng, octx, msg, [back[0], forged, offcurve], _vseeds(3))
    assert g == o
    assert g[0] == [1, 0, 0] and g[1][1] == 0 and g[1][2] != 0
    eng.close()


def test_batched_and_per_proof_verification_agree():
    """zk_ctx_set_batch_verify: the chunk-wide bucket-method check (k_msm.hip) and the per-proof sums give the same verdicts and
    statuses -- all-honest chunks (fast path), chunks with a forged proof (fallback), ragged chunks, one and two lanes."""
    eng, octx, msg, proofs = _setup(606, 12, 10)
    vs = _vseeds(10)
    bad = bytearray(proofs[7])
    bad[-1] ^= 1                                  # zd
    forged = proofs[:7] + [bytes(bad)] + proofs[8:]
    exp_ok = {True: ([1] * 10, [0] * 10), False: ([1] * 7 + [0] + [1] * 2, [0] * 10)}
    for chunk in (10, 4):
        eng.set_chunk(chunk)
        for lanes in (1, 2):
            eng.set_lanes(lanes)
            for on in (True, False):
                eng.set_batch_verify(1 if on else 0)
                assert eng.verify_batch(msg, proofs, vseeds=vs) == exp_ok[True], (chunk, lanes, on)
                fam = eng.last_timing()[1]
                # an all-honest batch must be settled by the chunk-wide sum alone (no silent fallback)
                assert ('v_msm_tom' in fam) == on and ('v_straus_tom' in fam) == (not on), (fam, on)
                assert eng.verify_batch(msg, forged, vseeds=vs) == exp_ok[False], (chunk, lanes, on)
                fam = eng.last_timing()[1]
                assert 'v_straus_tom' in fam
                assert eng.verify_batch(msg, proofs) == exp_ok[True]       # OS-random seeds
    assert octx.verify_batch(msg, forged, nthreads=8, vseeds=vs) == exp_ok[False]
    eng.close()


def test_key_grouping_of_the_batched_check_is_exact(monkeypatch):
    """k_msm.hip groups the (window, group, digit) keys of all live terms with its own two counting passes (no library sort).  ZK_MSM_CHECK=1 makes
    run_msm audit the result on the device before the bucket sums use it: every listed position holds a live term whose key owns that position, no term
    twice, as many positions per window as there are non-zero digits.  Both shapes (8 groups x 16-bit windows, 64 x 13-bit), a ragged tiny chunk and
    a chunk of 3 000 proofs; honest proofs accepted and forged ones found through the audited pass."""
    monkeypatch.setenv('ZK_MSM_CHECK', '1')
    eng, octx, msg, proofs = _setup(611, 12, 10)
    vs = _vseeds(10)
    bad = bytearray(proofs[3])
    bad[-1] ^= 1
    forged = proofs[:3] + [bytes(bad)] + proofs[4:]
    for groups in (8, 64):
        eng.set_verify_groups(groups)
        for chunk in (10, 4):
            eng.set_chunk(chunk)
            eng.set_batch_verify(1)
            assert eng.verify_batch(msg, proofs, vseeds=vs) == ([1] * 10, [0] * 10), (groups, chunk)
            assert 'v_msm_tom' in eng.last_timing()[1]
            assert eng.verify_batch(msg, forged, vseeds=vs) == ([1] * 3 + [0] + [1] * 6, [0] * 10), (groups, chunk)
    eng.close()
    import zkp_ecdsa_amd as Z
    B, nkeys = 3000, 4096
    eng = Z.Engine(0)
    eng.set_timing(1)   # the tests below read which kernel families ran off zk_last_timing
    eng.set_params(*eng.synth_params(78), 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(78, nkeys, B)
    eng.set_ring(ring, nkeys)
    eng.set_chunk(B)
    proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    vs = _vseeds(B)
    for groups in (8, 64):
        eng.set_verify_groups(groups)
        assert eng.verify_batch(msg, proofs, vseeds=vs) == ([1] * B, [0] * B), groups
        assert 'v_straus_tom' not in eng.last_timing()[1]
    eng.close()


def test_one_forged_proof_only_sends_its_group_to_the_per_proof_sums():
    """The chunk-wide check sums MSM_G = 8 contiguous groups of proofs separately: one forged proof must cost about an eighth
    of the per-proof work of its chunk (not all of it), and the verdicts must not change.  Forgeries at a group boundary, in
    the first and in the last group exercise the range views of the per-proof path."""
    import zkp_ecdsa_amd as Z
    B, nkeys = 8192, 8192
    eng = Z.Engine(0)
    eng.set_timing(1)   # the tests below read which kernel families ran off zk_last_timing
    params = eng.synth_params(77)
    eng.set_params(*params, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(77, nkeys, B)
    eng.set_ring(ring, nkeys)
    eng.set_chunk(B)
    proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    vs = _vseeds(B)
    assert eng.verify_batch(msg, proofs, vseeds=vs) == ([1] * B, [0] * B)
    assert 'v_straus_tom' not in eng.last_timing()[1]
    eng.set_batch_verify(0)
    assert eng.verify_batch(msg, proofs, vseeds=vs) == ([1] * B, [0] * B)
    assert 'v_straus_tom' in eng.last_timing()[1]
    eng.set_batch_verify(256)
    groups = {(2700,): 1, (0,): 1, (B - 1,): 1, (1023, 1024): 2, (3, 6000): 2}
    for bad, ng in groups.items():    # groups of 1024 proofs
        bad = list(bad)
        forged = list(proofs)
        for b in bad:
            f = bytearray(proofs[b])
            f[-9] ^= 2                                   # zd
            forged[b] = bytes(f)
        before = eng.test_counter(0)
        ok, vst = eng.verify_batch(msg, forged, vseeds=vs)
        assert [b for b in range(B) if not ok[b]] == bad and vst == [0] * B
        # WORK, not time (the boxes of the pool differ by 2-3x): exactly the groups holding a forgery went through the per-proof sums
        assert eng.test_counter(0) - before == ng * (B // 8), (bad, eng.test_counter(0) - before)
        assert 'v_straus_tom' in eng.last_timing()[1]
    before = eng.test_counter(0)
    assert eng.verify_batch(msg, proofs, vseeds=vs) == ([1] * B, [0] * B)
    assert eng.test_counter(0) == before                 # an all-honest chunk never reaches them
    eng.close()


def test_batched_check_with_every_kind_of_bad_proof_in_the_chunk():
    """A chunk mixing honest proofs with malformed, forged, partially tampered and wrong-statement ones: the chunk-wide sum
    (forced on: min_chunk = 1) must end in exactly the verdicts and statuses of the per-proof sums, for several verifier
    seeds, and of the oracle; then the same chunk with the bad proofs replaced (all honest) goes through the fast path."""
    eng, octx, msg, proofs = _setup(4711, 16, 12)   # ring >= batch: every proof's key is in the ring
    n = 4
    mixed = list(proofs)

    def tamper(i, pos, bit=1):
        b = bytearray(proofs[i])
        b[pos] ^= bit
        mixed[i] = bytes(b)
    gk_off = lambda p: len(p) - (n * (4 * 72 + 96) + 32)
    tamper(1, len(proofs[1]) - 1)                         # zd: membership relation fails
    tamper(3, 32 + 40)                                    # R off the curve: status 10
    tamper(4, gk_off(proofs[4]) + 4 * 72 * n + 31)        # f_0
    tamper(6, 304 + 208 + 31)                             # a response scalar of rep 0: depends on the sampled subset
    tamper(8, 31)                                         # header bits: layout / 'params not found'
    mixed[10] = proofs[9]                                 # valid proof for another message
    for tag in (b'x', b'y', b'z', b'w'):
        vs = _vseeds(12, tag)
        eng.set_batch_verify(0)
        ref = eng.verify_batch(msg, mixed, vseeds=vs)
        eng.set_batch_verify(1)
        got = eng.verify_batch(msg, mixed, vseeds=vs)
        assert got == ref, (tag, got, ref)
        o = octx.verify_batch(msg, mixed, nthreads=8, vseeds=vs)
        assert got == o, (tag, got, o)
        assert got[0][0] == 1 and got[0][1] == 0 and got[0][3] == 0 and got[0][4] == 0 and got[0][10] == 0
    # bad proofs whose terms never reach the sums (malformed ones) must not spoil the fast path for the others
    only_malformed = list(proofs)
    b = bytearray(proofs[3])
    b[32 + 40] ^= 1
    only_malformed[3] = bytes(b)
    got = eng.verify_batch(msg, only_malformed, vseeds=_vseeds(12))
    assert got[0] == [1, 1, 1, 0] + [1] * 8 and got[1][3] == 10
    fam = eng.last_timing()[1]
    assert 'v_msm_tom' in fam and 'v_straus_tom' not in fam
    eng.close()


def test_offsets_that_run_backwards_are_an_argument_error():
    import zkp_ecdsa_amd as Z
    eng, octx, msg, proofs = _setup(5150, 4, 3)
    import ctypes as C
    raw = b''.join(proofs)
    buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
    off = (C.c_uint64 * 4)(0, len(proofs[0]), len(proofs[0]) - 4, len(raw))
    with pytest.raises(Z.ZkError) as e:
        eng.verify_batch_host_raw(msg, buf, off, 3)
    assert e.value.status == 14
    off = (C.c_uint64 * 4)(0, len(proofs[0]), len(proofs[0]) + len(proofs[1]), len(raw))
    _, ok, st = eng.verify_batch_host_raw(msg, buf, off, 3)
    assert list(ok) == [1, 1, 1] and list(st) == [0, 0, 0]
    eng.close()


@pytest.mark.parametrize('nkeys,B', [(4096, 37), (8192, 300), (65536, 1029)])
def test_ring_fold_on_the_matrix_pipe_equals_the_vector_form(nkeys, B):
    """verifyMembership's total (gk.ts:239-250) through v_mfma_i32_16x16x64_i8 (k_gk_mfma.hip: balanced base-256 digits, exact
    int32 accumulation, the same final reduction) against the 64-bit multiply-add form: identical verdicts and statuses for
    honest proofs, forged membership responses and a proof made for another ring position; batch sizes that leave ragged tiles."""
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    eng.set_timing(1)   # the tests below read which kernel families ran off zk_last_timing
    eng.set_comb_bits(16)
    params = eng.synth_params(1212)
    eng.set_params(*params, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(1212, nkeys, B)
    eng.set_ring(ring, nkeys)
    eng.set_chunk(512)
    proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    bad = list(proofs)
    for b, pos in ((1, -9), (B // 2, -40), (B - 1, -9 - 32 * 3)):     # zd, a zb / za / f response of the GK proof
        f = bytearray(proofs[b])
        f[pos] ^= 0x20
        bad[b] = bytes(f)
    vs = _vseeds(B)
    got = {}
    for pipe in (0, 1):
        eng.set_ring_fold(pipe)
        got[pipe] = (eng.verify_batch(msg, proofs, vseeds=vs), eng.verify_batch(msg, bad, vseeds=vs))
    assert got[0] == got[1]
    assert got[1][0] == ([1] * B, [0] * B)
    assert [b for b in range(B) if not got[1][1][0][b]] == sorted({1, B // 2, B - 1})
    eng.close()


def test_sixty_four_groups_give_the_same_verdicts_and_a_finer_fallback():
    """zk_ctx_set_verify_groups(64): the chunk-wide check with 64 groups and 13-bit windows instead of 8 groups and 16-bit windows.
    Same verdicts and statuses for honest, forged and malformed proofs; a forged proof sends a 64th of its chunk (not an eighth) to
    the per-proof sums -- counted, not timed."""
    import zkp_ecdsa_amd as Z
    B, nkeys = 8192, 8192
    eng = Z.Engine(0)
    eng.set_timing(1)   # the tests below read which kernel families ran off zk_last_timing
    eng.set_comb_bits(16)
    params = eng.synth_params(64)
    eng.set_params(*params, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(64, nkeys, B)
    eng.set_ring(ring, nkeys)
    eng.set_chunk(B)
    proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    vs = _vseeds(B)
    mixed = list(proofs)
    for b, pos in ((5, -9), (127, -9), (128, 400), (4000, -40), (B - 1, -9)):
        f = bytearray(proofs[b])
        f[pos] ^= 4
        mixed[b] = bytes(f)
    mixed[700] = proofs[700][:32] + bytes(64) + proofs[700][96:]     # R = (0, 0): not on the curve
    res = {}
    for groups in (8, 64):
        eng.set_verify_groups(groups)
        before = eng.test_counter(0)
        honest = eng.verify_batch(msg, proofs, vseeds=vs)
        assert honest == ([1] * B, [0] * B) and eng.test_counter(0) == before
        res[groups] = (eng.verify_batch(msg, mixed, vseeds=vs), eng.test_counter(0) - before)
    assert res[8][0] == res[64][0]
    ok, vst = res[64][0]
    bad = [b for b in range(B) if not ok[b]]
    assert set(bad) >= {5, 127, 4000, B - 1, 700} and vst[700] != 0
    # forged proofs sit in groups {0} + {0 or 1} + {3} + {7} of 8 (1024 proofs each) and in {0}, {0 or 1}, {31}, {63}, ... of 64 (128 each)
    assert res[64][1] < res[8][1] and res[64][1] <= 6 * (B // 64) and res[8][1] >= 3 * (B // 8)
    eng.close()


def _rep_offsets(p, sec=80):
    bits = int.from_bytes(p[16:32], 'big')
    out, off = [], 304
    for i in range(sec):
        out.append(off)
        off += 336 + (0 if (bits >> i) & 1 else 3392)
    return out


def _bad_p256_only(p):
    """second response scalar of EVERY repetition (s_H of the P-256 relation, exp.ts:270-276 / 305-317): whichever repetitions the verifier samples,
    the proof's P-256 sum is off while its Tom-256 relations may all hold"""
    b = bytearray(p)
    for o in _rep_offsets(p):
        b[o + 240 + 31] ^= 1
    return bytes(b)


def test_p256_relations_summed_across_proofs_give_the_same_verdicts(monkeypatch):
    """k_pmsm.hip (the P-256 half): chunks of at least ZKATTEST_P256_BATCH proofs (forced to 1 here) sum their P-256 relations per GROUP of
    proofs with the bucket method.  Honest chunks must be settled by that pass alone (counter 3, no per-proof P-256 family in the timing); a proof whose
    P-256 relation alone is broken must be found through the fallback; ragged chunks, empty groups, one and two lanes, 8 and 64 groups; the oracle agrees."""
    monkeypatch.setenv('ZKATTEST_P256_BATCH', '1')
    eng, octx, msg, proofs = _setup(4712, 32, 24)
    monkeypatch.delenv('ZKATTEST_P256_BATCH')
    B = 24
    vs = _vseeds(B)
    forged = list(proofs)
    forged[5], forged[20] = _bad_p256_only(proofs[5]), _bad_p256_only(proofs[20])
    want = ([0 if i in (5, 20) else 1 for i in range(B)], [0] * B)
    eng.set_batch_verify(1)
    for groups in (8, 64):
        eng.set_verify_groups(groups)
        for chunk, lanes in ((12, 1), (12, 2), (7, 2)):
            eng.set_chunk(chunk), eng.set_lanes(lanes)
            c3, c0 = eng.test_counter(3), eng.test_counter(0)
            assert eng.verify_batch(msg, proofs, vseeds=vs) == ([1] * B, [0] * B), (groups, chunk, lanes)
            fam = eng.last_timing()[1]
            assert eng.test_counter(3) - c3 == B and eng.test_counter(0) == c0 and 'v_msm_p256' in fam and 'v_straus_p256' not in fam, (groups, chunk, lanes, fam)
            c3 = eng.test_counter(3)
            assert eng.verify_batch(msg, forged, vseeds=vs) == want, (groups, chunk, lanes)
            # counter 3: proofs settled by the pass -- every GROUP without a forged proof (round 6: the fallback re-checks failing groups, not their chunk)
            settled = 0
            for s in range(0, B, chunk):
                c = min(chunk, B - s)
                gsz = (c + groups - 1) // groups
                settled += sum(min(c, g0 + gsz) - g0 for g0 in range(0, c, gsz) if not any(s + g0 <= b < s + min(c, g0 + gsz) for b in (5, 20)))
            assert eng.test_counter(3) - c3 == settled and 'v_straus_p256' in eng.last_timing()[1]
            assert eng.verify_batch(msg, proofs) == ([1] * B, [0] * B)   # OS-random seeds
    assert octx.verify_batch(msg, forged, nthreads=8, vseeds=vs) == want
    eng.close()


def test_p256_cross_proof_pass_at_its_default_size():
    """Two chunks of 8 192 proofs (the default threshold): groups of 1 024 proofs fill the buckets like the bench does -- the windows above bit 128 hold only
    SL >> 128, a handful of values, so their buckets take the oversized path (k_pm_big).  Honest: settled by the pass; forgeries of the P-256 relation in the
    first, a middle and the last group: found, and only their GROUP of 1 024 proofs pays the per-proof sums (round 6; until then the whole chunk did)."""
    import zkp_ecdsa_amd as Z
    B, nkeys = 16384, 16384
    eng = Z.Engine(0)
    eng.set_timing(1)   # the tests below read which kernel families ran off zk_last_timing
    eng.set_comb_bits(16)
    eng.set_params(*eng.synth_params(91), 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(91, nkeys, B)
    eng.set_ring(ring, nkeys)
    eng.set_chunk(8192), eng.set_lanes(2)
    proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    vs = _vseeds(B)
    c3 = eng.test_counter(3)
    assert eng.verify_batch(msg, proofs, vseeds=vs) == ([1] * B, [0] * B)
    fam = eng.last_timing()[1]
    assert eng.test_counter(3) - c3 == B and 'v_msm_p256' in fam and 'v_straus_p256' not in fam, fam
    for bad in ((0,), (8191 + 3000,), (B - 1,), (1023, 1024)):
        forged = list(proofs)
        for b in bad:
            forged[b] = _bad_p256_only(proofs[b])
        c3 = eng.test_counter(3)
        ok, vst = eng.verify_batch(msg, forged, vseeds=vs)
        assert [b for b in range(B) if not ok[b]] == list(bad) and vst == [0] * B
        assert eng.test_counter(3) - c3 == B - 1024 * len({b // 1024 for b in bad})   # every group without a forged proof
    # the same with 64 groups of 128 proofs (10-bit digits, 14 windows)
    eng.set_verify_groups(64)
    c3 = eng.test_counter(3)
    assert eng.verify_batch(msg, proofs, vseeds=vs) == ([1] * B, [0] * B) and eng.test_counter(3) - c3 == B
    forged = list(proofs)
    forged[127], forged[B - 129] = _bad_p256_only(proofs[127]), _bad_p256_only(proofs[B - 129])
    ok, vst = eng.verify_batch(msg, forged, vseeds=vs)
    assert [b for b in range(B) if not ok[b]] == [127, B - 129] and vst == [0] * B
    eng.close()
