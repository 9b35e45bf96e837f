"""-m gpu: proveSignatureList on the HIP engine vs the oracle, byte for byte (ZKA1), under the RNG contract."""
import hashlib
import os

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


def test_one_lane_and_cooperative_chains_make_the_same_bytes():
    """ZKATTEST_ONE_LANE_CHAINS (the round-5 one-lane kernels for the dependent chains: A/B switch of profiles/r06_ab_variants.txt) against the default
    (cooperating waves, csrc/coop.h): the same proofs and the same verdicts / statuses for honest and tampered proofs, each build of the switch in its own
    process (it is read once)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    recs = []
    # third run: the default kernels with the small verify calls' auxiliary streams waiting for stage 1 on the device (ZKATTEST_NO_HOST_RELEASE) instead of
    # being released by the host (api_verify.hip: VerifyJob::host_release; the one-lane run takes the waiting path too)
    for switch in (None, 'ZKATTEST_ONE_LANE_CHAINS', 'ZKATTEST_NO_HOST_RELEASE'):
        env = dict(os.environ)
        env.pop('ZKATTEST_ONE_LANE_CHAINS', None)
        env.pop('ZKATTEST_NO_HOST_RELEASE', None)
        if switch:
            env[switch] = '1'
        out = subprocess.run([sys.executable, os.path.join(root, 'tests', 'chains_check.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        recs.append(json.loads(out.stdout.decode().strip().splitlines()[-1]))
    assert recs[0]['one_lane'] is False and recs[1]['one_lane'] is True and recs[2]['one_lane'] is False
    for r in recs[1:]:
        assert recs[0]['sha256'] == r['sha256'] and recs[0]['verdicts'] == r['verdicts']


def test_uniform_control_flow_build_makes_the_same_bytes():
    """lib/libzkattest_hip_uniform.so (csrc/Makefile `uniform`, -DZK_UNIFORM_CF=1: the prover's table sums compute and discard the addition of a zero digit
    instead of branching around it, k_tom_commit never skips a window) is the same engine: same bytes from the one-lane and from the wide kernels.  Each
    library in its own process (one process holds one build); the cost of the uniform build is printed."""
    import json
    import os
    import subprocess
    import sys
    import zkp_ecdsa_amd as Z
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    uni = os.path.join(os.path.dirname(Z.LIB_PATH), 'libzkattest_hip_uniform.so')
    if not os.path.exists(uni):
        pytest.skip('the uniform build is not there (make -C zkp-ecdsa_amd/csrc uniform)')
    recs = []
    for lib in (Z.LIB_PATH, uni):
        env = dict(os.environ, ZKATTEST_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(root, 'tests', 'uniform_check.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        recs.append(json.loads(out.stdout.decode().strip().splitlines()[-1]))
    print(json.dumps(recs))
    assert recs[0]['sha256'] == recs[1]['sha256'] and recs[1]['lib'].endswith('_uniform.so')


def test_prove_ring_1024_sample():
    """BASELINE config 2 shape at reduced batch: ring 2^10, 64 proofs, all diffed against the oracle."""
    eng, octx, (msg, sig, pk, which, seeds) = _setup(7, 1024, 64)
    eng.set_chunk(48)  # two chunks, the second one ragged
    got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    exp, est = octx.prove_batch(msg, sig, pk, which, seeds=seeds, nthreads=64)
    assert st == est == [0] * 64
    assert [hashlib.sha256(g).hexdigest() for g in got] == [hashlib.sha256(e).hexdigest() for e in exp]
    eng.close()


@pytest.mark.parametrize('name', ['small_full', 'ring6_sec80', 'ring37_sec80', 'rejection_stream'])
def test_engine_matches_golden(name):
    """The committed fixtures (tests/golden/golden.json, made by the Python restatement) through the C ABI: seed mode,
    and stream mode with planted rejected fills (the rnd() retry path, big.ts:171-181, at whole-proof level)."""
    import json
    import os
    import zkp_ecdsa_amd as Z
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'golden.json')))
    case = gold[name]
    eng = Z.Engine(0)
    eng.set_params(bytes.fromhex(case['nist_h']), bytes.fromhex(case['tom_g']), bytes.fromhex(case['tom_h']), case['sec'])
    eng.set_ring(b''.join(int(v, 16).to_bytes(32, 'big') for v in case['ring']), case['nkeys'])
    for rec in case['proofs']:
        args = (bytes.fromhex(rec['msg']), bytes.fromhex(rec['sig']), bytes.fromhex(rec['pk']), [rec['which']])
        if 'seed' in rec:
            proofs, st = eng.prove_batch(*args, seeds=bytes.fromhex(rec['seed']))
        else:
            seed = bytes.fromhex(rec['stream_seed'])
            blocks = [hashlib.sha256(seed + k.to_bytes(8, 'big')).digest() for k in range(rec['stream_blocks'])]
            for idx, val in rec['plant']:
                blocks[idx] = int(val, 16).to_bytes(32, 'big')
            proofs, st = eng.prove_batch(*args, streams=b''.join(blocks), stream_blocks=len(blocks))
        assert st == [0]
        assert len(proofs[0]) == rec['len'] and hashlib.sha256(proofs[0]).hexdigest() == rec['sha256']
        if 'proof' in rec:
            assert proofs[0].hex() == rec['proof']
    eng.close()


def test_error_statuses():
    """Per-proof statuses mirror the reference's thrown errors; a bad proof does not disturb its neighbours."""
    eng, octx, (msg, sig, pk, which, seeds) = _setup(31, 8, 6)
    msg, sig, pk = bytearray(msg), bytearray(sig), bytearray(pk)
    pk[64 * 1 + 63] ^= 1                       # proof 1: public key off the curve
    sig[64 * 2 + 32:64 * 2 + 64] = bytes(32)   # proof 2: s = 0 -> R at infinity
    sig[64 * 4:64 * 4 + 32] = bytes(32)        # proof 4: r = 0 -> invMod(0) = 0 (big.ts:113-119): s1 = z1 = 0, T1 = T_i: "Points don't add up!"
    which = list(which)                        # (pointAdd.ts:105; rounds 1-3 reported 3 here)
    sig[64 * 5:64 * 5 + 32] = (0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551).to_bytes(32, 'big')   # proof 5: r = n, the same
    got, st = eng.prove_batch(bytes(msg), bytes(sig), bytes(pk), which, seeds=seeds)
    exp, est = octx.prove_batch(bytes(msg), bytes(sig), bytes(pk), which, seeds=seeds, nthreads=6)
    assert st == est == [0, 1, 3, 0, 6, 6]
    assert got[0] == exp[0] and got[3] == exp[3] and got[1] is None and got[2] is None and got[4] == exp[4]
    # stream too short -> randomness exhausted, not garbage
    blocks = b''.join(hashlib.sha256(b'blk%d' % k).digest() for k in range(100))
    _, st = eng.prove_batch(bytes(msg[:32]), bytes(sig[:64]), bytes(pk[:64]), which[:1], streams=blocks, stream_blocks=100)
    _, est = octx.prove_batch(bytes(msg[:32]), bytes(sig[:64]), bytes(pk[:64]), which[:1], streams=blocks, stream_blocks=100)
    assert st == est == [11]
    eng.close()


def test_baseline_config2_batch1024_ring1024_all_diffed():
    """BASELINE.json configs[1]: batch = 1024 proofs, ring = 2^10, every proof compared with the oracle (SHA-256 of the
    ZKA1 bytes), then every proof verified on the GPU."""
    eng, octx, (msg, sig, pk, which, seeds) = _setup(2024, 1024, 1024)
    eng.set_chunk(300)  # ragged chunks
    got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    exp, est = octx.prove_batch(msg, sig, pk, which, seeds=seeds, nthreads=64)
    assert st == est == [0] * 1024
    bad = [b for b in range(1024) if got[b] != exp[b]]
    assert not bad, bad[:10]
    ok, vst = eng.verify_batch(msg, got)
    assert ok == [1] * 1024 and vst == [0] * 1024
    eng.close()


def test_ring_2_20_prove_and_verify():
    """Largest ring of BASELINE.json (2^20 keys, n = 20): the multi-pass finish of the GK fold and the verifier's ring fold."""
    eng, octx, (msg, sig, pk, which, seeds) = _setup(5, 1 << 20, 3)
    got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    exp, est = octx.prove_batch(msg, sig, pk, which, seeds=seeds, nthreads=3)
    assert st == est == [0] * 3
    assert got == exp
    vs = b''.join(hashlib.sha256(b'vs%d' % i).digest() for i in range(3))
    assert eng.verify_batch(msg, got, vseeds=vs) == octx.verify_batch(msg, got, nthreads=3, vseeds=vs) == ([1] * 3, [0] * 3)
    bad = bytearray(got[1])
    bad[-40] ^= 4   # zb_{n-1}
    proofs = [got[0], bytes(bad), got[2]]
    assert eng.verify_batch(msg, proofs, vseeds=vs) == octx.verify_batch(msg, proofs, nthreads=3, vseeds=vs) == ([1, 0, 1], [0] * 3)
    eng.close()


def test_lanes_and_chunking_do_not_change_the_bytes():
    """One lane / two lanes / different chunk sizes: identical proofs (every proof only depends on its own inputs)."""
    eng, octx, (msg, sig, pk, which, seeds) = _setup(321, 48, 40)  # ring >= batch: every proof's key is in the ring
    ref = None
    for lanes, chunk in ((1, 40), (1, 7), (2, 7), (2, 13), (2, 20)):
        eng.set_lanes(lanes)
        eng.set_chunk(chunk)
        got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
        assert st == [0] * 40
        digest = hashlib.sha256(b''.join(got)).hexdigest()
        ref = ref or digest
        assert digest == ref, (lanes, chunk)
        ok, vst = eng.verify_batch(msg, got)
        assert ok == [1] * 40 and vst == [0] * 40
    exp, _ = octx.prove_batch(msg, sig, pk, which, seeds=seeds, nthreads=16)
    assert hashlib.sha256(b''.join(exp)).hexdigest() == ref
    # the comb width of the Tom-256 tables is a pure performance knob as well
    raw = eng.synth_params(321)
    for bits in (10, 20):
        eng.set_comb_bits(bits)
        eng.set_params(*raw, 80)
        got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
        assert st == [0] * 40 and hashlib.sha256(b''.join(got)).hexdigest() == ref, bits
        assert eng.verify_batch(msg, got) == ([1] * 40, [0] * 40)
    eng.close()


@pytest.mark.parametrize('nkeys,B,chunk', [(512, 300, 128), (700, 64, 64), (8192, 40, 40), (65536, 24, 16)])
def test_gk_table_path_equals_plain_fold_and_oracle(nkeys, B, chunk, monkeypatch):
    """The per-ring table path of the ring polynomial (k_gk.hip: 8 low index bits as one multilinear step, proofs
    sorted by l_low, XCD-segmented work list) against the plain fold (ZKATTEST_GK_TABLE=0) on every proof and against
    the oracle's 2*N*n loop + interpolation (gk.ts:141-171) on the first ones; rings on both sides of the
    one-proof-per-workgroup threshold (n = 9, 10 padded, 13, 16), several l_low groups per chunk, ragged chunks."""
    eng, octx, (msg, sig, pk, which, seeds) = _setup(900 + nkeys, nkeys, B)
    eng.set_chunk(chunk)
    got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    monkeypatch.setenv('ZKATTEST_GK_TABLE', '0')
    import zkp_ecdsa_amd as Z
    plain = Z.Engine(0)
    plain.set_params(*eng.synth_params(900 + nkeys), 80)
    ring = eng.synth_workload(900 + nkeys, nkeys, B)[0]
    plain.set_ring(ring, nkeys)
    plain.set_chunk(chunk)
    ref, st2 = plain.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st2 == [0] * B
    assert [hashlib.sha256(g).hexdigest() for g in got] == [hashlib.sha256(r).hexdigest() for r in ref]
    k = 4
    exp, est = octx.prove_batch(msg[:32 * k], sig[:64 * k], pk[:64 * k], which[:k], seeds=seeds[:32 * k], nthreads=k)
    assert est == [0] * k and got[:k] == exp
    assert eng.verify_batch(msg, got) == ([1] * B, [0] * B)
    eng.close(), plain.close()


@pytest.mark.parametrize('nkeys,B0', [(4096, 4096), (65536, 6144)])
def test_gk_table_path_on_the_matrix_cores_equals_the_vector_form(nkeys, B0, monkeypatch):
    """Coefficient classes 2..6 of the table path as int8 matrix products (k_gk_mfma.hip: k_gk_block_mfma, rings of >= 4096 keys)
    against the all-VALU table path (ZKATTEST_GK_MFMA_PROVE=0) on every proof and against the oracle on the first ones.  The
    workload is thinned to three l_low groups so that a group fills a 16-proof tile (ring 2^12: exactly one; ring 2^16: one and a
    partial one), plus one proof that is alone in its group.  (The generator gives proof b the ring slot b mod nkeys: B0 <= nkeys.)"""
    import coracle as CO
    import zkp_ecdsa_amd as Z
    S = 4400 + nkeys
    eng, vec = Z.Engine(0), None
    nh, tg, th = eng.synth_params(S)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, B0)
    eng.set_ring(ring, nkeys)
    groups = sorted({w & 255 for w in which})[:3]
    sel = [b for b in range(B0) if (which[b] & 255) in groups] + [next(b for b in range(B0) if (which[b] & 255) not in groups)]
    B = len(sel)
    assert max(sum(1 for b in sel if (which[b] & 255) == g) for g in groups) >= 16
    cut = lambda buf, n: b''.join(buf[n * b:n * (b + 1)] for b in sel)
    msg, sig, pk, seeds, which = cut(msg, 32), cut(sig, 64), cut(pk, 64), cut(seeds, 32), [which[b] for b in sel]
    got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    monkeypatch.setenv('ZKATTEST_GK_MFMA_PROVE', '0')
    vec = Z.Engine(0)
    vec.set_params(nh, tg, th, 80)
    vec.set_ring(ring, nkeys)
    ref, st2 = vec.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st2 == [0] * B
    assert [hashlib.sha256(g).hexdigest() for g in got] == [hashlib.sha256(r).hexdigest() for r in ref]
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, nkeys)
    exp, est = octx.prove_batch(msg[:32 * 2], sig[:64 * 2], pk[:64 * 2], which[:2], seeds=seeds[:32 * 2], nthreads=2)
    assert est == [0, 0] and got[:2] == exp
    assert eng.verify_batch(msg, got) == ([1] * B, [0] * B)
    eng.close(), vec.close()


@pytest.mark.parametrize('sec,nkeys,B', [(1, 4, 3), (7, 5, 4), (33, 12, 3), (128, 9, 2), (96, 300, 2)])
def test_security_levels_other_than_80(sec, nkeys, B):
    """secLevel is a run-time parameter of the reference (SystemParametersList.SecLevel): repetition counts that are not a
    multiple of 32, the maximum the 128-bit challenge allows, and a ring on the table path with a non-default level."""
    eng, octx, (msg, sig, pk, which, seeds) = _setup(7000 + sec, nkeys, B, sec)
    got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    exp, est = octx.prove_batch(msg, sig, pk, which, seeds=seeds, nthreads=4)
    assert st == est == [0] * B
    assert got == exp
    if sec >= 20:   # verifyExp needs secLevel >= 20 (zkpAttestList.ts:177); below that the reference throws
        assert eng.verify_batch(msg, got) == octx.verify_batch(msg, got, nthreads=4)[:2] == ([1] * B, [0] * B)
    else:
        # 'security level not achieved' (exp.ts:244): the reference throws it per call once membership has passed; the
        # engine refuses the whole batch up front (the level is a property of the context, not of a proof)
        import zkp_ecdsa_amd as Z
        with pytest.raises(Z.ZkError) as e:
            eng.verify_batch(msg, got)
        assert e.value.status == 9
        ook, ovst = octx.verify_batch(msg, got, nthreads=4)[:2]
        assert ook == [0] * B and ovst == [9] * B
    eng.close()


def test_c_abi_demo_program_runs(tmp_path):
    """examples/c_abi_demo.c: prove, JSON round trip, verify and reject a forged proof from a plain C host."""
    import os
    import subprocess
    import zkp_ecdsa_amd as Z
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / 'zk_demo'
    libdir = os.path.dirname(Z.LIB_PATH)
    subprocess.check_call(['gcc', '-std=c99', '-O1', '-I' + os.path.join(root, 'include'), os.path.join(root, 'examples', 'c_abi_demo.c'),
                           '-o', str(out), '-L' + libdir, '-lzkattest_hip', '-Wl,-rpath,' + libdir])
    res = subprocess.run([str(out), '5', '16'], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'round trip identical' in res.stdout and 'verified 5 of 5' in res.stdout and 'ok[0] = 0' in res.stdout


def test_page_locked_host_buffers_give_the_same_bytes_and_verdicts():
    """zk_host_alloc: `out` of zk_prove_batch is filled by per-chunk DMA behind the kernels, `proofs` of zk_verify_batch is read
    the same way; bytes, offsets, statuses and verdicts must equal those of pageable buffers (several ragged chunks, both lanes)."""
    import ctypes as C
    import zkp_ecdsa_amd as Z
    eng, octx, (msg, sig, pk, which, seeds) = _setup(31337, 64, 23)
    ref, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * 23
    pin = Z.PinnedBuffer(eng.proof_max_size() * 23)
    for chunk, lanes in ((5, 2), (23, 1), (8, 1), (4, 2)):
        eng.set_chunk(chunk), eng.set_lanes(lanes)
        C.memset(pin.ptr, 0xA5, pin.nbytes)
        _, out, off, pst = eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=pin)
        assert list(pst) == [0] * 23 and off[0] == 0
        raw = bytes(pin.view[:off[23]])
        assert [raw[off[b]:off[b + 1]] for b in range(23)] == ref, (chunk, lanes)
        vs = b''.join(hashlib.sha256(b'pl' + bytes([i])).digest() for i in range(23))
        _, ok, vst = eng.verify_batch_host_raw(msg, pin, off, 23, vseeds=vs)
        assert list(ok) == [1] * 23 and list(vst) == [0] * 23
        # a forged proof inside the page-locked buffer is found, the others still pass
        pos = off[7 + 1] - 1
        pin.view[pos] ^= 1
        _, ok, vst = eng.verify_batch_host_raw(msg, pin, off, 23, vseeds=vs)
        assert list(ok) == [1] * 7 + [0] + [1] * 15
        pin.view[pos] ^= 1
        # the same packed bytes from pageable memory
        page = (C.c_uint8 * off[23]).from_buffer_copy(raw)
        _, ok2, vst2 = eng.verify_batch_host_raw(msg, page, off, 23, vseeds=vs)
        assert list(ok2) == [1] * 23 and list(vst2) == [0] * 23
    # an `out` that is too small is reported, not overrun
    small = Z.PinnedBuffer(len(ref[0]) * 3)
    with pytest.raises(Z.ZkError):
        eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=small)
    small.free()
    pin.free()
    eng.close()


def test_key_tables_equal_the_per_proof_tables_and_fall_back_where_they_must(monkeypatch):
    """Per-key tables (k_ktab.hip; zk_ctx_set_key_tables): u2 * pk and alpha_i * R as sums of gathered multiples of the signer's ring key
    against the per-proof tables of R (ZKATTEST_KEYTAB=0) and the oracle, byte for byte.  The batch holds honest proofs (both square roots
    of a ring value occur among the keys), proofs whose `which` names ANOTHER key, and proofs that name ring values which are no
    x-coordinate at all: the latter two must take the per-proof path and still produce the bytes the reference would."""
    import coracle as CO
    import zkp_ecdsa_amd as Z
    S, nkeys, B = 5150, 300, 96
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(S)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, B)
    ring = bytearray(ring)
    for i in range(200, 232):   # 32 ring values that are (mostly) no key: half of all residues have no square root
        ring[32 * i:32 * i + 32] = hashlib.sha256(b'not a key %d' % i).digest()
    ring = bytes(ring)
    which = list(which)
    for b in range(60, 72):
        which[b] = (which[b] + 7) % 200        # another signer's slot
    for b in range(72, 96):
        which[b] = 200 + (b - 72)              # a slot that holds no x-coordinate of theirs
    eng.set_ring(ring, nkeys)
    eng.set_chunk(128)
    got, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    assert eng.test_counter(1) == 60           # exactly the honest proofs went through the key tables
    monkeypatch.setenv('ZKATTEST_KEYTAB', '0')
    ref = Z.Engine(0)
    ref.set_params(nh, tg, th, 80)
    ref.set_ring(ring, nkeys)
    ref.set_chunk(128)
    exp, st2 = ref.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st2 == [0] * B and ref.test_counter(1) == 0
    assert [hashlib.sha256(g).hexdigest() for g in got] == [hashlib.sha256(e).hexdigest() for e in exp]
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, nkeys)
    pick = [0, 1, 2, 59, 60, 71, 72, 95]
    cut = lambda buf, n: b''.join(buf[n * b:n * (b + 1)] for b in pick)
    oexp, est = octx.prove_batch(cut(msg, 32), cut(sig, 64), cut(pk, 64), [which[b] for b in pick], seeds=cut(seeds, 32), nthreads=8)
    assert est == [0] * len(pick) and [got[b] for b in pick] == oexp
    ok, vst = eng.verify_batch(msg, got)
    assert ok[:60] == [1] * 60 and not any(ok[60:]) and vst == [0] * B   # a proof for a slot that is not the signer's does not verify
    eng.close(), ref.close()


def test_degenerate_prover_inputs_give_the_oracles_bytes_and_statuses():
    """Inputs the reference does not reject and a service must not mis-handle: the reference validates neither the signature nor that the key sits at
    `which` (zkpAttestList.ts:104-145 -- it proves whatever it is handed; such proofs simply do not verify), reduces what it needs mod n and keeps
    invMod(0) = 0.  Whatever it makes of them, the engine makes the same bytes and statuses -- on the key-table path and without it."""
    n = 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551
    for keytab in (1, 0):
        import os
        os.environ['ZKATTEST_KEYTAB'] = str(keytab)
        try:
            eng, octx, (msg, sig, pk, which, seeds) = _setup(88, 8, 8)
        finally:
            del os.environ['ZKATTEST_KEYTAB']
        msg, sig, which = bytearray(msg), bytearray(sig), list(which)
        msg[32 * 1:32 * 2] = bytes(32)                          # z = 0: u1 = 0, z1 = 0, Q = 0 * G
        msg[32 * 2:32 * 3] = n.to_bytes(32, 'big')              # z = n: the same after reduction
        msg[32 * 3:32 * 4] = b'\xff' * 32                       # z >= n
        which[4] = (which[4] + 1) % 8                           # the key is not the one at `which`
        s5 = int.from_bytes(sig[64 * 5 + 32:64 * 5 + 64], 'big')
        sig[64 * 5 + 32:64 * 5 + 64] = (n - s5).to_bytes(32, 'big')   # s -> n - s (the other root: R -> -R)
        sig[64 * 6:64 * 6 + 32] = b'\xff' * 32                  # r >= n: reduced
        sig[64 * 7 + 32:64 * 7 + 64] = n.to_bytes(32, 'big')    # s = n = 0 mod n: R at infinity
        args = (bytes(msg), bytes(sig), pk, which)
        got, st = eng.prove_batch(*args, seeds=seeds)
        exp, est = octx.prove_batch(*args, seeds=seeds, nthreads=8)
        assert st == est, (keytab, st, est)
        assert st[7] == 3 and st[0] == 0
        for b in range(8):
            assert got[b] == exp[b], (keytab, b)
        live = [b for b in range(8) if st[b] == 0]
        vs = b''.join(hashlib.sha256(b'dg%d' % b).digest() for b in live)
        sub = lambda buf, w: b''.join(buf[w * b:w * b + w] for b in live)
        gv = eng.verify_batch(sub(args[0], 32), [got[b] for b in live], vseeds=vs)
        assert gv == octx.verify_batch(sub(args[0], 32), [got[b] for b in live], nthreads=8, vseeds=vs)
        assert gv[0][0] == 1
        eng.close()


def test_wipe_zeroes_the_workspaces_and_changes_nothing_else():
    """zk_ctx_wipe (also run by zk_ctx_destroy and after a failed prove call): the prover lanes' workspaces -- RNG stream, nonces, s1, blinders -- and the staged
    inputs are zeroed; the next call makes the same bytes; refused while a streamed job is in flight."""
    import zkp_ecdsa_amd as Z
    eng, octx, (msg, sig, pk, which, seeds) = _setup(515, 64, 6)
    a, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * 6
    eng.wipe()
    b, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * 6 and a == b
    assert eng.verify_batch(msg, b) == ([1] * 6, [0] * 6)
    eng.wipe()
    eng.wipe()   # idempotent
    # a failed call (output buffer too small) wipes by itself and leaves the context usable
    import ctypes as C
    with pytest.raises(Z.ZkError):
        eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=(C.c_uint8 * 1000)())
    c, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert c == a
    eng.close()
