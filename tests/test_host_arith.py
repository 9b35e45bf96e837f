"""not-gpu: the engine's arithmetic headers (zkp-ecdsa_amd/csrc/field.h, curve.h) compiled for the host CPU (g++ -DZK_HOST_BUILD,
tests/host_arith/host_arith.cpp) against the oracle.  These are the templates the HIP kernels instantiate -- radix-2^30
Montgomery product, magnitude-typed lazy add/sub, fused double subtraction, Fermat inversion, the P-256 complete formulas
(weier.ts:133-230) and the Tom-256 extended / niels formulas on the a = 1 image (edwards.ts:141-183) -- so the CPU tier fails
when that source breaks, not only when the GPU tier runs."""
import ctypes as C
import os
import random
import shutil
import subprocess

import pytest

import zkattest_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(shutil.which('g++') is None, reason='no g++')


@pytest.fixture(scope='module')
def ha(tmp_path_factory):
    out = tmp_path_factory.mktemp('host_arith') / 'libhost_arith.so'
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-shared', '-fPIC', '-Wall', '-Werror', '-Wno-unknown-pragmas',
                           '-I' + os.path.join(ROOT, 'zkp-ecdsa_amd', 'csrc'), os.path.join(ROOT, 'tests', 'host_arith', 'host_arith.cpp'), '-o', str(out)])
    return C.CDLL(str(out))


def _field(ha, which, op, a, b):
    n = len(a)
    ab = b''.join(x.to_bytes(40, 'big') for x in a)
    bb = b''.join(x.to_bytes(40, 'big') for x in b)
    out = C.create_string_buffer(40 * n)
    assert ha.ha_field_op(which, op, C.c_uint64(n), ab, bb, out) == 0
    return [int.from_bytes(out.raw[40 * i:40 * i + 40], 'big') for i in range(n)]


def test_field_templates_on_the_host(ha):
    rnd = random.Random(11)
    for which, m in enumerate([R.p256.p, R.p256.order, R.tomEdwards256.p]):
        a = [rnd.randrange(m) for _ in range(400)] + [0, 1, m - 1, m - 1, 0, 2, m - 2, (1 << 255) % m]
        b = [rnd.randrange(m) for _ in range(400)] + [0, m - 1, m - 1, 1, m - 1, m - 2, 2, m - 1]
        assert _field(ha, which, 0, a, b) == [x * y % m for x, y in zip(a, b)]
        assert _field(ha, which, 1, a, b) == [(x + y) % m for x, y in zip(a, b)]
        assert _field(ha, which, 2, a, b) == [(x - y) % m for x, y in zip(a, b)]
        assert _field(ha, which, 3, a[:40] + a[-8:], b[:48]) == [pow(x, -1, m) if x else 0 for x in a[:40] + a[-8:]]
        assert _field(ha, which, 4, a, b) == [(x * y - x - y) % m for x, y in zip(a, b)]                       # fe_sub2
        assert _field(ha, which, 5, a, b) == [(2 * (x + y) * (x - y) - (y * y + x)) % m for x, y in zip(a, b)]  # lazy chain, K up to 10
        assert _field(ha, which, 6, a, b) == [(x + y) ** 2 % m for x, y in zip(a, b)]                           # limbs_mont_sqr (also inside every inversion)


def _tom_xy(pt):
    x, y = pt.toAffine()
    return x.to_bytes(36, 'big') + y.to_bytes(36, 'big')


def test_tom256_formulas_on_the_host(ha):
    rnd = random.Random(5)
    g, q = R.tomEdwards256, R.tomEdwards256.order
    bases = [g.generator().mul(g.newScalar(rnd.randrange(1, q))) for _ in range(6)]
    ks = [rnd.randrange(1 << 256) for _ in range(8)] + [0, 1, 2, q - 1, q, q + 1, (1 << 256) - 1, 1 << 255]
    pts, kk = [], []
    for i, k in enumerate(ks):
        pts.append(bases[i % len(bases)])
        kk.append(k)
    n = len(pts)
    out = C.create_string_buffer(72 * n)
    assert ha.ha_tom_mul(C.c_uint64(n), b''.join(_tom_xy(p) for p in pts), b''.join(k.to_bytes(32, 'big') for k in kk), out) == 0
    for i in range(n):
        exp = pts[i].mul(g.newScalar(kk[i]))
        assert out.raw[72 * i:72 * i + 72] == _tom_xy(exp), (i, hex(kk[i]))
    # comb entry forms: P + Q - R via from_niels / add_niels / negated entry + add_niels_last, incl. P = Q, R = P + Q (-> identity)
    P = [bases[0], bases[1], bases[2], bases[3], bases[0]]
    Q = [bases[1], bases[1], bases[3], bases[4], bases[5]]
    Rr = [bases[2], bases[5], bases[2].add(bases[3]), bases[3], bases[5]]
    m = len(P)
    out = C.create_string_buffer(72 * m)
    assert ha.ha_tom_combo(C.c_uint64(m), b''.join(map(_tom_xy, P)), b''.join(map(_tom_xy, Q)), b''.join(map(_tom_xy, Rr)), out) == 0
    for i in range(m):
        assert out.raw[72 * i:72 * i + 72] == _tom_xy(P[i].add(Q[i]).sub(Rr[i])), i
    # a point off the curve and a coordinate >= t are refused by tom_from_affine_words (edwards.ts:52-65, 74-77)
    bad = bytearray(_tom_xy(bases[0]))
    bad[40] ^= 1
    big = (g.p + 1).to_bytes(36, 'big') + (5).to_bytes(36, 'big')
    out = C.create_string_buffer(144)
    assert ha.ha_tom_mul(C.c_uint64(2), bytes(bad) + big, (3).to_bytes(32, 'big') * 2, out) == 2


def test_tom256_plain_domain_curve_check_on_the_host(ha):
    """tom_words_on_curve (five products on plain coordinates, what the verifier's validation pass runs) agrees with the loader's check and with the
    reference's equation (edwards.ts:52-65) on points, near-points, non-canonical coordinates and the special points."""
    rnd = random.Random(17)
    g, q, t = R.tomEdwards256, R.tomEdwards256.order, R.tomEdwards256.p
    cases = []
    for _ in range(40):
        x, y = g.generator().mul(g.newScalar(rnd.randrange(1, q))).toAffine()
        cases += [(x, y), (x ^ 1, y), (x, y ^ (1 << rnd.randrange(256))), (t - x, y), (x, t - y), (y, x)]
    cases += [(0, 1), (0, t - 1), (0, 0), (1, 0), (0, t), (t, 1), (0, 1 + t), (rnd.randrange(t), rnd.randrange(t)), ((1 << 288) - 1, 1)]
    out = C.create_string_buffer(len(cases))
    assert ha.ha_tom_on_curve(C.c_uint64(len(cases)), b''.join(x.to_bytes(36, 'big') + y.to_bytes(36, 'big') for x, y in cases), out) == 0
    a, d = g.a, g.d
    for (x, y), got in zip(cases, out.raw):
        exp = x < t and y < t and (a * x * x + y * y - 1 - d * x * x * y * y) % t == 0
        assert got == (3 if exp else 0), (hex(x), hex(y), got)
    assert sum(out.raw) >= 3 * 80   # the sweep holds real points, not only rejects


def _p_xy(pt):
    c = pt.toAffine()
    return bytes(64) if not c else c[0].to_bytes(32, 'big') + c[1].to_bytes(32, 'big')


def test_p256_complete_formulas_on_the_host(ha):
    rnd = random.Random(8)
    g, n = R.p256, R.p256.order
    bases = [g.generator().mul(g.newScalar(rnd.randrange(1, n))) for _ in range(5)] + [g.generator()]
    ks = [rnd.randrange(1 << 256) for _ in range(8)] + [0, 1, 2, n - 1, n, n + 1, (1 << 256) - 1]
    pts = [bases[i % len(bases)] for i in range(len(ks))]
    cnt = len(ks)
    out = C.create_string_buffer(64 * cnt)
    assert ha.ha_p256_mul(C.c_uint64(cnt), b''.join(_p_xy(p) for p in pts), b''.join(k.to_bytes(32, 'big') for k in ks), out) == 0
    for i in range(cnt):
        assert out.raw[64 * i:64 * i + 64] == _p_xy(pts[i].mul(g.newScalar(ks[i]))), (i, hex(ks[i]))
    # RFC 6979 A.2.5 public key = x * G for the published private key: a public vector through the host build
    d = 0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721
    out = C.create_string_buffer(64)
    assert ha.ha_p256_mul(C.c_uint64(1), _p_xy(g.generator()), d.to_bytes(32, 'big'), out) == 0
    assert out.raw.hex().upper() == ('60FED4BA255A9D31C961EB74C6356D68C049B8923B61FA6CE669622E60F29FB6'
                                     '7903FE1008B8BC99A41AE9E95628BC64F2F1B20C2D7E9F5177A3C294D4462299')
    # complete addition: generic, doubling (P = Q) and inverse (P = -Q -> identity = 64 zero bytes)
    A = [bases[0], bases[1], bases[2], bases[3]]
    B = [bases[1], bases[1], bases[2].neg(), bases[5]]
    out = C.create_string_buffer(64 * 4)
    assert ha.ha_p256_add(C.c_uint64(4), b''.join(map(_p_xy, A)), b''.join(map(_p_xy, B)), out) == 0
    for i in range(4):
        assert out.raw[64 * i:64 * i + 64] == _p_xy(A[i].add(B[i])), i
    bad = bytearray(_p_xy(bases[0]))
    bad[5] ^= 4
    assert ha.ha_p256_mul(C.c_uint64(1), bytes(bad), (7).to_bytes(32, 'big'), C.create_string_buffer(64)) == 1


def test_p256_jacobian_doubling_chain_on_the_host(ha):
    """p256_jdbl (3M + 5S, k_rtab_base's chain of 256 doublings) and p256_from_jac against the oracle's 2^k * P; the identity stays the identity."""
    rnd = random.Random(21)
    g, n = R.p256, R.p256.order
    pts = [g.generator().mul(g.newScalar(rnd.randrange(1, n))) for _ in range(6)] + [g.generator()]
    cnt = len(pts)
    for nd in (0, 1, 2, 4, 5, 64, 255, 256, 260):
        out = C.create_string_buffer(64 * cnt)
        assert ha.ha_p256_jdbl_chain(C.c_uint64(cnt), b''.join(_p_xy(p) for p in pts), nd, 0, out) == 0
        for i in range(cnt):
            assert out.raw[64 * i:64 * i + 64] == _p_xy(pts[i].mul(g.newScalar(pow(2, nd, n)))), (nd, i)
    out = C.create_string_buffer(64)
    assert ha.ha_p256_jdbl_chain(C.c_uint64(1), _p_xy(pts[0]), 9, 1, out) == 0 and out.raw == bytes(64)


def test_sha256_byte_absorber_on_the_host(ha):
    import hashlib
    for ln in (0, 1, 3, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 268, 603, 1000):
        msgs = [bytes((i * 7 + j * 13 + ln) & 255 for j in range(ln)) for i in range(9)]
        out = C.create_string_buffer(32 * len(msgs))
        assert ha.ha_sha256(C.c_uint64(len(msgs)), C.c_uint64(ln), b''.join(msgs), out) == 0
        assert [out.raw[32 * i:32 * i + 32] for i in range(len(msgs))] == [hashlib.sha256(m).digest() for m in msgs], ln
    out = C.create_string_buffer(32)
    ha.ha_sha256(C.c_uint64(1), C.c_uint64(3), b'abc', out)    # FIPS 180-4 B.1
    assert out.raw.hex() == 'ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad'


def test_sha256_word_path_at_every_alignment_on_the_host(ha):
    """put_be<32|33|36> -> put_word: whole words into a stream that is not word-aligned (hashPoints: 0x04 || X || Y with 33-byte
    Tom coordinates), across block boundaries, for every number of leading bytes mod 4."""
    import hashlib
    for nbytes in (32, 33, 36):
        for lead in range(0, 9):
            for count in (1, 2, 3, 7, 20):
                for trail in (0, 1, 5):
                    lb = bytes((3 * i + lead) & 255 for i in range(lead))
                    vals = bytes((i * 11 + nbytes * 5 + count) & 255 for i in range(nbytes * count))
                    tb = bytes((7 * i + 1) & 255 for i in range(trail))
                    out = C.create_string_buffer(32)
                    assert ha.ha_sha256_values(C.c_uint64(lead), lb, C.c_uint64(count), nbytes, vals, C.c_uint64(trail), tb, out) == 0
                    assert out.raw == hashlib.sha256(lb + vals + tb).digest(), (nbytes, lead, count, trail)


def _reference_draws(fills, sec, ndraws):
    """rnd() of big.ts:171-181 over the fill sequence: draw j uses modulus n or q and consumes
    fills until one is below it.  Returns the accepted 32-byte fills."""
    n, q = R.p256.order, R.p256.p
    out, f = [], 0
    for j in range(ndraws):
        is_n = j == 0 or (3 <= j < 3 + 4 * sec and ((j - 3) & 3) < 2)
        m = n if is_n else q
        while int.from_bytes(fills[f], 'big') >= m:
            f += 1
        out.append(fills[f])
        f += 1
    return out


def test_rng_draw_mapping_on_the_host(ha):
    """rng_map: logical draw k -> fill index, given the per-proof list of suspicious fills (first word 0xffffffff) the prepass
    leaves; explicit streams with planted values >= n, in [n, q) and >= q at n-draws and q-draws; seed mode against SHA-256."""
    import hashlib
    rnd = random.Random(3)
    n, q = R.p256.order, R.p256.p
    sec, nblk, ndraws, B = 20, 140, 120, 6
    streams, excs = [], []
    for p in range(B):
        fills = [bytes([rnd.randrange(255)]) + bytes(rnd.randrange(256) for _ in range(31)) for _ in range(nblk)]   # first byte < 0xff: accepted
        plant = {0: [], 1: [(0, n + 5)], 2: [(2, q + 1), (3, (1 << 256) - 1)], 3: [(7, n + 99), (8, n + 7), (11, q + 3)],
                 4: [(1, n + 1), (5, (1 << 256) - 2), (6, n + 2), (40, q + 9), (41, n + 10)], 5: [(3, n + 12345), (4, n + 1), (90, q)]}[p]
        for idx, val in plant:
            fills[idx] = val.to_bytes(32, 'big')
        streams.append(fills)
        e = [(i, (1 if int.from_bytes(f, 'big') >= n else 0) | (2 if int.from_bytes(f, 'big') >= q else 0))
             for i, f in enumerate(fills) if f[:4] == b'\xff\xff\xff\xff']
        rnd.shuffle(e)                      # the prepass appends in arbitrary order
        excs.append(e)
    idx = (C.c_uint32 * (8 * B))()
    fl = (C.c_uint32 * (8 * B))()
    cnt = (C.c_uint32 * B)()
    for p, e in enumerate(excs):
        cnt[p] = len(e)
        for i, (a, b) in enumerate(e):
            idx[8 * p + i], fl[8 * p + i] = a, b
    data = b''.join(b''.join(s) for s in streams)
    out = C.create_string_buffer(32 * B * ndraws)
    assert ha.ha_rng_draws(1, sec, C.c_uint64(B), data, C.c_uint64(nblk), idx, fl, cnt, 0, ndraws, out) == 0
    for p in range(B):
        exp = _reference_draws(streams[p], sec, ndraws)
        got = [out.raw[32 * (p * ndraws + k):32 * (p * ndraws + k) + 32] for k in range(ndraws)]
        assert got == exp, p
    # seed mode: fill k = SHA-256(seed || be64(k)); no suspicious fills for these seeds
    seeds = [hashlib.sha256(b'seed%d' % i).digest() for i in range(4)]
    zero = (C.c_uint32 * 4)()
    out = C.create_string_buffer(32 * 4 * 10)
    assert ha.ha_rng_draws(0, sec, C.c_uint64(4), b''.join(seeds), C.c_uint64(0), idx, fl, zero, 5, 10, out) == 0
    for p in range(4):
        for k in range(10):
            assert out.raw[32 * (p * 10 + k):32 * (p * 10 + k) + 32] == hashlib.sha256(seeds[p] + (5 + k).to_bytes(8, 'big')).digest()


def test_comb_digit_recoding_reproduces_the_scalar(ha):
    """comb_digits.h: for every supported table width (8..24 unsigned, 25 and 26 signed) the digits (index, sign) of a 256-bit
    scalar sum back to it, indices stay inside the table (2^W entries, or 2^(W-1) + 1 for signed digits) and nothing is left over."""
    rnd = random.Random(17)
    ha.ha_comb_digits.restype = C.c_uint32
    specials = [0, 1, (1 << 256) - 1, 1 << 255, (1 << 255) - 1, R.p256.p - 1, int('55' * 32, 16), int('aa' * 32, 16)]
    for bits in range(8, 27):
        signed = bits > 24
        nwin = ((257 if signed else 256) + bits - 1) // bits          # engine.h: tom_nwin
        entries = (1 << (bits - 1)) + 1 if signed else 1 << bits         # engine.h: tom_win_entries
        for k in specials + [rnd.randrange(1 << 256) for _ in range(40)] + [((1 << bits) - 1) << (bits * j) & ((1 << 256) - 1) for j in range(3)]:
            idx = (C.c_uint32 * nwin)()
            neg = (C.c_uint8 * nwin)()
            rest = ha.ha_comb_digits(bits, nwin, k.to_bytes(32, 'big'), idx, neg)
            assert rest == 0, (bits, hex(k))
            assert all(i < entries for i in idx), (bits, hex(k))
            assert signed or not any(neg)
            assert sum((-int(i) if s else int(i)) << (bits * j) for j, (i, s) in enumerate(zip(idx, neg))) == k, (bits, hex(k))


def test_key_table_digits_on_the_host(ha):
    """comb_digits.h, KeyDigits: the signed 8-bit digits the per-key tables are indexed with (ktab.h) sum back to the scalar, stay within
    the 128 stored multiples, and leave nothing behind after 33 windows -- including the carry into the 33rd."""
    rnd = random.Random(23)
    ha.ha_key_digits.restype = C.c_uint32
    specials = [0, 1, 128, 129, 255, 256, (1 << 256) - 1, 1 << 255, (1 << 255) - 1, R.p256.order - 1, int('81' * 32, 16), int('80' * 32, 16), int('ff' * 31 + '81', 16)]
    for k in specials + [rnd.randrange(1 << 256) for _ in range(300)]:
        dig = (C.c_uint32 * 33)()
        neg = (C.c_uint8 * 33)()
        assert ha.ha_key_digits(k.to_bytes(32, 'big'), dig, neg) == 0, hex(k)
        assert all(d <= 128 for d in dig) and dig[32] <= 1 and not neg[32], hex(k)
        assert sum((-int(d) if s else int(d)) << (8 * j) for j, (d, s) in enumerate(zip(dig, neg))) == k, hex(k)


def test_key_table_multiplication_on_the_host(ha):
    """ktab.h: a key's table built like k_ktab.hip builds it and acc + k * (+-P) through p256_ktab_mul_acc (33 gathered entries, signed
    digits, entries negated on their words when the prover's key is the table's other root) against the oracle's Point.mul."""
    rnd = random.Random(31)
    g, n = R.p256, R.p256.order
    P = g.generator().mul(g.newScalar(rnd.randrange(1, n)))
    S = g.generator().mul(g.newScalar(rnd.randrange(1, n)))
    ks = [rnd.randrange(1 << 256) for _ in range(12)] + [0, 1, 127, 128, 129, 255, 256, n - 1, n, (1 << 256) - 1, int('80' * 32, 16), int('81' * 32, 16)]
    negs = bytes(i & 1 for i in range(len(ks)))
    for start in (None, S):
        out = C.create_string_buffer(64 * len(ks))
        assert ha.ha_ktab_mul(_p_xy(P), bytes(64) if start is None else _p_xy(start), C.c_uint64(len(ks)), b''.join(k.to_bytes(32, 'big') for k in ks), negs, out) == 0
        for i, k in enumerate(ks):
            want = (P.neg() if negs[i] else P).mul(g.newScalar(k % n))
            if start is not None:
                want = want.add(start)
            assert out.raw[64 * i:64 * i + 64] == _p_xy(want), (i, hex(k), negs[i])


def test_divsteps_inversion_on_the_host(ha):
    """field.h: fe_inv_gcd (Bernstein-Yang divsteps in the radix-2^30 limbs) against pow(x, -1, m); inv(0) = 0 like big.ts:113-119."""
    rnd = random.Random(23)
    for which, m in enumerate([R.p256.p, R.p256.order, R.tomEdwards256.p]):
        a = [rnd.randrange(m) for _ in range(600)] + [0, 1, 2, 3, m - 1, m - 2, (m + 1) // 2, (1 << 255) % m, (1 << 30) - 1, 1 << 30, (1 << 240) % m]
        a += [rnd.randrange(1 << k) for k in (1, 8, 29, 30, 31, 60, 61, 200, 240, 241)]
        assert _field(ha, which, 7, a, a) == [pow(x, -1, m) if x else 0 for x in a]


def _co_field(ha, which, op, a, b):
    n = len(a)
    assert n % 4 == 0
    ab = b''.join(x.to_bytes(40, 'big') for x in a)
    bb = b''.join(x.to_bytes(40, 'big') for x in b)
    out = C.create_string_buffer(40 * n)
    assert ha.ha_co_field_op(which, op, C.c_uint64(n), ab, bb, out) == 0
    return [int.from_bytes(out.raw[40 * i:40 * i + 40], 'big') for i in range(n)]


def test_lane_cooperative_field_arithmetic_on_the_host(ha):
    """coop.h (one element per 16-lane row, one limb per lane; DPP broadcasts and shifts emulated lane by lane): the Montgomery product, add / sub with the
    parallel carry step, a lazy chain at large magnitudes and a Fermat power, four elements per wave, against Python integers."""
    rnd = random.Random(29)
    for which, m in enumerate([R.p256.p, R.p256.order, R.tomEdwards256.p]):
        a = [rnd.randrange(m) for _ in range(200)] + [0, 1, m - 1, m - 1, 0, 2, m - 2, (1 << 255) % m]
        b = [rnd.randrange(m) for _ in range(200)] + [0, m - 1, m - 1, 1, m - 1, m - 2, 2, m - 1]
        # limbs of all ones / carries that ripple through every limb
        a += [(1 << 240) - 1, (1 << 256) % m, m - 1, ((1 << 30) - 1) << 30]
        b += [1, m - 1, 1, (1 << 210) + 1]
        assert _co_field(ha, which, 0, a, b) == [x * y % m for x, y in zip(a, b)]
        assert _co_field(ha, which, 1, a, b) == [(x + y) % m for x, y in zip(a, b)]
        assert _co_field(ha, which, 2, a, b) == [(x - y) % m for x, y in zip(a, b)]
        assert _co_field(ha, which, 3, a, b) == [(2 * (x + y) * (x - y) - (y * y + x)) % m for x, y in zip(a, b)]
        assert _co_field(ha, which, 4, a[:8] + a[-8:], b[:16]) == [pow(x, m - 2, m) for x in a[:8] + a[-8:]]


def test_lane_cooperative_tom256_formulas_on_the_host(ha):
    """co_tom_dbl / co_tom_add (rows X, Y, T, Z; ds_bpermute row moves emulated) in a double-and-add against the oracle's scalar multiplication."""
    rnd = random.Random(31)
    g, q = R.tomEdwards256, R.tomEdwards256.order
    bases = [g.generator().mul(g.newScalar(rnd.randrange(1, q))) for _ in range(3)]
    ks = [rnd.randrange(1 << 256) for _ in range(3)] + [0, 1, 2, q - 1, q, q + 1]
    pts = [bases[i % len(bases)] for i in range(len(ks))]
    n = len(pts)
    out = C.create_string_buffer(72 * n)
    assert ha.ha_co_tom_mul(C.c_uint64(n), b''.join(_tom_xy(p) for p in pts), b''.join(k.to_bytes(32, 'big') for k in ks), out) == 0
    for i in range(n):
        assert out.raw[72 * i:72 * i + 72] == _tom_xy(pts[i].mul(g.newScalar(ks[i]))), (i, hex(ks[i]))


def _p256_xy(pt):
    x, y = pt.toAffine()
    return x.to_bytes(32, 'big') + y.to_bytes(32, 'big')


def test_lane_cooperative_p256_formulas_on_the_host(ha):
    """co_p256_dbl / co_p256_add (the complete laws of weier.ts:133-230 in four passes of four rows) and the Jacobian-with-ZZ doubling chain of k_rtab_base,
    against the oracle: scalar multiples from the identity (O + P, P + P, P + O on the way), P + (-P), and 2^n P incl. the identity."""
    rnd = random.Random(37)
    g, q = R.p256, R.p256.order
    bases = [g.generator().mul(g.newScalar(rnd.randrange(1, q))) for _ in range(3)]
    ks = [rnd.randrange(1 << 256) for _ in range(3)] + [1, 2, 3, q - 1, q + 1]
    pts = [bases[i % len(bases)] for i in range(len(ks))]
    n = len(pts)
    out = C.create_string_buffer(64 * n)
    assert ha.ha_co_p256_mul(C.c_uint64(n), b''.join(_p256_xy(p) for p in pts), b''.join(k.to_bytes(32, 'big') for k in ks), out) == 0
    for i in range(n):
        assert out.raw[64 * i:64 * i + 64] == _p256_xy(pts[i].mul(g.newScalar(ks[i]))), (i, hex(ks[i]))
    # k = q: the identity (64 zero bytes from the harness's store)
    out = C.create_string_buffer(64)
    assert ha.ha_co_p256_mul(C.c_uint64(1), _p256_xy(bases[0]), q.to_bytes(32, 'big'), out) == 0
    assert out.raw == bytes(64)
    P = [bases[0], bases[0], bases[1], bases[2]]
    Q = [bases[0], bases[0].neg(), bases[2], bases[1].add(bases[2]).neg()]
    out = C.create_string_buffer(64 * len(P))
    assert ha.ha_co_p256_add(C.c_uint64(len(P)), b''.join(map(_p256_xy, P)), b''.join(map(_p256_xy, Q)), out) == 0
    for i in range(len(P)):
        s = P[i].add(Q[i])
        assert out.raw[64 * i:64 * i + 64] == (bytes(64) if s.isIdentity() else _p256_xy(s)), i
    for nd in (1, 2, 13, 256):
        out = C.create_string_buffer(64 * 3)
        assert ha.ha_co_p256_jdbl_chain(C.c_uint64(3), b''.join(map(_p256_xy, bases)), nd, 0, out) == 0
        for i in range(3):
            assert out.raw[64 * i:64 * i + 64] == _p256_xy(bases[i].mul(g.newScalar(pow(2, nd, q)))), (nd, i)
    out = C.create_string_buffer(64)
    assert ha.ha_co_p256_jdbl_chain(C.c_uint64(1), _p256_xy(bases[0]), 40, 1, out) == 0
    assert out.raw == bytes(64)
