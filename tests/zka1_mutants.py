"""Seeded, stratified mutants of honest ZKA1 proofs (include/zkattest.h layout) for the differential tests of the verifier:
tests/test_gpu_mutants.py (-m gpu: engine vs oracle) and tests/test_oracle_mutants.py (CPU tier: the sweep reaches every outcome)."""
import random

P256_N = 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551
TOM_Q = 0xffffffff00000001000000000000000000000000ffffffffffffffffffffffff   # Tom-256 scalar field = the P-256 base field
REP_HEAD, PADD, MULT, EQ = 336, 3392, 656, 240


class Layout:
    """Byte offsets of every field of one honest ZKA1 proof (include/zkattest.h)."""

    def __init__(self, proof, n, sec=80):
        self.n, self.sec = n, sec
        self.bits = int.from_bytes(proof[16:32], 'big')
        self.rep = []
        off = 304
        for i in range(sec):
            self.rep.append(off)
            off += REP_HEAD + (0 if (self.bits >> i) & 1 else PADD)
        self.gk = off
        assert off + n * (4 * 72 + 96) + 32 == len(proof)
        self.zero_reps = [i for i in range(sec) if not (self.bits >> i) & 1]
        self.one_reps = [i for i in range(sec) if (self.bits >> i) & 1]

    def top_points(self):   # (offset, size)
        return [(32, 64), (96, 64), (160, 72), (232, 72)]

    def rep_points(self, i):
        r = self.rep[i]
        return [(r, 64), (r + 64, 72), (r + 136, 72)]

    def rep_scalars(self, i):
        return [self.rep[i] + 208 + 32 * k for k in range(4)]

    def padd_points(self, i):
        a = self.rep[i] + REP_HEAD
        pts = [a + 72 * k for k in range(4)]
        for m in range(4):
            pts += [a + 288 + MULT * m + 72 * k for k in range(6)]
        for e in range(2):
            pts += [a + 288 + 4 * MULT + EQ * e + 72 * k for k in range(2)]
        return pts

    def padd_scalars(self, i):
        a = self.rep[i] + REP_HEAD
        sc = []
        for m in range(4):
            sc += [a + 288 + MULT * m + 432 + 32 * k for k in range(7)]
        for e in range(2):
            sc += [a + 288 + 4 * MULT + EQ * e + 144 + 32 * k for k in range(3)]
        return sc

    def gk_points(self):
        return [self.gk + 72 * k for k in range(4 * self.n)]

    def gk_scalars(self):
        return [self.gk + 288 * self.n + 32 * k for k in range(3 * self.n + 1)]


def _flip(b, pos, bit):
    m = bytearray(b)
    m[pos] ^= bit
    return bytes(m)


def _put(b, pos, data):
    m = bytearray(b)
    m[pos:pos + len(data)] = data
    return bytes(m)


def _swap(b, p0, p1, size):
    m = bytearray(b)
    m[p0:p0 + size], m[p1:p1 + size] = m[p1:p1 + size], m[p0:p0 + size]
    return bytes(m)


def _set_len(b):
    return b[:4] + len(b).to_bytes(4, 'big') + b[8:]


def mutants(proofs, n, S, synth_S):
    """[(name, message index, bytes)]: seeded, stratified mutants of the honest proofs `proofs` (ring of 2^n keys)."""
    rnd = random.Random(S)
    out = []
    L = [Layout(p, n) for p in proofs]
    np_ = len(proofs)

    def add(name, j, b):
        out.append(('%s/p%d' % (name, j), j, b))

    def rbit():
        return 1 << rnd.randrange(8)

    for j in range(np_):
        add('honest', j, proofs[j])
    # 1. header, byte by byte
    for pos in range(32):
        j = pos % np_
        add('hdr-byte%d' % pos, j, _flip(proofs[j], pos, rbit()))
    # 2. the four top-level points
    for j in range(np_):
        for (o, sz) in L[j].top_points():
            for _ in range(2):
                add('top@%d' % o, j, _flip(proofs[j], o + rnd.randrange(sz), rbit()))
    # 3. A / Tx / Ty of repetitions
    for _ in range(36):
        j = rnd.randrange(np_)
        i = rnd.randrange(80)
        o, sz = rnd.choice(L[j].rep_points(i))
        add('rep%d-point' % i, j, _flip(proofs[j], o + rnd.randrange(sz), rbit()))
    # 4. response scalars of repetitions of both kinds
    for _ in range(52):
        j = rnd.randrange(np_)
        i = rnd.randrange(80)
        o = rnd.choice(L[j].rep_scalars(i))
        add('rep%d-scalar' % i, j, _flip(proofs[j], o + rnd.randrange(32), rbit()))
    # 5. every kind of PointAdd sub-proof field
    for _ in range(48):
        j = rnd.randrange(np_)
        i = rnd.choice(L[j].zero_reps)
        k = rnd.randrange(32)
        add('rep%d-padd-point%d' % (i, k), j, _flip(proofs[j], L[j].padd_points(i)[k] + rnd.randrange(72), rbit()))
    for _ in range(48):
        j = rnd.randrange(np_)
        i = rnd.choice(L[j].zero_reps)
        k = rnd.randrange(34)
        add('rep%d-padd-scalar%d' % (i, k), j, _flip(proofs[j], L[j].padd_scalars(i)[k] + rnd.randrange(32), rbit()))
    # 6. Groth-Kohlweiss points and scalars
    for _ in range(28):
        j = rnd.randrange(np_)
        k = rnd.randrange(4 * n)
        add('gk-point%d' % k, j, _flip(proofs[j], L[j].gk_points()[k] + rnd.randrange(72), rbit()))
    for _ in range(28):
        j = rnd.randrange(np_)
        k = rnd.randrange(3 * n + 1)
        add('gk-scalar%d' % k, j, _flip(proofs[j], L[j].gk_scalars()[k] + rnd.randrange(32), rbit()))
    # 7. truncations and extensions (multiples of 4 bytes: ZKA1 proofs are packed 4-byte aligned), with and without the length fixed up
    for cut in (4, 32, 96, 336, 3392, 3728):
        j = rnd.randrange(np_)
        add('cut%d' % cut, j, proofs[j][:-cut])
        add('cut%d-len' % cut, j, _set_len(proofs[j][:-cut]))
    for ext in (4, 32, 72):
        j = rnd.randrange(np_)
        add('ext%d' % ext, j, proofs[j] + bytes(ext))
        add('ext%d-len' % ext, j, _set_len(proofs[j] + bytes(ext)))
    add('only-header', 0, proofs[0][:32])
    add('only-header-len', 0, _set_len(proofs[0][:32]))
    add('cut-inside-gk', 0, _set_len(proofs[0][:L[0].gk + 72]))
    # 8. challenge-bit field: one bit either way, two bits (one of each: the size stays), bits above secLevel
    for j in range(np_):
        z, o = rnd.choice(L[j].zero_reps), rnd.choice(L[j].one_reps)
        for name, bits in (('bits-0to1', [z]), ('bits-1to0', [o]), ('bits-swap', [z, o]), ('bits-above-sec', [80 + rnd.randrange(48)])):
            v = L[j].bits
            for b in bits:
                v ^= 1 << b
            add(name, j, _put(proofs[j], 16, v.to_bytes(16, 'big')))
    # 9. n and secLevel of the header
    for v in (n - 1, n + 1, 0, 63, 64, 255):
        add('hdr-n=%d' % v, 0, _put(proofs[0], 12, v.to_bytes(4, 'big')))
    for v in (79, 81, 19, 0, 128, 255):
        add('hdr-sec=%d' % v, 0, _put(proofs[0], 8, v.to_bytes(4, 'big')))
    # a structurally complete proof whose GKProof has another length: "return false" (gk.ts:208-218) -- unless one of its points is bad
    gk_sz = 4 * 72 + 96
    pts, scs = L[0].gk_points(), L[0].gk_scalars()
    for n2 in (n - 1, n + 1):
        m = min(n, n2)
        body = b''
        for grp in range(4):
            for k in range(n2):
                body += proofs[0][pts[grp * n + min(k, n - 1)]:][:72]
        for grp in range(3):
            for k in range(n2):
                body += proofs[0][scs[grp * n + min(k, n - 1)]:][:32]
        body += proofs[0][scs[3 * n]:][:32]
        assert len(body) == n2 * gk_sz + 32 and m > 0
        full = _put(_set_len(proofs[0][:L[0].gk] + body), 12, n2.to_bytes(4, 'big'))
        add('gk-of-n=%d' % n2, 0, full)
        add('gk-of-n=%d-badpoint' % n2, 0, _flip(full, L[0].gk + 72 * (4 * n2 - 1) + 40, 2))
        add('gk-of-n=%d-badrep' % n2, 0, _flip(full, L[0].rep[3] + 70, 2))
    # 10. a valid point swapped for another valid point
    for j in range(np_):
        lj = L[j]
        i0, i1 = rnd.sample(range(80), 2)
        add('swap-Tx-Ty', j, _swap(proofs[j], lj.rep[i0] + 64, lj.rep[i0] + 136, 72))
        add('swap-A-A', j, _swap(proofs[j], lj.rep[i0], lj.rep[i1], 64))
        add('swap-kx-ky', j, _swap(proofs[j], 160, 232, 72))
        add('swap-R-comS1', j, _swap(proofs[j], 32, 96, 64))
        g = lj.gk_points()
        add('swap-cl0-ca0', j, _swap(proofs[j], g[0], g[n], 72))
        add('swap-cb-cd', j, _swap(proofs[j], g[2 * n + rnd.randrange(n)], g[3 * n + rnd.randrange(n)], 72))
        z = rnd.choice(lj.zero_reps)
        pp = lj.padd_points(z)
        a, b = rnd.sample(range(32), 2)
        add('swap-padd-points', j, _swap(proofs[j], pp[a], pp[b], 72))
        add('swap-C8-kx', j, _swap(proofs[j], pp[0], 160, 72))
    # 11. scalars replaced by 0, by the group order (reduces to 0), by 2^256 - 1
    for j in range(np_):
        lj = L[j]
        for name, val in (('zero', bytes(32)), ('order-n', P256_N.to_bytes(32, 'big')), ('order-q', TOM_Q.to_bytes(32, 'big')), ('ones', b'\xff' * 32)):
            i = rnd.choice(lj.one_reps)
            add('alpha=%s-rep%d' % (name, i), j, _put(proofs[j], lj.rep[i] + 208, val))       # alpha = 0: 'T is at infinity' if sampled
            i = rnd.choice(lj.zero_reps)
            add('z=%s-rep%d' % (name, i), j, _put(proofs[j], lj.rep[i] + 208, val))
            add('gk-f=%s' % name, j, _put(proofs[j], lj.gk_scalars()[rnd.randrange(n)], val))
            add('zd=%s' % name, j, _put(proofs[j], lj.gk_scalars()[3 * n], val))
    # 12. non-canonical encodings: Tom coordinate + t (fits 36 bytes), non-zero padding byte; P-256 point negated (valid), Tom negated
    import zkattest_ref as R
    T, P = R.tomEdwards256.p, R.p256.p
    for j in range(np_):
        lj = L[j]
        i = rnd.randrange(80)
        x = int.from_bytes(proofs[j][lj.rep[i] + 64:][:36], 'big')
        add('Tx+t', j, _put(proofs[j], lj.rep[i] + 64, (x + T).to_bytes(36, 'big')))
        add('Tx-padbyte', j, _flip(proofs[j], lj.rep[i] + 64, 1))
        add('kx-padbyte', j, _flip(proofs[j], 160 + 36 + 2, 0x80))
        y = int.from_bytes(proofs[j][lj.rep[i] + 32:][:32], 'big')
        add('A-negated', j, _put(proofs[j], lj.rep[i] + 32, (P - y).to_bytes(32, 'big')))
        x = int.from_bytes(proofs[j][lj.gk:][:36], 'big')
        add('cl0-negated', j, _put(proofs[j], lj.gk, (T - x).to_bytes(36, 'big')))
        y = int.from_bytes(proofs[j][64:96], 'big')
        add('R-negated', j, _put(proofs[j], 64, (P - y).to_bytes(32, 'big')))
        if y + P < 1 << 256:   # a coordinate in [p, 2^256): accepted where deserializePoint accepts it (weier.ts:74-89), hashed reduced
            add('R-y+p', j, _put(proofs[j], 64, (y + P).to_bytes(32, 'big')))
    # 13. the first exception in SAMPLED order wins (exp.ts:265-346): alpha = 0 next to a changed challenge ('params not found' at every
    # repetition whose recomputed bit differs), several carriers so that the three seeds see both orders
    for j in range(np_):
        lj = L[j]
        for _ in range(8):
            i = rnd.choice(lj.one_reps)
            k = rnd.randrange(80)
            m = _put(proofs[j], lj.rep[i] + 208, bytes(32))
            # Ty of rep k replaced by Tx of rep k (a valid point): the Exp challenge changes, the layout does not
            add('alpha0-rep%d+Ty%d=Tx%d' % (i, k, k), j, _put(m, lj.rep[k] + 136, proofs[j][lj.rep[k] + 64:][:72]))
        for _ in range(4):
            k = rnd.randrange(80)
            add('Ty%d=Tx%d' % (k, k), j, _put(proofs[j], lj.rep[k] + 136, proofs[j][lj.rep[k] + 64:][:72]))
    # 14. 'T1 is at infinity' (exp.ts:312): z = -z1 / k makes z R + Q the identity (R = k G, Q = z1 G; the synthetic nonce k is known)
    for j in range(np_):
        lj = L[j]
        msgh, sig, pkb, which, d, seed = R.synth_proof_input(synth_S, j, 1 << 30)
        k = R.fromBytes(R.synth_tag(b'nonce', synth_S, j)) % (P256_N - 1) + 1
        r = int.from_bytes(sig[:32], 'big')
        z1 = pow(r, -1, P256_N) * (int.from_bytes(msgh, 'big') % P256_N) % P256_N
        for sign in (1, -1):   # R = +-k G depending on the low-s normalisation of the signer
            zbad = (-sign * z1 * pow(k, -1, P256_N)) % P256_N
            for i in rnd.sample(lj.zero_reps, 3):
                add('T1inf(%+d)-rep%d' % (sign, i), j, _put(proofs[j], lj.rep[i] + 208, zbad.to_bytes(32, 'big')))
    # 16. the P-256 point with x = 0, (0, sqrt(b)): a valid point whose x has no inverse mod n -- as R it makes rinv = invMod(0) = 0 (big.ts:113-119), so
    # z1 = 0 and Q = 0 * G is the identity (zkpAttestList.ts:156-163); also as comS1 and as the A of a repetition of either kind
    y0 = pow(R.p256.b, (P + 1) // 4, P)
    assert y0 * y0 % P == R.p256.b % P
    for j in range(np_):
        lj = L[j]
        for yy, tag in ((y0, '+'), (P - y0, '-')):
            pt = bytes(32) + yy.to_bytes(32, 'big')
            add('R=(0,%ssqrt b)' % tag, j, _put(proofs[j], 32, pt))
            add('comS1=(0,%ssqrt b)' % tag, j, _put(proofs[j], 96, pt))
            add('A%d=(0,%ssqrt b)' % (lj.zero_reps[0], tag), j, _put(proofs[j], lj.rep[lj.zero_reps[0]], pt))
            add('A%d=(0,%ssqrt b)' % (lj.one_reps[0], tag), j, _put(proofs[j], lj.rep[lj.one_reps[0]], pt))
    # 15. a whole repetition taken from the other proof (same kind), the whole GK proof of the other proof
    if np_ >= 2:
        for _ in range(4):
            i = rnd.choice(L[0].zero_reps)
            i2 = rnd.choice(L[1].zero_reps)
            add('rep%d<-other-rep%d' % (i, i2), 0, _put(proofs[0], L[0].rep[i], proofs[1][L[1].rep[i2]:][:REP_HEAD + PADD]))
        add('gk<-other', 0, _put(proofs[0], L[0].gk, proofs[1][L[1].gk:]))
        add('other-message', 1, proofs[0])
    return out


