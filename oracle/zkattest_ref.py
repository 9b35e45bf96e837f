"""
ORACLE (test infrastructure, NOT product code) -- Python restatement of cloudflare/zkp-ecdsa.

This file restates, line by line, the reference TypeScript algorithms on the
`proveSignatureList` / `verifySignatureList` hot path with Python big ints.  It is
the independent cross-check and golden-vector generator for the C oracle
(`oracle/zkattest_oracle.c`) and for the HIP engine.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it.

PARITY STATUS: "parity unpinned" at proof level.  The reference is TypeScript needing
Node >= 24 + tsc + typedjson; none of that exists in the build container, so the
reference cannot be executed here and its own tests hold no golden proof vectors
(all protocol tests are randomised prove->verify round trips).  What IS pinned:
  * curve constants (src/curves/instances.ts:22-54), byte encodings and the challenge
    derivation (src/curves/group.ts:221-233) are copied as data/rules;
  * the reference's KATs: 3^-1 mod 5 = 2, 7^-1 mod 41 = 6 (test/bignum/big.test.ts:19-21),
    interpolate([1,2,3],[1,2,3],401) = [0,1,0] (test/proofGK/interpolate.test.ts:19-26);
  * structural tests of test/curves/ec.test.ts, test/curves/multimult.test.ts and the
    prove->verify round trips of test/*;
  * public vectors: FIPS 180-4 SHA-256, RFC 6979 A.2.5 ECDSA P-256.
Proof-level bit-exactness is defined under the deterministic RNG contract below and is
established by this restatement, the C restatement and the JavaScript restatement
(oracle/js/zkattest_ref.js, V8 BigInt) agreeing byte for byte.

RNG contract (replaces crypto.getRandomValues, src/bignum/big.ts:171-181): the k-th
32-byte fill (k counts every fill, including rejected ones) of a proof with 32-byte
seed S returns SHA-256(S || be64(k)).  Alternatively an explicit block stream can be
injected (used to exercise the rejection path).

All `file:line` citations are into /root/reference/src.
"""
import hashlib
import math

# --------------------------------------------------------------------------------------
# bignum/big.ts
# --------------------------------------------------------------------------------------


def bitLen(n):  # big.ts:23-25
    return n.bit_length() if n > 0 else 1


def byteLen(n):  # big.ts:26-28
    return (bitLen(n) + 7) // 8


def posMod(n, p):  # big.ts:36-42 (JS % truncates toward zero; result identical to Python %)
    return n % p


def expMod(n, e, p):  # big.ts:44-59
    if e < 0:
        raise ValueError('neg expo')
    r, q, k = 1, n, e
    while k > 0:
        if k & 1:
            r = (r * q) % p
        q = (q * q) % p
        k >>= 1
    return r


def _jsdiv(a, b):
    """BigInt '/' truncates toward zero."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def extendedEuclid(X, Y):  # big.ts:80-111
    a, b, c, d, x, y = 1, 0, 0, 1, X, Y
    while y != 0:
        q = _jsdiv(x, y)
        a = a - c * q
        b = b - d * q
        x = x - q * y
        x, y = y, x
        a, c = c, a
        b, d = d, b
    return x, a, b


def invMod(t, N):  # big.ts:76-78,113-119  (invMod(0, N) == 0)
    _, inv, _ = extendedEuclid(t, N)
    if inv < 0:
        inv += N
    return inv


def toBytes(n, length):  # big.ts:121-134
    if not (length > 0 and 0 <= n < (1 << (8 * length))):
        raise ValueError("number doesn't fit in array")
    return n.to_bytes(length, 'big')


def fromBytes(a):  # big.ts:161-168
    return int.from_bytes(bytes(a), 'big')


class SeedRng:
    """RNG contract: fill k -> SHA-256(seed || be64(k)); only 32-byte fills are defined."""

    def __init__(self, seed):
        assert len(seed) == 32
        self.seed = bytes(seed)
        self.k = 0

    def fill(self, nbytes):
        assert nbytes == 32, 'RNG contract defines 32-byte fills only'
        out = hashlib.sha256(self.seed + self.k.to_bytes(8, 'big')).digest()
        self.k += 1
        return out


class StreamRng:
    """Explicit block stream (list of 32-byte blocks), for rejection-path tests."""

    def __init__(self, blocks):
        self.blocks = [bytes(b) for b in blocks]
        self.k = 0

    def fill(self, nbytes):
        assert nbytes == 32
        out = self.blocks[self.k]
        self.k += 1
        return out


class OsRng:
    """Non-contract randomness (verifier randomisers, generateIndices): any length."""

    def __init__(self, seed=b'verifier'):
        self.state = hashlib.sha256(seed).digest()
        self.k = 0

    def fill(self, nbytes):
        out = b''
        while len(out) < nbytes:
            out += hashlib.sha256(self.state + self.k.to_bytes(8, 'big')).digest()
            self.k += 1
        return out[:nbytes]


def rnd(n, rng):  # big.ts:171-181
    blen = byteLen(n)
    while True:
        ret = fromBytes(rng.fill(blen))
        if ret < n:
            return ret


def rndRange(lo, hi, rng):  # big.ts:183-185
    return rnd(hi - lo + 1, rng) + lo


# --------------------------------------------------------------------------------------
# curves/group.ts, weier.ts, edwards.ts, instances.ts
# --------------------------------------------------------------------------------------


class Scalar:  # group.ts:155-218
    __slots__ = ('group', 'k')

    def __init__(self, group, s):
        self.group = group
        self.k = posMod(s, group.order) if s else 0  # group.ts:164-167

    def base16(self):
        return format(self.k, 'x')

    def add(self, s):
        return Scalar(self.group, self.k + s.k)

    def sub(self, s):
        return Scalar(self.group, self.k - s.k)

    def mul(self, s):
        return Scalar(self.group, self.k * s.k)

    def neg(self):
        return Scalar(self.group, -self.k)

    def isZero(self):
        return self.k == 0

    def cmp(self, s):
        return -1 if self.k < s.k else (1 if self.k > s.k else 0)

    def eq(self, s):
        return self.group is s.group and self.k == s.k


_DIGITS = '0123456789abcdef'


class Point:
    def sub(self, pt):  # group.ts:94-96
        return self.add(pt.neg())

    def dblmul(self, s1, p2, s2):  # group.ts:97-132
        mult1, mult2 = {}, {}
        curr1, curr2 = self.group.identity(), p2.group.identity()
        for digit in _DIGITS:
            mult1[digit] = curr1
            mult2[digit] = curr2
            curr1 = curr1.add(self)
            curr2 = curr2.add(p2)
        k1, k2 = s1.base16(), s2.base16()
        if len(k1) < len(k2):
            k1 = k1.rjust(len(k2), '0')
        if len(k2) < len(k1):
            k2 = k2.rjust(len(k1), '0')
        q = self.group.identity()
        for i in range(len(k1)):
            q = q.dbl().dbl().dbl().dbl()
            q = q.add(mult1[k1[i]])
            q = q.add(mult2[k2[i]])
        return q

    def mul(self, s):  # group.ts:133-152
        k = s.base16()
        q = self.group.identity()
        mults = {}
        curr = self.group.identity()
        for digit in _DIGITS:
            mults[digit] = curr
            curr = curr.add(self)
        for ki in k:
            q = q.dbl().dbl().dbl().dbl()
            q = q.add(mults[ki])
        return q


class Group:
    def sizeFieldBytes(self):  # group.ts:49-52
        return (self.p.bit_length() + 7) // 8

    def sizePointBytes(self):
        return 1 + 2 * self.sizeFieldBytes()

    def newScalar(self, s):
        return Scalar(self, s)

    def randomScalar(self, rng):  # group.ts:59-61
        return self.newScalar(rnd(self.order, rng))


class WeierstrassGroup(Group):  # weier.ts:25-89
    def __init__(self, name, p, a, b, order, gen):
        self.name, self.p, self.a, self.b, self.order, self.gen = name, p, a, b, order, gen
        if posMod(a, p) != p - 3:
            raise ValueError('only supports a=-3')
        if not self.isOnGroup(self.generator()):
            raise ValueError('generator not on group')

    def identity(self):
        return WeierstrassPoint(self, 0, 1, 0)

    def generator(self):
        return WeierstrassPoint(self, self.gen[0], self.gen[1], 1)

    def isOnGroup(self, pt):  # weier.ts:56-70
        p, a, b = self.p, self.a, self.b
        x, y, z = pt.x, pt.y, pt.z
        y2z = (y * y % p) * z % p
        x3 = (x * x * x) % p
        z2 = (z * z) % p
        axz2 = (a * x % p) * z2 % p
        bz3 = b * (z2 * z % p) % p
        return pt.group is self and posMod(y2z - (x3 + axz2 + bz3), p) == 0

    def deserializePoint(self, a):  # weier.ts:74-89
        a = bytes(a)
        if len(a) == 1 and a[0] == 0:
            return self.identity()
        if len(a) == self.sizePointBytes() and a[0] == 0x04:
            cs = self.sizeFieldBytes()
            pt = WeierstrassPoint(self, fromBytes(a[1:1 + cs]), fromBytes(a[1 + cs:]))
            if not self.isOnGroup(pt):
                raise ValueError('point not in group')
            return pt
        raise ValueError('error deserializing Point')


class WeierstrassPoint(Point):  # weier.ts:96-261
    __slots__ = ('group', 'x', 'y', 'z')

    def __init__(self, g, x, y, z=1):
        self.group, self.x, self.y, self.z = g, x, y, z

    def isIdentity(self):  # weier.ts:117-119
        return self.x == 0 and self.y != 0 and self.z == 0

    def eq(self, pt):  # weier.ts:120-128
        p = self.group.p
        return (self.group is pt.group and (self.x * pt.z) % p == (pt.x * self.z) % p
                and (self.y * pt.z) % p == (pt.y * self.z) % p)

    def neg(self):
        return WeierstrassPoint(self.group, self.x, posMod(-self.y, self.group.p), self.z)

    def dbl(self):  # weier.ts:133-175 (RCB complete doubling, a = -3)
        x, y, z = self.x, self.y, self.z
        p, b = self.group.p, self.group.b
        t0 = (x * x) % p
        t1 = (y * y) % p
        t2 = (z * z) % p
        t3 = (x * y) % p
        t3 = (t3 + t3) % p
        z3 = (x * z) % p
        z3 = (z3 + z3) % p
        y3 = (b * t2) % p
        y3 = (y3 - z3) % p
        x3 = (y3 + y3) % p
        y3 = (x3 + y3) % p
        x3 = (t1 - y3) % p
        y3 = (t1 + y3) % p
        y3 = (x3 * y3) % p
        x3 = (x3 * t3) % p
        t3 = (t2 + t2) % p
        t2 = (t2 + t3) % p
        z3 = (b * z3) % p
        z3 = (z3 - t2) % p
        z3 = (z3 - t0) % p
        t3 = (z3 + z3) % p
        z3 = (z3 + t3) % p
        t3 = (t0 + t0) % p
        t0 = (t3 + t0) % p
        t0 = (t0 - t2) % p
        t0 = (t0 * z3) % p
        y3 = (y3 + t0) % p
        t0 = (y * z) % p
        t0 = (t0 + t0) % p
        z3 = (t0 * z3) % p
        x3 = (x3 - z3) % p
        z3 = (t0 * t1) % p
        z3 = (z3 + z3) % p
        z3 = (z3 + z3) % p
        return WeierstrassPoint(self.group, x3 % p, y3 % p, z3 % p)

    def add(self, pt):  # weier.ts:176-230 (RCB complete addition, a = -3)
        x1, y1, z1 = self.x, self.y, self.z
        x2, y2, z2 = pt.x, pt.y, pt.z
        p, b = self.group.p, self.group.b
        t0 = (x1 * x2) % p
        t1 = (y1 * y2) % p
        t2 = (z1 * z2) % p
        t3 = (x1 + y1) % p
        t4 = (x2 + y2) % p
        t3 = (t3 * t4) % p
        t4 = (t0 + t1) % p
        t3 = (t3 - t4) % p
        t4 = (y1 + z1) % p
        x3 = (y2 + z2) % p
        t4 = (t4 * x3) % p
        x3 = (t1 + t2) % p
        t4 = (t4 - x3) % p
        x3 = (x1 + z1) % p
        y3 = (x2 + z2) % p
        x3 = (x3 * y3) % p
        y3 = (t0 + t2) % p
        y3 = (x3 - y3) % p
        z3 = (b * t2) % p
        x3 = (y3 - z3) % p
        z3 = (x3 + x3) % p
        x3 = (x3 + z3) % p
        z3 = (t1 - x3) % p
        x3 = (t1 + x3) % p
        y3 = (b * y3) % p
        t1 = (t2 + t2) % p
        t2 = (t1 + t2) % p
        y3 = (y3 - t2) % p
        y3 = (y3 - t0) % p
        t1 = (y3 + y3) % p
        y3 = (t1 + y3) % p
        t1 = (t0 + t0) % p
        t0 = (t1 + t0) % p
        t0 = (t0 - t2) % p
        t1 = (t4 * y3) % p
        t2 = (t0 * y3) % p
        y3 = (x3 * z3) % p
        y3 = (y3 + t2) % p
        x3 = (t3 * x3) % p
        x3 = (x3 - t1) % p
        z3 = (t4 * z3) % p
        t1 = (t3 * t0) % p
        z3 = (z3 + t1) % p
        return WeierstrassPoint(self.group, x3 % p, y3 % p, z3 % p)

    def toAffine(self):  # weier.ts:231-243 (normalises in place)
        if self.isIdentity():
            self.y = 1
            return False
        p = self.group.p
        zInv = invMod(self.z, p)
        x, y = posMod(self.x * zInv, p), posMod(self.y * zInv, p)
        self.x, self.y, self.z = x, y, 1
        return (x, y)

    def toBytes(self):  # weier.ts:244-255
        coord = self.toAffine()
        if not coord:
            return bytes(1)
        cs = self.group.sizeFieldBytes()
        return b'\x04' + toBytes(coord[0], cs) + toBytes(coord[1], cs)


class TEdwards(Group):  # edwards.ts:25-86
    def __init__(self, name, p, a, d, order, gen):
        self.name, self.p, self.a, self.d, self.order, self.gen = name, p, a, d, order, gen
        if not self.isOnGroup(self.generator()):
            raise ValueError('generator not on group')

    def identity(self):
        return TEdwardsPoint(self, 0, 1)

    def generator(self):
        return TEdwardsPoint(self, self.gen[0], self.gen[1], posMod(self.gen[0] * self.gen[1], self.p), 1)

    def isOnGroup(self, pt):  # edwards.ts:52-65
        p, a, d = self.p, self.a, self.d
        x, y, t, z = pt.x, pt.y, pt.t, pt.z
        l0 = (a * (x * x % p) + (y * y % p)) % p
        r0 = ((z * z % p) + d * (t * t % p)) % p
        return pt.group is self and posMod(l0 - r0, p) == 0 and posMod(x * y - z * t, p) == 0

    def deserializePoint(self, b):  # edwards.ts:70-86
        b = bytes(b)
        if len(b) == self.sizePointBytes() and b[0] == 0x04:
            cs = self.sizeFieldBytes()
            x, y = fromBytes(b[1:1 + cs]), fromBytes(b[1 + cs:])
            if not (0 <= x < self.p and 0 <= y < self.p):
                raise ValueError('a not in range')
            pt = TEdwardsPoint(self, x, y, posMod(x * y, self.p), 1)
            if not self.isOnGroup(pt):
                raise ValueError('point not on TEdwards group')
            return pt
        raise ValueError('error deserializing TEdwardsPoint')


class TEdwardsPoint(Point):  # edwards.ts:93-210
    __slots__ = ('group', 'x', 'y', 't', 'z')

    def __init__(self, g, x, y, t=None, z=None):
        self.group, self.x, self.y = g, x, y
        self.t = t if t is not None else x * y
        self.z = z if z is not None else 1

    def isIdentity(self):  # edwards.ts:117-125
        return self.x == 0 and self.y != 0 and self.t == 0 and self.z != 0 and self.y == self.z

    def eq(self, pt):  # edwards.ts:126-135
        p = self.group.p
        return (self.group is pt.group and posMod(self.x * pt.z, p) == posMod(pt.x * self.z, p)
                and posMod(self.y * pt.z, p) == posMod(pt.y * self.z, p))

    def neg(self):
        p = self.group.p
        return TEdwardsPoint(self.group, posMod(-self.x, p), self.y, posMod(-self.t, p), self.z)

    def dbl(self):  # edwards.ts:141-160 (Hisil et al. sec. 3.3)
        x, y, z = self.x, self.y, self.z
        p, a = self.group.p, self.group.a
        A = (x * x) % p
        B = (y * y) % p
        C = (2 * z * z) % p
        D = (a * A) % p
        E = ((x + y) * (x + y) - A - B) % p
        G = (D + B) % p
        F = (G - C) % p
        H = (D - B) % p
        return TEdwardsPoint(self.group, posMod(E * F, p), posMod(G * H, p), posMod(E * H, p), posMod(F * G, p))

    def add(self, pt):  # edwards.ts:161-183 (Hisil et al. sec. 3.1)
        x1, y1, t1, z1 = self.x, self.y, self.t, self.z
        x2, y2, t2, z2 = pt.x, pt.y, pt.t, pt.z
        p, a, d = self.group.p, self.group.a, self.group.d
        A = (x1 * x2) % p
        B = (y1 * y2) % p
        C = (d * t1 * t2) % p
        D = (z1 * z2) % p
        E = (((x1 + y1) % p) * ((x2 + y2) % p) - A - B) % p
        F = (D - C) % p
        G = (D + C) % p
        H = (B - a * A) % p
        return TEdwardsPoint(self.group, posMod(E * F, p), posMod(G * H, p), posMod(E * H, p), posMod(F * G, p))

    def toAffine(self):  # edwards.ts:184-193 (normalises in place; never fails)
        p = self.group.p
        zInv = invMod(self.z, p)
        x, y = posMod(self.x * zInv, p), posMod(self.y * zInv, p)
        self.x, self.y, self.t, self.z = x, y, posMod(x * y, p), 1
        return (x, y)

    def toBytes(self):  # edwards.ts:194-203
        x, y = self.toAffine()
        cs = self.group.sizeFieldBytes()
        return b'\x04' + toBytes(x, cs) + toBytes(y, cs)


# instances.ts:22-54
p256 = WeierstrassGroup(
    'p256',
    0xffffffff00000001000000000000000000000000ffffffffffffffffffffffff,
    0xffffffff00000001000000000000000000000000fffffffffffffffffffffffc,
    0x5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b,
    0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551,
    (0x6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296,
     0x4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5))

tomEdwards256 = TEdwards(
    'tomEdwards256',
    0x3fffffffc000000040000000000000002ae382c7957cc4ff9713c3d82bc47d3af,
    0x1abce3fd8e1d7a21252515332a512e09d4249bd5b1ec35e316c02254fe8cedf5d,
    0x051781d9823abde00ec99295ba542c8b1401874bcbeb9e9c861174c7bca6a02aa,
    0x0ffffffff00000001000000000000000000000000ffffffffffffffffffffffff,
    (0x7907055d0a7d4abc3eafdc25d431d9659fbe007ee2d8ddc4e906206ea9ba4fdb,
     0xbe231cb9f9bf18319c9f081141559b0a33dddccd2221f0464a9cd57081b01a01))


def hashPoints(points, suffix=b''):  # group.ts:221-233 (SHA-256; first 10 bytes, big-endian); suffix: hardened mode only
    data = b''.join(p.toBytes() for p in points) + suffix
    return fromBytes(hashlib.sha256(data).digest()[:10])


# --------------------------------------------------------------------------------------
# curves/multimult.ts
# --------------------------------------------------------------------------------------


class _Pair:
    __slots__ = ('pt', 'scalar')

    def __init__(self, pt, scalar):
        self.pt, self.scalar = pt, scalar

    def cmp(self, b):
        return self.scalar.cmp(b.scalar)


def _bubbleup(arr, index):  # multimult.ts:113-124
    while index > 1:
        parent = index // 2
        if arr[parent - 1].cmp(arr[index - 1]) < 0:
            arr[index - 1], arr[parent - 1] = arr[parent - 1], arr[index - 1]
            index = parent
        else:
            return


def _pushdown(arr, parent):  # multimult.ts:126-145
    while True:
        son, daughter = 2 * parent, 2 * parent + 1
        if son > len(arr):
            return
        child = son
        if daughter <= len(arr) and arr[daughter - 1].cmp(arr[son - 1]) > 0:
            child = daughter
        if arr[parent - 1].cmp(arr[child - 1]) < 0:
            arr[child - 1], arr[parent - 1] = arr[parent - 1], arr[child - 1]
            parent = child
        else:
            return


def _extractMax(arr):  # multimult.ts:92-104
    arr[0], arr[-1] = arr[-1], arr[0]
    mx = arr.pop()
    _pushdown(arr, 1)
    return mx


class MultiMult:  # multimult.ts:31-90
    def __init__(self, g):
        self.group, self.pairs, self.known = g, [], []

    def addKnown(self, pt):
        if not any(pt.eq(k[0]) for k in self.known):
            self.pairs.append(_Pair(pt, self.group.newScalar(0)))
            self.known.append((pt, len(self.pairs) - 1))

    def insert(self, pt, s):
        for kpt, idx in self.known:
            if pt.eq(kpt):
                self.pairs[idx].scalar = self.pairs[idx].scalar.add(s)
                return
        self.pairs.append(_Pair(pt, s))

    def evaluate(self):  # Bos-Coster, multimult.ts:61-89
        if len(self.pairs) == 0:
            return self.group.identity()
        if len(self.pairs) == 1:
            return self.pairs[0].pt.mul(self.pairs[0].scalar)
        for i in range(len(self.pairs)):
            _bubbleup(self.pairs, i + 1)
        while True:
            if len(self.pairs) == 1:
                a = self.pairs[0]
                return a.pt.mul(a.scalar)
            a = _extractMax(self.pairs)
            b = self.pairs[0]
            if b.scalar.isZero():
                return a.pt.mul(a.scalar)
            c = _Pair(a.pt, a.scalar.sub(b.scalar))
            d = _Pair(b.pt.add(a.pt), b.scalar)
            self.pairs[0] = d
            if not c.scalar.isZero():
                self.pairs.append(c)
                _bubbleup(self.pairs, len(self.pairs))


class Relation:  # multimult.ts:147-174
    def __init__(self, g):
        self.group, self.pairs = g, []

    def insert(self, pt, s):
        self.pairs.append(_Pair(pt, s))

    def insertM(self, pts, scalars):
        if len(pts) != len(scalars):
            raise ValueError('arrays are not the same length')
        for pt, s in zip(pts, scalars):
            self.insert(pt, s)

    def drain(self, m, vrng):
        randomizer = self.group.randomScalar(vrng)
        for pr in self.pairs:
            m.insert(pr.pt, pr.scalar.mul(randomizer))


# --------------------------------------------------------------------------------------
# commit/pedersen.ts, equality.ts, mult.ts
# --------------------------------------------------------------------------------------


class Commitment:  # pedersen.ts:21-36
    __slots__ = ('p', 'r')

    def __init__(self, p, r):
        self.p, self.r = p, r

    def add(self, c):
        return Commitment(self.p.add(c.p), self.r.add(c.r))

    def sub(self, c):
        return Commitment(self.p.sub(c.p), self.r.sub(c.r))


class PedersenParams:  # pedersen.ts:40-58
    def __init__(self, c, g, h):
        self.c, self.g, self.h = c, g, h

    def commit(self, value, rng):
        r = self.c.randomScalar(rng)
        v = self.c.newScalar(value)
        return Commitment(self.h.dblmul(r, self.g, v), r)


def generatePedersenParams(c, rng, g=None):  # pedersen.ts:61-69
    if g is None:
        g = c.generator()
    r = c.randomScalar(rng)
    return PedersenParams(c, g, g.mul(r))


class EqualityProof:  # equality.ts:27-52
    def __init__(self, A_1, A_2, t_x, t_r1, t_r2):
        self.A_1, self.A_2, self.t_x, self.t_r1, self.t_r2 = A_1, A_2, t_x, t_r1, t_r2


def proveEquality(params, x, C1, C2, rng):  # equality.ts:60-78
    k = rnd(params.c.order, rng)
    A1 = params.commit(k, rng)
    A2 = params.commit(k, rng)
    c = hashPoints([C1.p, C2.p, A1.p, A2.p])
    cc, xx, kk = params.c.newScalar(c), params.c.newScalar(x), params.c.newScalar(k)
    tx = kk.sub(cc.mul(xx))
    tr1 = A1.r.sub(cc.mul(C1.r))
    tr2 = A2.r.sub(cc.mul(C2.r))
    return EqualityProof(A1.p, A2.p, tx, tr1, tr2)


def aggregateEquality(params, C1, C2, pi, multi, vrng):  # equality.ts:94-116
    cc = params.c.newScalar(hashPoints([C1, C2, pi.A_1, pi.A_2]))
    one = params.c.newScalar(1)
    A1rel = Relation(params.c)
    A1rel.insert(params.g, pi.t_x)
    A1rel.insert(params.h, pi.t_r1)
    A1rel.insert(C1, cc)
    A1rel.insert(pi.A_1.neg(), one)
    A2rel = Relation(params.c)
    A2rel.insert(params.g, pi.t_x)
    A2rel.insert(params.h, pi.t_r2)
    A2rel.insert(C2, cc)
    A2rel.insert(pi.A_2.neg(), one)
    A1rel.drain(multi, vrng)
    A2rel.drain(multi, vrng)
    return True


class MultProof:  # mult.ts:26-91
    FIELDS_P = ('C_4', 'A_x', 'A_y', 'A_z', 'A_4_1', 'A_4_2')
    FIELDS_S = ('t_x', 't_y', 't_z', 't_rx', 't_ry', 't_rz', 't_r4')

    def __init__(self, *a):
        for name, v in zip(self.FIELDS_P + self.FIELDS_S, a):
            setattr(self, name, v)


def proveMult(params, x, y, z, Cx, Cy, Cz, rng):  # mult.ts:93-131
    c_ = params.c
    xx = c_.newScalar(x)
    C4 = Cy.p.mul(xx)
    r4 = Cy.r.mul(xx)
    k_x = rnd(c_.order, rng)
    k_y = rnd(c_.order, rng)
    k_z = rnd(c_.order, rng)
    kx = c_.newScalar(k_x)
    Ax = params.commit(k_x, rng)
    Ay = params.commit(k_y, rng)
    Az = params.commit(k_z, rng)
    A4_1 = params.commit(k_z, rng)
    A4_2 = Cy.p.mul(kx)
    c = hashPoints([Cx.p, Cy.p, Cz.p, C4, Ax.p, Ay.p, Az.p, A4_1.p, A4_2])
    cc, ky, kz = c_.newScalar(c), c_.newScalar(k_y), c_.newScalar(k_z)
    yy, zz = c_.newScalar(y), c_.newScalar(z)
    t_x = kx.sub(cc.mul(xx))
    t_y = ky.sub(cc.mul(yy))
    t_z = kz.sub(cc.mul(zz))
    t_rx = Ax.r.sub(cc.mul(Cx.r))
    t_ry = Ay.r.sub(cc.mul(Cy.r))
    t_rz = Az.r.sub(cc.mul(Cz.r))
    t_r4 = A4_1.r.sub(cc.mul(r4))
    return MultProof(C4, Ax.p, Ay.p, Az.p, A4_1.p, A4_2, t_x, t_y, t_z, t_rx, t_ry, t_rz, t_r4)


def aggregateMult(params, Cx, Cy, Cz, pi, multi, vrng):  # mult.ts:148-175
    c_ = params.c
    cc = c_.newScalar(hashPoints([Cx, Cy, Cz, pi.C_4, pi.A_x, pi.A_y, pi.A_z, pi.A_4_1, pi.A_4_2]))
    one = c_.newScalar(1)
    rels = []
    for pts, scs in (
        ([params.g, params.h, Cx, pi.A_x.neg()], [pi.t_x, pi.t_rx, cc, one]),
        ([params.g, params.h, Cy, pi.A_y.neg()], [pi.t_y, pi.t_ry, cc, one]),
        ([params.g, params.h, Cz, pi.A_z.neg()], [pi.t_z, pi.t_rz, cc, one]),
        ([params.g, params.h, pi.C_4, pi.A_4_1.neg()], [pi.t_z, pi.t_r4, cc, one]),
        ([Cy, pi.C_4, pi.A_4_2.neg()], [pi.t_x, cc, one]),
    ):
        r = Relation(c_)
        r.insertM(pts, scs)
        rels.append(r)
    for r in rels:
        r.drain(multi, vrng)
    return True


# --------------------------------------------------------------------------------------
# exp/pointAdd.ts, exp/exp.ts
# --------------------------------------------------------------------------------------


class PointAddProof:  # pointAdd.ts:28-76
    def __init__(self, C_8, C_10, C_11, C_13, pi_8, pi_10, pi_11, pi_13, pi_x, pi_y):
        self.C_8, self.C_10, self.C_11, self.C_13 = C_8, C_10, C_11, C_13
        self.pi_8, self.pi_10, self.pi_11, self.pi_13, self.pi_x, self.pi_y = pi_8, pi_10, pi_11, pi_13, pi_x, pi_y


def provePointAdd(params, P, Q, R, PX, PY, QX, QY, RX, RY, rng):  # pointAdd.ts:92-163
    if not P.add(Q).eq(R):
        raise ValueError("Points don't add up!")
    prime = params.c.order
    C1, C2, C3, C4, C5, C6 = PX, QX, RX, PY, QY, RY
    coordP, coordQ, coordR = P.toAffine(), Q.toAffine(), R.toAffine()
    if not coordP:
        raise ValueError('P is at infinity')
    if not coordQ:
        raise ValueError('Q is at infinity')
    if not coordR:
        raise ValueError('R is at infinity')
    (x1, y1), (x2, y2), (x3, _) = coordP, coordQ, coordR
    i7 = posMod(x2 - x1, prime)
    i8 = invMod(i7, prime)
    i9 = posMod(y2 - y1, prime)
    i10 = posMod(i8 * i9, prime)
    i11 = posMod(i10 * i10, prime)
    i12 = posMod(x1 - x3, prime)
    i13 = posMod(i10 * i12, prime)
    C7 = C2.sub(C1)
    C8 = params.commit(i8, rng)
    C9 = C5.sub(C4)
    C10 = params.commit(i10, rng)
    C11 = params.commit(i11, rng)
    C12 = C1.sub(C3)
    C13 = params.commit(i13, rng)
    C14 = Commitment(params.g, params.c.newScalar(0))
    pi8 = proveMult(params, i7, i8, 1, C7, C8, C14, rng)
    pi10 = proveMult(params, i8, i9, i10, C8, C9, C10, rng)
    pi11 = proveMult(params, i10, i10, i11, C10, C10, C11, rng)
    Cint = Commitment(C3.p.add(C1.p).add(C2.p), C3.r.add(C1.r).add(C2.r))
    pix = proveEquality(params, i11, C11, Cint, rng)
    pi13 = proveMult(params, i10, i12, i13, C10, C12, C13, rng)
    Cint = Commitment(C6.p.add(C4.p), C6.r.add(C4.r))
    piy = proveEquality(params, i13, C13, Cint, rng)
    return PointAddProof(C8.p, C10.p, C11.p, C13.p, pi8, pi10, pi11, pi13, pix, piy)


def aggregatePointAdd(params, PX, PY, QX, QY, RX, RY, pi, multi, vrng):  # pointAdd.ts:199-259
    C1, C2, C3, C4, C5, C6 = PX, QX, RX, PY, QY, RY
    C7, C9, C12 = C2.sub(C1), C5.sub(C4), C1.sub(C3)
    C_14 = params.g
    if not aggregateMult(params, C7, pi.C_8, C_14, pi.pi_8, multi, vrng):
        return False
    if not aggregateMult(params, pi.C_8, C9, pi.C_10, pi.pi_10, multi, vrng):
        return False
    if not aggregateMult(params, pi.C_10, pi.C_10, pi.C_11, pi.pi_11, multi, vrng):
        return False
    Cint = C3.add(C1).add(C2)
    if not aggregateEquality(params, pi.C_11, Cint, pi.pi_x, multi, vrng):
        return False
    if not aggregateMult(params, pi.C_10, C12, pi.C_13, pi.pi_13, multi, vrng):
        return False
    Cint = C4.add(C6)
    if not aggregateEquality(params, pi.C_13, Cint, pi.pi_y, multi, vrng):
        return False
    return True


class ExpProof:  # exp.ts:26-84
    def __init__(self, A, Tx, Ty, alpha=None, beta1=None, beta2=None, beta3=None,
                 z=None, z2=None, proof=None, r1=None, r2=None):
        self.A, self.Tx, self.Ty = A, Tx, Ty
        self.alpha, self.beta1, self.beta2, self.beta3 = alpha, beta1, beta2, beta3
        self.z, self.z2, self.proof, self.r1, self.r2 = z, z2, proof, r1, r2


def paddedBits(val, length):  # exp.ts:86-93
    return [((val >> i) & 1) == 1 for i in range(length)]


def generateIndices(indnum, limit, vrng):  # exp.ts:95-109 (slice result discarded in the reference)
    ret = list(range(limit))
    for i in range(limit - 2):
        j = rndRange(i, limit - 1, vrng)
        ret[i], ret[j] = ret[j], ret[i]
    return ret


def proveExp(paramsNIST, paramsWario, s, Cs, P, Px, Py, secparam, rng, Q=None):  # exp.ts:126-231
    alpha, r, T, A, Tx, Ty = [], [], [], [], [], []
    for i in range(secparam):
        alpha.append(paramsNIST.c.randomScalar(rng))
        r.append(paramsNIST.c.randomScalar(rng))
        T.append(paramsNIST.g.mul(alpha[i]))
        A.append(T[i].add(paramsNIST.h.mul(r[i])))
        coordT = T[i].toAffine()
        if not coordT:
            raise ValueError('T[i] is at infinity')
        Tx.append(paramsWario.commit(coordT[0], rng))
        Ty.append(paramsWario.commit(coordT[1], rng))
    arr = [Px.p, Py.p]
    for i in range(secparam):
        arr += [A[i], Tx[i].p, Ty[i].p]
    challenge = hashPoints(arr)
    allProofs = []
    for i in range(secparam):
        if challenge & 1:
            proof = ExpProof(A[i], Tx[i].p, Ty[i].p, alpha[i], r[i], Tx[i].r, Ty[i].r)
        else:
            z = alpha[i].sub(paramsNIST.c.newScalar(s))
            T1 = paramsNIST.g.mul(z)
            if Q is not None:
                T1 = T1.add(Q)
            coordT1 = T1.toAffine()
            if not coordT1:
                raise ValueError('T1 is at infinity')
            T1x = paramsWario.commit(coordT1[0], rng)
            T1y = paramsWario.commit(coordT1[1], rng)
            pap = provePointAdd(paramsWario, T1, P, T[i], T1x, T1y, Px, Py, Tx[i], Ty[i], rng)
            proof = ExpProof(A[i], Tx[i].p, Ty[i].p, None, None, None, None,
                             z, r[i].sub(Cs.r), pap, T1x.r, T1y.r)
        allProofs.append(proof)
        challenge >>= 1
    return allProofs


def verifyExp(paramsNIST, paramsWario, Clambda, Px, Py, pi, secparam, vrng, Q=None):  # exp.ts:233-349
    if secparam > len(pi):
        raise ValueError('security level not achieved')
    multiW, multiN = MultiMult(paramsWario.c), MultiMult(paramsNIST.c)
    multiW.addKnown(paramsWario.g)
    multiW.addKnown(paramsWario.h)
    multiN.addKnown(paramsNIST.g)
    multiN.addKnown(paramsNIST.h)
    multiN.addKnown(Clambda)
    arr = [Px, Py]
    for e in pi:
        arr += [e.A, e.Tx, e.Ty]
    challenge = hashPoints(arr)
    indices = generateIndices(secparam, len(pi), vrng)
    challengeBits = paddedBits(challenge, len(pi))
    cN, cW = paramsNIST.c, paramsWario.c
    for j in range(secparam):
        i = indices[j]
        e = pi[i]
        if challengeBits[i]:
            if not (e.alpha and e.beta1 and e.beta2 and e.beta3):
                raise ValueError('params not found')
            T = paramsNIST.g.mul(e.alpha)
            relA = Relation(cN)
            relA.insertM([T, paramsNIST.h, e.A.neg()], [cN.newScalar(1), e.beta1, cN.newScalar(1)])
            relA.drain(multiN, vrng)
            coordT = T.toAffine()
            if not coordT:
                raise ValueError('T is at infinity')
            sx, sy = cW.newScalar(coordT[0]), cW.newScalar(coordT[1])
            relTx, relTy = Relation(cW), Relation(cW)
            relTx.insertM([paramsWario.g, paramsWario.h, e.Tx.neg()], [sx, e.beta2, cW.newScalar(1)])
            relTy.insertM([paramsWario.g, paramsWario.h, e.Ty.neg()], [sy, e.beta3, cW.newScalar(1)])
            relTx.drain(multiW, vrng)
            relTy.drain(multiW, vrng)
        else:
            if not (e.z and e.z2 and e.proof and e.r1 and e.r2):
                raise ValueError('params not found')
            T1 = paramsNIST.g.mul(e.z)
            relA = Relation(cN)
            relA.insertM([T1, Clambda, e.A.neg(), paramsNIST.h],
                         [cN.newScalar(1), cN.newScalar(1), cN.newScalar(1), e.z2])
            relA.drain(multiN, vrng)
            if Q is not None:
                T1 = T1.add(Q)
            coordT1 = T1.toAffine()
            if not coordT1:
                raise ValueError('T1 is at infinity')
            sx, sy = cW.newScalar(coordT1[0]), cW.newScalar(coordT1[1])
            T1x = paramsWario.g.dblmul(sx, paramsWario.h, e.r1)
            T1y = paramsWario.g.dblmul(sy, paramsWario.h, e.r2)
            if not aggregatePointAdd(paramsWario, T1x, T1y, Px, Py, e.Tx, e.Ty, e.proof, multiW, vrng):
                return False
    return multiW.evaluate().isIdentity() and multiN.evaluate().isIdentity()


# --------------------------------------------------------------------------------------
# proofGK/interpolate.ts, gk.ts
# --------------------------------------------------------------------------------------


def _jsmod(a, m):
    """JS BigInt '%': sign follows the dividend."""
    r = abs(a) % m
    return -r if a < 0 else r


def eval_poly(coeff, x, m):  # interpolate.ts:19-25
    ret = 0
    for i in range(len(coeff) - 1, -1, -1):
        ret = posMod(coeff[i] + x * ret, m)
    return ret


def interpolate(x, y, m):  # interpolate.ts:27-70
    if len(x) != len(y):
        raise ValueError('inconsistent args')
    n = len(x)
    s = [0] * (n + 1)
    coeff = [0] * n
    s[n] = 1
    s[n - 1] = _jsmod(-x[0], m)
    for i in range(1, n):
        for j in range(n - i - 1, n - 1):
            s[j] = _jsmod(s[j] - x[i] * s[j + 1], m)
        s[n - 1] = _jsmod(s[n - 1] - x[i], m)
    for i in range(n):
        phi = 0
        for j in range(n, 0, -1):
            phi = j * s[j] + x[i] * phi
        phi = posMod(phi, m)
        ff = invMod(phi, m) % m
        b = 1
        for j in range(n - 1, -1, -1):
            coeff[j] = posMod(coeff[j] + b * ff * y[i], m)
            b = s[j] + x[i] * b
    for i in range(n):
        if y[i] != eval_poly(coeff, x[i], m):
            raise ValueError('incorrect interpolation')
    return coeff


class GKProof:  # gk.ts:31-73
    def __init__(self, cl, ca, cb, cd, f, za, zb, zd):
        self.cl, self.ca, self.cb, self.cd, self.f, self.za, self.zb, self.zd = cl, ca, cb, cd, f, za, zb, zd


def _ceil_log2(n):
    return math.ceil(math.log2(n))


def pad(vals, c):  # gk.ts:75-86
    ret = [c.newScalar(v) for v in vals]
    padLen = 2 ** _ceil_log2(len(vals))
    for _ in range(len(vals), padLen):
        ret.append(ret[0])
    return ret


def gk_commit(params, val, blinder):  # gk.ts:88-92
    order = params.c.order
    return params.g.dblmul(params.c.newScalar(posMod(val, order)), params.h,
                           params.c.newScalar(posMod(blinder, order)))


def proveMembership(params, com, index, initialValues, rng, dv_override=None, statement=b''):  # gk.ts:94-195
    values = pad(initialValues, params.c)
    c = params.c
    n = _ceil_log2(len(values))
    eli = []
    l_tmp = index
    for i in range(n):
        eli.append(l_tmp % 2)
        l_tmp //= 2
    ri, ai, si, ti, rho = [], [], [], [], []
    for i in range(n):
        ri.append(rnd(c.order, rng))
        ai.append(rnd(c.order, rng))
        si.append(rnd(c.order, rng))
        ti.append(rnd(c.order, rng))
        rho.append(rnd(c.order, rng))
    cl, ca, cb, cd = [], [], [], []
    for i in range(n):
        cl.append(gk_commit(params, eli[i], ri[i]))
        ca.append(gk_commit(params, ai[i], si[i]))
        cb.append(gk_commit(params, eli[i] * ai[i], ti[i]))
    omegas = list(range(n))
    dv = []
    for w in omegas:
        f0j, f1j, ratio = [], [], []
        for j in range(n):
            f0j.append(posMod((1 - eli[j]) * w - ai[j], c.order))
            f1j.append(posMod(eli[j] * w + ai[j], c.order))
            ratio.append(posMod(f1j[j] * invMod(f0j[j], c.order), c.order))
        prod = 1
        for i in range(len(f0j)):
            prod = posMod(prod * f0j[i], c.order)
        p = [prod]
        for i in range(n):
            oldlen = len(p)
            for j in range(oldlen):
                p.append(posMod(ratio[i] * p[j], c.order))
        dval = 0
        vl = values[index].k
        for i in range(len(values)):
            dval = posMod(dval + (vl - values[i].k) * p[i], c.order)
        dv.append(dval)
    di = interpolate(omegas, dv, c.order)
    for i in range(n):
        cd.append(gk_commit(params, di[i], rho[i]))
    x = hashPoints(cl + ca + cb + cd, statement)
    f, za, zb = [], [], []
    zd = (com.r.k * expMod(x, n, c.order)) % c.order
    for i in range(n):
        f.append(c.newScalar(posMod(eli[i] * x + ai[i], c.order)))
        za.append(c.newScalar(posMod(ri[i] * x + si[i], c.order)))
        zb.append(c.newScalar(posMod(ri[i] * (x - f[i].k) + ti[i], c.order)))
    for i in range(n):
        zd = posMod(zd - rho[i] * expMod(x, i, c.order), c.order)
    return GKProof(cl, ca, cb, cd, f, za, zb, c.newScalar(zd))


def verifyMembership(params, com, initVec, proof, vrng, statement=b''):  # gk.ts:197-262
    c = params.c
    multi = MultiMult(c)
    vec = pad(initVec, c)
    n = _ceil_log2(len(vec))
    if not (n == len(proof.cl) == len(proof.ca) == len(proof.cb) == len(proof.cd)
            == len(proof.f) == len(proof.za) == len(proof.zb)):
        return False
    f = proof.f
    x = hashPoints(proof.cl + proof.ca + proof.cb + proof.cd, statement)
    multi.addKnown(params.g)
    multi.addKnown(params.h)
    for i in range(n):
        rel0 = Relation(c)
        rel0.insertM([proof.cl[i], proof.ca[i], params.g, params.h],
                     [c.newScalar(x), c.newScalar(1), proof.f[i].neg(), proof.za[i].neg()])
        rel0.drain(multi, vrng)
        rel1 = Relation(c)
        rel1.insertM([proof.cl[i], proof.cb[i], params.h],
                     [c.newScalar(posMod(x - f[i].k, c.order)), c.newScalar(1), proof.zb[i].neg()])
        rel1.drain(multi, vrng)
    # gk.ts:239-250, evaluated in the (exactly equal) fold form to keep the oracle usable at N=2^16
    layer = [v.k for v in vec]
    q = c.order
    for j in range(n):
        fj, gj = f[j].k, posMod(x - f[j].k, q)
        layer = [(gj * layer[2 * i] + fj * layer[2 * i + 1]) % q for i in range(len(layer) // 2)]
    total = layer[0] if n > 0 else (vec[0].k % q)
    relFinal = Relation(c)
    for i in range(n):
        relFinal.insert(proof.cd[i], c.newScalar(posMod(-expMod(x, i, c.order), c.order)))
    relFinal.insert(com, c.newScalar(expMod(x, n, c.order)))
    relFinal.insertM([params.g, params.h], [c.newScalar(posMod(-total, c.order)), proof.zd.neg()])
    relFinal.drain(multi, vrng)
    return multi.evaluate().isIdentity()


def gk_total_naive(vec_k, f_k, x, q):
    """gk.ts:239-250 exactly as written (O(N n)); used by tests to pin the fold form above."""
    n = len(f_k)
    total = 0
    for i in range(len(vec_k)):
        pix = 1
        for j in range(n):
            pix = posMod(pix * (f_k[j] if (i & (1 << j)) else (x - f_k[j])), q)
        total = posMod(total + vec_k[i] * pix, q)
    return total


# --------------------------------------------------------------------------------------
# zkpAttestList.ts
# --------------------------------------------------------------------------------------


class SignatureProofList:  # zkpAttestList.ts:27-60
    def __init__(self, R, comS1, keyXcom, keyYcom, expProof, membershipProof):
        self.R, self.comS1, self.keyXcom, self.keyYcom = R, comS1, keyXcom, keyYcom
        self.expProof, self.membershipProof = expProof, membershipProof


class SystemParametersList:  # zkpAttestList.ts:62-78
    def __init__(self, NistGroup, ProofGroup, SecLevel):
        self.NistGroup, self.ProofGroup, self.SecLevel = NistGroup, ProofGroup, SecLevel


def truncateToN(msg, n):  # zkpAttestList.ts:80-86
    delta = bitLen(msg) - bitLen(n)
    if delta > 0:
        msg >>= delta
    return msg


def generateParamsList(rng, secLevel=80):  # zkpAttestList.ts:88-92
    nistGroup = generatePedersenParams(p256, rng)
    proofGroup = generatePedersenParams(tomEdwards256, rng)
    return SystemParametersList(nistGroup, proofGroup, secLevel)


def keyToInt(pkBytes):  # zkpAttestList.ts:94-102 (pkBytes = WebCrypto 'raw' export, 65 B)
    pkPoint = p256.deserializePoint(pkBytes)
    pkCoords = pkPoint.toAffine()
    if not pkCoords:
        raise ValueError('invalid public key')
    return pkCoords[0]


def proveSignatureList(params, msgHash, sigBytes, pkBytes, which, keys, rng, hardened=False):  # zkpAttestList.ts:104-145
    ec = p256
    pkPoint = p256.deserializePoint(pkBytes)
    pkCoords = pkPoint.toAffine()
    if not pkCoords:
        raise ValueError('invalid public key')
    ln = len(sigBytes)
    groupOrder = ec.order
    z = truncateToN(fromBytes(msgHash), groupOrder)
    r = fromBytes(sigBytes[:ln // 2])
    s = fromBytes(sigBytes[ln // 2:])
    sinv = invMod(s, groupOrder)
    u1 = posMod(sinv * z, groupOrder)
    u2 = posMod(sinv * r, groupOrder)
    R = ec.generator().mul(ec.newScalar(u1)).add(pkPoint.mul(ec.newScalar(u2)))
    rinv = invMod(r, groupOrder)
    s1 = posMod(rinv * s, groupOrder)
    z1 = posMod(rinv * z, groupOrder)
    Q = ec.generator().mul(ec.newScalar(z1))
    paramsSigExp = PedersenParams(p256, R, params.NistGroup.h)
    comS1 = paramsSigExp.commit(s1, rng)
    pkX = params.ProofGroup.commit(pkCoords[0], rng)
    pkY = params.ProofGroup.commit(pkCoords[1], rng)
    sigProof = proveExp(paramsSigExp, params.ProofGroup, s1, comS1, pkPoint, pkX, pkY, params.SecLevel, rng, Q)
    stmt = gk_statement([v.k for v in pad(keys, params.ProofGroup.c)], msgHash, R, pkX.p) if hardened else b''
    membershipProof = proveMembership(params.ProofGroup, pkX, which, keys, rng, statement=stmt)
    return SignatureProofList(R, comS1.p, pkX.p, pkY.p, sigProof, membershipProof)


def verifySignatureList(params, msgHash, keys, proof, vrng=None, hardened=False):  # zkpAttestList.ts:147-184
    vrng = vrng or OsRng()
    ec = p256
    groupOrder = ec.order
    z = truncateToN(fromBytes(msgHash), groupOrder)
    R = proof.R
    coordR = R.toAffine()
    if not coordR:
        raise ValueError('R is at infinity')
    rinv = invMod(coordR[0], groupOrder)
    paramsSigExp = PedersenParams(p256, R, params.NistGroup.h)
    z1 = posMod(rinv * z, groupOrder)
    Q = ec.generator().mul(ec.newScalar(z1))
    stmt = gk_statement([v.k for v in pad(keys, params.ProofGroup.c)], msgHash, R, proof.keyXcom) if hardened else b''
    if not verifyMembership(params.ProofGroup, proof.keyXcom, keys, proof.membershipProof, vrng, statement=stmt):
        return False
    if not verifyExp(paramsSigExp, params.ProofGroup, proof.comS1, proof.keyXcom, proof.keyYcom,
                     proof.expProof, 20, vrng, Q):
        return False
    return True


# --------------------------------------------------------------------------------------
# Binary proof layout "ZKA1" (defined by this build; include/zkattest.h documents it).
# All integers big-endian.  P-256 coordinate 32 B, Tom-256 coordinate 36 B (zero-padded
# from the reference's 33-byte encoding so every field is 4-byte aligned), scalars 32 B.
# --------------------------------------------------------------------------------------

PB, TB, SB = 32, 36, 32
MAGIC = b'ZKA1'


def _pp(pt):
    x, y = pt.toAffine()
    return toBytes(x, PB) + toBytes(y, PB)


def _tp(pt):
    x, y = pt.toAffine()
    return toBytes(x, TB) + toBytes(y, TB)


def _sc(s):
    return toBytes(s.k, SB)


def _mult_bytes(m):
    return b''.join(_tp(getattr(m, f)) for f in MultProof.FIELDS_P) + b''.join(_sc(getattr(m, f)) for f in MultProof.FIELDS_S)


def _eq_bytes(e):
    return _tp(e.A_1) + _tp(e.A_2) + _sc(e.t_x) + _sc(e.t_r1) + _sc(e.t_r2)


def _pointadd_bytes(p):
    return (_tp(p.C_8) + _tp(p.C_10) + _tp(p.C_11) + _tp(p.C_13) + _mult_bytes(p.pi_8) + _mult_bytes(p.pi_10)
            + _mult_bytes(p.pi_11) + _mult_bytes(p.pi_13) + _eq_bytes(p.pi_x) + _eq_bytes(p.pi_y))


MULT_SZ = 6 * 2 * TB + 7 * SB       # 656
EQ_SZ = 2 * 2 * TB + 3 * SB         # 240
PADD_SZ = 4 * 2 * TB + 4 * MULT_SZ + 2 * EQ_SZ  # 3392
REP_HEAD = 2 * PB + 2 * 2 * TB + 4 * SB          # 336 : A, Tx, Ty, 4 scalars
HEADER_SZ = 32


def proof_to_bytes(proof):
    """Serialise SignatureProofList into the ZKA1 layout."""
    reps = proof.expProof
    sec = len(reps)
    n = len(proof.membershipProof.cl)
    bits = 0
    body = b''
    for i, e in enumerate(reps):
        body += _pp(e.A) + _tp(e.Tx) + _tp(e.Ty)
        if e.alpha is not None:
            bits |= 1 << i
            body += _sc(e.alpha) + _sc(e.beta1) + _sc(e.beta2) + _sc(e.beta3)
        else:
            body += _sc(e.z) + _sc(e.z2) + _sc(e.r1) + _sc(e.r2) + _pointadd_bytes(e.proof)
    g = proof.membershipProof
    gk = b''.join(_tp(p) for p in g.cl + g.ca + g.cb + g.cd) + b''.join(_sc(s) for s in g.f + g.za + g.zb) + _sc(g.zd)
    fixed = _pp(proof.R) + _pp(proof.comS1) + _tp(proof.keyXcom) + _tp(proof.keyYcom)
    total = HEADER_SZ + len(fixed) + len(body) + len(gk)
    header = MAGIC + total.to_bytes(4, 'big') + sec.to_bytes(4, 'big') + n.to_bytes(4, 'big') + bits.to_bytes(16, 'big')
    assert len(header) == HEADER_SZ
    return header + fixed + body + gk


class _Reader:
    def __init__(self, b):
        self.b, self.o = bytes(b), 0

    def take(self, n):
        v = self.b[self.o:self.o + n]
        if len(v) != n:
            raise ValueError('truncated proof')
        self.o += n
        return v

    def pp(self):
        x, y = fromBytes(self.take(PB)), fromBytes(self.take(PB))
        pt = WeierstrassPoint(p256, x, y)
        if not p256.isOnGroup(pt):  # weier.ts:256-260
            raise ValueError('point not on Weierstrass group: p256')
        return pt

    def tp(self):
        x, y = fromBytes(self.take(TB)), fromBytes(self.take(TB))
        g = tomEdwards256
        if not (x < g.p and y < g.p):
            raise ValueError('a not in range')
        pt = TEdwardsPoint(g, x, y, posMod(x * y, g.p), 1)
        if not g.isOnGroup(pt):  # edwards.ts:204-209
            raise ValueError('point not on TEdwards group: tomEdwards256')
        return pt

    def sc(self, group):
        return group.newScalar(fromBytes(self.take(SB)))


def proof_from_bytes(b):
    rd = _Reader(b)
    if rd.take(4) != MAGIC:
        raise ValueError('bad magic')
    total = fromBytes(rd.take(4))
    sec = fromBytes(rd.take(4))
    n = fromBytes(rd.take(4))
    bits = fromBytes(rd.take(16))
    if total != len(rd.b):
        raise ValueError('bad length')
    if bits >> sec:   # ZKA1 (include/zkattest.h): the challenge-bit field is zero above secLevel
        raise ValueError('challenge bits above secLevel')
    W, Nn = tomEdwards256, p256
    R, comS1, kx, ky = rd.pp(), rd.pp(), rd.tp(), rd.tp()

    def mult():
        pts = [rd.tp() for _ in range(6)]
        return MultProof(*pts, *[rd.sc(W) for _ in range(7)])

    def eq():
        a1, a2 = rd.tp(), rd.tp()
        return EqualityProof(a1, a2, rd.sc(W), rd.sc(W), rd.sc(W))

    reps = []
    for i in range(sec):
        A, Tx, Ty = rd.pp(), rd.tp(), rd.tp()
        if (bits >> i) & 1:
            reps.append(ExpProof(A, Tx, Ty, rd.sc(Nn), rd.sc(Nn), rd.sc(W), rd.sc(W)))
        else:
            z, z2, r1, r2 = rd.sc(Nn), rd.sc(Nn), rd.sc(W), rd.sc(W)
            c8, c10, c11, c13 = rd.tp(), rd.tp(), rd.tp(), rd.tp()
            p8, p10, p11, p13 = mult(), mult(), mult(), mult()
            px, py = eq(), eq()
            reps.append(ExpProof(A, Tx, Ty, None, None, None, None, z, z2,
                                 PointAddProof(c8, c10, c11, c13, p8, p10, p11, p13, px, py), r1, r2))
    cl = [rd.tp() for _ in range(n)]
    ca = [rd.tp() for _ in range(n)]
    cb = [rd.tp() for _ in range(n)]
    cd = [rd.tp() for _ in range(n)]
    f = [rd.sc(W) for _ in range(n)]
    za = [rd.sc(W) for _ in range(n)]
    zb = [rd.sc(W) for _ in range(n)]
    zd = rd.sc(W)
    if rd.o != len(rd.b):
        raise ValueError('trailing bytes')
    return SignatureProofList(R, comS1, kx, ky, reps, GKProof(cl, ca, cb, cd, f, za, zb, zd))


# --------------------------------------------------------------------------------------
# JSON wire format (writeJson/readJson, src/serde.ts:21-36, over the typedjson decorators).
# typedjson 1.8.0 is not vendored in the reference, so the exact text is UNPINNED; this follows
# its documented behaviour: members in declaration order, undefined optionals omitted, and a
# "__type" hint where the runtime class differs from the declared constructor.
# --------------------------------------------------------------------------------------
def _jhex(v):  # serdeBigInt.serializer, big.ts:230-239
    return ('-0x%x' % -v) if v < 0 else ('0x%x' % v)


def _jgroup(g):
    return {'name': g.name, '__type': 'WeierstrassGroup' if isinstance(g, WeierstrassGroup) else 'TEdwards'}


def _jpoint(pt):  # weier.ts:92-101 / edwards.ts:89-98 (beforeSerialization: toAffine)
    x, y = pt.toAffine()
    w = isinstance(pt.group, WeierstrassGroup)
    return {'group': _jgroup(pt.group), 'x': _jhex(x), 'y': _jhex(y), '__type': 'WeierstrassPoint' if w else 'TEdwardsPoint'}


def _jscalar(s):  # group.ts:155-161 (beforeSerialization: reduce)
    return {'group': _jgroup(s.group), 'k': _jhex(s.k % s.group.order)}


def _jmult(m):
    d = {f: _jpoint(getattr(m, f)) for f in MultProof.FIELDS_P}
    d.update({f: _jscalar(getattr(m, f)) for f in MultProof.FIELDS_S})
    return d


def _jeq(e):
    return {'A_1': _jpoint(e.A_1), 'A_2': _jpoint(e.A_2), 't_x': _jscalar(e.t_x), 't_r1': _jscalar(e.t_r1), 't_r2': _jscalar(e.t_r2)}


def proof_to_json(proof):
    """SignatureProofList -> JSON text (writeJson(SignatureProofList, proof))."""
    import json
    reps = []
    for e in proof.expProof:
        d = {'A': _jpoint(e.A), 'Tx': _jpoint(e.Tx), 'Ty': _jpoint(e.Ty)}
        if e.alpha is not None:
            d.update(alpha=_jscalar(e.alpha), beta1=_jscalar(e.beta1), beta2=_jscalar(e.beta2), beta3=_jscalar(e.beta3))
        else:
            p = e.proof
            pa = {k: _jpoint(getattr(p, k)) for k in ('C_8', 'C_10', 'C_11', 'C_13')}
            pa.update({k: _jmult(getattr(p, k)) for k in ('pi_8', 'pi_10', 'pi_11', 'pi_13')})
            pa.update(pi_x=_jeq(p.pi_x), pi_y=_jeq(p.pi_y))
            d.update(z=_jscalar(e.z), z2=_jscalar(e.z2), proof=pa, r1=_jscalar(e.r1), r2=_jscalar(e.r2))
        reps.append(d)
    g = proof.membershipProof
    gk = {k: [_jpoint(p) for p in getattr(g, k)] for k in ('cl', 'ca', 'cb', 'cd')}
    gk.update({k: [_jscalar(s) for s in getattr(g, k)] for k in ('f', 'za', 'zb')})
    gk['zd'] = _jscalar(g.zd)
    top = {'R': _jpoint(proof.R), 'comS1': _jpoint(proof.comS1), 'keyXcom': _jpoint(proof.keyXcom),
           'keyYcom': _jpoint(proof.keyYcom), 'expProof': reps, 'membershipProof': gk}
    return json.dumps(top, separators=(',', ':'))


def proof_from_json(text):
    """JSON text -> SignatureProofList (readJson(SignatureProofList, text)); group names as instances.ts:58-78."""
    import json
    W, Nn = tomEdwards256, p256

    def big(v):  # serdeBigInt.deserializer, big.ts:240-248
        if not v:
            raise ValueError('required field')
        return -int(v[1:], 16) if v[0] == '-' else int(v, 16)

    def grp(d):
        n = d['group']['name']
        if n == p256.name:
            return p256
        if n == tomEdwards256.name:
            return tomEdwards256
        raise ValueError('invalid group name: %s' % n)

    def pt(d):
        g = grp(d)
        x, y = big(d['x']), big(d['y'])
        if g is p256:
            p = WeierstrassPoint(p256, x, y)
        else:
            p = TEdwardsPoint(g, x, y, posMod(x * y, g.p), 1)
        if not g.isOnGroup(p):  # afterJson
            raise ValueError('point not on group')
        return p

    def sc(d):
        return grp(d).newScalar(big(d['k']))

    def mult(d):
        return MultProof(*[pt(d[f]) for f in MultProof.FIELDS_P], *[sc(d[f]) for f in MultProof.FIELDS_S])

    def eq(d):
        return EqualityProof(pt(d['A_1']), pt(d['A_2']), sc(d['t_x']), sc(d['t_r1']), sc(d['t_r2']))

    top = json.loads(text)
    reps = []
    for d in top['expProof']:
        A, Tx, Ty = pt(d['A']), pt(d['Tx']), pt(d['Ty'])
        if 'alpha' in d:
            reps.append(ExpProof(A, Tx, Ty, sc(d['alpha']), sc(d['beta1']), sc(d['beta2']), sc(d['beta3'])))
        else:
            p = d['proof']
            pa = PointAddProof(pt(p['C_8']), pt(p['C_10']), pt(p['C_11']), pt(p['C_13']), mult(p['pi_8']), mult(p['pi_10']),
                               mult(p['pi_11']), mult(p['pi_13']), eq(p['pi_x']), eq(p['pi_y']))
            reps.append(ExpProof(A, Tx, Ty, None, None, None, None, sc(d['z']), sc(d['z2']), pa, sc(d['r1']), sc(d['r2'])))
    g = top['membershipProof']
    gk = GKProof([pt(p) for p in g['cl']], [pt(p) for p in g['ca']], [pt(p) for p in g['cb']], [pt(p) for p in g['cd']],
                 [sc(s) for s in g['f']], [sc(s) for s in g['za']], [sc(s) for s in g['zb']], sc(g['zd']))
    return SignatureProofList(pt(top['R']), pt(top['comS1']), pt(top['keyXcom']), pt(top['keyYcom']), reps, gk)


# --------------------------------------------------------------------------------------
# ECDSA P-256 helpers for building inputs (not part of the reference; WebCrypto does this
# in the reference's tests).  Deterministic nonces per RFC 6979 (HMAC-SHA-256).
# --------------------------------------------------------------------------------------
import hmac


def _bits2int(b, qlen):
    v = int.from_bytes(b, 'big')
    blen = len(b) * 8
    return v >> (blen - qlen) if blen > qlen else v


def rfc6979_k(d, h1, q=p256.order):
    qlen = q.bit_length()
    rlen = (qlen + 7) // 8
    int2octets = lambda v: v.to_bytes(rlen, 'big')
    bits2octets = lambda b: int2octets(_bits2int(b, qlen) % q)
    V = b'\x01' * 32
    K = b'\x00' * 32
    K = hmac.new(K, V + b'\x00' + int2octets(d) + bits2octets(h1), hashlib.sha256).digest()
    V = hmac.new(K, V, hashlib.sha256).digest()
    K = hmac.new(K, V + b'\x01' + int2octets(d) + bits2octets(h1), hashlib.sha256).digest()
    V = hmac.new(K, V, hashlib.sha256).digest()
    while True:
        T = b''
        while len(T) < rlen:
            V = hmac.new(K, V, hashlib.sha256).digest()
            T += V
        k = _bits2int(T, qlen)
        if 1 <= k < q:
            return k
        K = hmac.new(K, V + b'\x00', hashlib.sha256).digest()
        V = hmac.new(K, V, hashlib.sha256).digest()


def ecdsa_pubkey(d):
    P = p256.generator().mul(p256.newScalar(d))
    return P.toBytes()


def ecdsa_sign(d, msgHash, k=None):
    n = p256.order
    z = truncateToN(fromBytes(msgHash), n)
    if k is None:
        k = rfc6979_k(d, msgHash)
    x, _ = p256.generator().mul(p256.newScalar(k)).toAffine()
    r = x % n
    s = (pow(k, -1, n) * (z + r * d)) % n
    assert r != 0 and s != 0
    return toBytes(r, 32) + toBytes(s, 32)


def ecdsa_verify(pkBytes, msgHash, sig):
    n = p256.order
    Q = p256.deserializePoint(pkBytes)
    r, s = fromBytes(sig[:32]), fromBytes(sig[32:])
    if not (1 <= r < n and 1 <= s < n):
        return False
    z = truncateToN(fromBytes(msgHash), n)
    w = pow(s, -1, n)
    X = p256.generator().dblmul(p256.newScalar(z * w), Q, p256.newScalar(r * w))
    c = X.toAffine()
    return bool(c) and c[0] % n == r


# --------------------------------------------------------------------------------------
# Deterministic synthetic inputs; mirrored by the C oracle and the
# engine's own generator.  tag(b'..', S, i) = SHA-256(tag || be64(S) || be64(i)).
# --------------------------------------------------------------------------------------


def synth_tag(tag, S, i):
    return hashlib.sha256(tag + S.to_bytes(8, 'big') + i.to_bytes(8, 'big')).digest()


def synth_params(S, secLevel=80):
    """h = g * k with k = tag mod order  (pedersen.ts:61-69 with a fixed scalar)."""
    kn = fromBytes(synth_tag(b'hnist', S, 0)) % p256.order
    kt = fromBytes(synth_tag(b'htom', S, 0)) % tomEdwards256.order
    gn, gt = p256.generator(), tomEdwards256.generator()
    return SystemParametersList(PedersenParams(p256, gn, gn.mul(p256.newScalar(kn))),
                                PedersenParams(tomEdwards256, gt, gt.mul(tomEdwards256.newScalar(kt))), secLevel)


def synth_ring_fast(S, N):
    """Ring of N uniform values in [0,q) (the reference itself uses non-curve values, test/zkpAttestList.test.ts:37)."""
    q = tomEdwards256.order
    return [fromBytes(synth_tag(b'ring', S, i)) % q for i in range(N)]


def synth_proof_input(S, b, N):
    """Returns (msgHash, sig, pkBytes, which, d, seed) for proof b; caller sets ring[which] = pk.x."""
    n = p256.order
    d = fromBytes(synth_tag(b'sk', S, b)) % (n - 1) + 1
    msgHash = synth_tag(b'msg', S, b)
    k = fromBytes(synth_tag(b'nonce', S, b)) % (n - 1) + 1
    pk = ecdsa_pubkey(d)
    sig = ecdsa_sign(d, msgHash, k)
    seed = synth_tag(b'rng', S, b)
    return msgHash, sig, pk, b % N, d, seed


# ----------------------------------------------------------------------------------------------------------------------
# HARDENED MODE (not in the reference: its two TODOs, src/commit/pedersen.ts:62 and src/proofGK/gk.ts:178).  Restatement of
# the engine's own specification (include/zkattest.h, csrc/h2c_host.cpp, k_hash.hip) for the parity tests of that mode.
def _sqrt_3mod4(v, p):
    y = pow(v, (p + 1) // 4, p)
    return y if y * y % p == v % p else None


def hardened_h(tag=b''):
    """Nothing-up-my-sleeve second generators: (h_NIST, h_Tom) as affine (x, y) pairs, by try-and-increment over SHA-256."""
    dom = b'ZKAttest-NUMS-h-v1'
    p, b = p256.p, p256.b
    ctr = 0
    while True:
        x = int.from_bytes(hashlib.sha256(dom + b'\x01' + tag + ctr.to_bytes(4, 'big')).digest(), 'big') % p
        y = _sqrt_3mod4((x * x * x - 3 * x + b) % p, p)
        ctr += 1
        if y is not None:
            h_nist = (x, y if y % 2 == 0 else p - y)
            break
    t, a, d = tomEdwards256.p, tomEdwards256.a, tomEdwards256.d
    ctr = 0
    while True:
        m = dom + b'\x02' + tag + ctr.to_bytes(4, 'big')
        ctr += 1
        x = (int.from_bytes(hashlib.sha256(m + b'\x00').digest(), 'big') << 256 | int.from_bytes(hashlib.sha256(m + b'\x01').digest(), 'big')) % t
        den = (1 - d * x * x) % t
        if den == 0:
            continue
        y = _sqrt_3mod4((1 - a * x * x) * pow(den, -1, t) % t, t)
        if y is None:
            continue
        y = y if y % 2 == 0 else t - y
        P = TEdwardsPoint(tomEdwards256, x, y)
        Q = P.dbl().dbl()
        qx, qy = Q.toAffine()
        if qx == 0:
            continue
        return h_nist, (qx, qy)


GK_STATEMENT_TAG = b'ZKAttest-GK-statement-v1'


def ring_digest(values):
    """SHA-256('ZKAttest-ring-v1' || be64(N) || leaf_0 || leaf_1 ...), leaf_i = SHA-256 of 256 consecutive padded ring entries
    (32-byte big-endian each; the last leaf may be shorter when N < 256)."""
    N = len(values)
    leaves = b''.join(hashlib.sha256(b''.join(int(v).to_bytes(32, 'big') for v in values[i:i + 256])).digest() for i in range(0, N, 256))
    return hashlib.sha256(b'ZKAttest-ring-v1' + N.to_bytes(8, 'big') + leaves).digest()


def gk_statement(padded_values, msgHash, R, keyXcom):
    """Bytes appended to the Groth-Kohlweiss transcript in hardened mode: the statement the membership proof is about."""
    return GK_STATEMENT_TAG + ring_digest(padded_values) + bytes(msgHash) + R.toBytes() + keyXcom.toBytes()
