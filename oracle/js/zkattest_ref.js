// ORACLE (test infrastructure, NOT product code) -- plain JavaScript (BigInt) restatement of the proveSignatureList /
// verifySignatureList path of cloudflare/zkp-ecdsa, written from this build's Python restatement (oracle/zkattest_ref.py) and
// checked against the same committed golden vectors (tests/golden/golden.json).  Purpose (SURVEY.md section 8(d)): a third,
// independent restatement in the reference's own arithmetic (V8 BigInt: '/' truncates, '%' follows the dividend) that runs on
// the Node 12 of this image, and a V8-BigInt timing of one proof as an approximation of `npm run bench` -- the reference itself
// needs Node >= 24, tsc and typedjson and cannot run here.  PARITY STATUS: "parity unpinned" (see oracle/zkattest_ref.py).
// Only tests/ and bench tooling may execute this file.  `file:line` citations are into /root/reference/src.
//
//   node oracle/js/zkattest_ref.js golden tests/golden/golden.json [case ...]   -> one JSON line per proof (sha256 match, ms)
//   node oracle/js/zkattest_ref.js bench  tests/golden/golden.json [case]       -> prove + verify timing of the case's first proof
'use strict'
const crypto = require('crypto')
const fs = require('fs')

// ---------------------------------------------------------------- bignum/big.ts
const bitLen = (n) => (n > 0n ? n.toString(2).length : 1) // big.ts:23-25
const byteLen = (n) => Math.floor((bitLen(n) + 7) / 8) // big.ts:26-28
function posMod(n, p) { // big.ts:36-42
    const r = n % p
    return r < 0n ? r + p : r
}
function expMod(n, e, p) { // big.ts:44-59
    if (e < 0n) throw new Error('neg expo')
    let r = 1n, q = n, k = e
    while (k > 0n) {
        if (k & 1n) r = (r * q) % p
        q = (q * q) % p
        k >>= 1n
    }
    return r
}
function extendedEuclid(X, Y) { // big.ts:80-111
    let a = 1n, b = 0n, c = 0n, d = 1n, x = X, y = Y
    while (y !== 0n) {
        const q = x / y
        a -= c * q
        b -= d * q
        x -= q * y
        let t = x; x = y; y = t
        t = a; a = c; c = t
        t = b; b = d; d = t
    }
    return [x, a, b]
}
function invMod(t, N) { // big.ts:76-78,113-119: invMod(0, N) is 0
    let inv = extendedEuclid(t, N)[1]
    if (inv < 0n) inv += N
    return inv
}
function toBytes(n, length) { // big.ts:121-134
    if (!(length > 0) || n < 0n || n >> BigInt(8 * length) !== 0n) throw new Error("number doesn't fit in array")
    return Buffer.from(n.toString(16).padStart(2 * length, '0'), 'hex')
}
const fromBytes = (a) => (a.length ? BigInt('0x' + Buffer.from(a).toString('hex')) : 0n) // big.ts:161-168

function be64(k) {
    const b = Buffer.alloc(8)
    b.writeUInt32BE(Math.floor(k / 4294967296), 0)
    b.writeUInt32BE(k >>> 0, 4)
    return b
}
const sha256 = (...parts) => { const h = crypto.createHash('sha256'); for (const p of parts) h.update(p); return h.digest() }
// RNG contract (replaces crypto.getRandomValues of big.ts:171-181): fill k of a proof with seed S is SHA-256(S || be64(k))
class SeedRng {
    constructor(seed) { this.seed = Buffer.from(seed); this.k = 0 }
    fill(n) { if (n !== 32) throw new Error('RNG contract defines 32-byte fills only'); return sha256(this.seed, be64(this.k++)) }
}
class StreamRng { // explicit blocks (rejection-path vectors)
    constructor(blocks) { this.blocks = blocks; this.k = 0 }
    fill(n) { if (n !== 32) throw new Error('32-byte fills only'); return this.blocks[this.k++] }
}
class OsRng { // verifier-side randomness (randomisers, generateIndices): any length, not part of the contract
    constructor(seed) { this.state = sha256(Buffer.from(seed || 'verifier')); this.k = 0 }
    fill(n) {
        let out = Buffer.alloc(0)
        while (out.length < n) out = Buffer.concat([out, sha256(this.state, be64(this.k++))])
        return out.slice(0, n)
    }
}
function rnd(n, rng) { // big.ts:171-181: redraw the whole fill until it is below n
    const len = byteLen(n)
    for (;;) {
        const v = fromBytes(rng.fill(len))
        if (v < n) return v
    }
}
const rndRange = (lo, hi, rng) => Number(rnd(BigInt(hi - lo + 1), rng)) + lo // big.ts:183-185

// ---------------------------------------------------------------- curves/group.ts
class Scalar { // group.ts:155-218
    constructor(group, s) { this.group = group; this.k = s ? posMod(s, group.order) : 0n }
    add(o) { return new Scalar(this.group, this.k + o.k) }
    sub(o) { return new Scalar(this.group, this.k - o.k) }
    mul(o) { return new Scalar(this.group, this.k * o.k) }
    neg() { return new Scalar(this.group, -this.k) }
    isZero() { return this.k === 0n }
    cmp(o) { return this.k < o.k ? -1 : this.k > o.k ? 1 : 0 }
}
const DIGITS = '0123456789abcdef'
class Point {
    sub(pt) { return this.add(pt.neg()) } // group.ts:94-96
    multiples() { // 0*P .. 15*P keyed by hex digit
        const t = {}
        let cur = this.group.identity()
        for (const dgt of DIGITS) { t[dgt] = cur; cur = cur.add(this) }
        return t
    }
    dblmul(s1, p2, s2) { // group.ts:97-132: window-4 Straus over the hex strings of both scalars
        const m1 = this.multiples(), m2 = p2.multiples()
        let k1 = s1.k.toString(16), k2 = s2.k.toString(16)
        k1 = k1.padStart(k2.length, '0')
        k2 = k2.padStart(k1.length, '0')
        let q = this.group.identity()
        for (let i = 0; i < k1.length; i++) {
            q = q.dbl().dbl().dbl().dbl()
            q = q.add(m1[k1[i]]).add(m2[k2[i]])
        }
        return q
    }
    mul(s) { // group.ts:133-152
        const m = this.multiples()
        let q = this.group.identity()
        for (const dgt of s.k.toString(16)) q = q.dbl().dbl().dbl().dbl().add(m[dgt])
        return q
    }
}
class Group {
    sizeFieldBytes() { return Math.floor((bitLen(this.p) + 7) / 8) } // group.ts:49-52
    newScalar(s) { return new Scalar(this, s) }
    randomScalar(rng) { return this.newScalar(rnd(this.order, rng)) } // group.ts:59-61
}

// ---------------------------------------------------------------- curves/weier.ts (a = -3, Renes-Costello-Batina complete formulas)
class WeierstrassGroup extends Group { // weier.ts:25-89
    constructor(name, p, a, b, order, gx, gy) { super(); Object.assign(this, { name, p, a, b, order, gx, gy }) }
    identity() { return new WPoint(this, 0n, 1n, 0n) }
    generator() { return new WPoint(this, this.gx, this.gy, 1n) }
    isOnGroup(pt) { // weier.ts:56-70
        const { p, a, b } = this, { x, y, z } = pt
        const lhs = (((y * y) % p) * z) % p, z2 = (z * z) % p
        const rhs = (x * x * x) % p + (((a * x) % p) * z2) % p + (b * ((z2 * z) % p)) % p
        return pt.group === this && posMod(lhs - rhs, p) === 0n
    }
    fromAffine(x, y) {
        const pt = new WPoint(this, x, y, 1n)
        if (!this.isOnGroup(pt)) throw new Error('point not in group')
        return pt
    }
}
class WPoint extends Point { // weier.ts:96-261
    constructor(g, x, y, z) { super(); this.group = g; this.x = x; this.y = y; this.z = z }
    isIdentity() { return this.x === 0n && this.y !== 0n && this.z === 0n } // weier.ts:117-119
    eq(o) { // weier.ts:120-128
        const p = this.group.p
        return this.group === o.group && posMod(this.x * o.z - o.x * this.z, p) === 0n && posMod(this.y * o.z - o.y * this.z, p) === 0n
    }
    neg() { return new WPoint(this.group, this.x, posMod(-this.y, this.group.p), this.z) }
    dbl() { // weier.ts:133-175
        const { x, y, z } = this, { p, b } = this.group
        const m = (v) => posMod(v, p)
        let t0 = m(x * x), t1 = m(y * y), t2 = m(z * z), t3 = m(2n * x * y), z3 = m(2n * x * z)
        let y3 = m(b * t2 - z3)
        y3 = m(3n * y3)
        let x3 = m(t1 - y3)
        y3 = m(t1 + y3)
        y3 = m(x3 * y3)
        x3 = m(x3 * t3)
        t2 = m(3n * t2)
        z3 = m(b * z3 - t2 - t0)
        z3 = m(3n * z3)
        t0 = m(3n * t0 - t2)
        t0 = m(t0 * z3)
        y3 = m(y3 + t0)
        t0 = m(2n * y * z)
        x3 = m(x3 - t0 * z3)
        z3 = m(4n * t0 * t1)
        return new WPoint(this.group, x3, y3, z3)
    }
    add(o) { // weier.ts:176-230
        const { p, b } = this.group
        const m = (v) => posMod(v, p)
        const x1 = this.x, y1 = this.y, z1 = this.z, x2 = o.x, y2 = o.y, z2 = o.z
        let t0 = m(x1 * x2), t1 = m(y1 * y2), t2 = m(z1 * z2)
        const t3 = m((x1 + y1) * (x2 + y2) - t0 - t1)
        const t4 = m((y1 + z1) * (y2 + z2) - t1 - t2)
        let y3 = m((x1 + z1) * (x2 + z2) - t0 - t2)
        let x3 = m(y3 - b * t2)
        x3 = m(3n * x3)
        let z3 = m(t1 - x3)
        x3 = m(t1 + x3)
        t2 = m(3n * t2)
        y3 = m(b * y3 - t2 - t0)
        y3 = m(3n * y3)
        t0 = m(3n * t0 - t2)
        t1 = m(t4 * y3)
        t2 = m(t0 * y3)
        y3 = m(x3 * z3 + t2)
        x3 = m(t3 * x3 - t1)
        z3 = m(t4 * z3 + t3 * t0)
        return new WPoint(this.group, x3, y3, z3)
    }
    toAffine() { // weier.ts:231-243: normalises in place, false for the identity
        if (this.isIdentity()) { this.y = 1n; return false }
        const p = this.group.p, zi = invMod(this.z, p)
        this.x = posMod(this.x * zi, p); this.y = posMod(this.y * zi, p); this.z = 1n
        return [this.x, this.y]
    }
    toBytes() { // weier.ts:244-255
        const c = this.toAffine()
        if (!c) return Buffer.alloc(1)
        const cs = this.group.sizeFieldBytes()
        return Buffer.concat([Buffer.from([4]), toBytes(c[0], cs), toBytes(c[1], cs)])
    }
}

// ---------------------------------------------------------------- curves/edwards.ts (extended coordinates, Hisil et al.)
class TEdwards extends Group { // edwards.ts:25-86
    constructor(name, p, a, d, order, gx, gy) { super(); Object.assign(this, { name, p, a, d, order, gx, gy }) }
    identity() { return new EPoint(this, 0n, 1n, 0n, 1n) }
    generator() { return new EPoint(this, this.gx, this.gy, posMod(this.gx * this.gy, this.p), 1n) }
    isOnGroup(pt) { // edwards.ts:52-65
        const { p, a, d } = this, { x, y, t, z } = pt
        return pt.group === this && posMod(a * x * x + y * y - z * z - d * t * t, p) === 0n && posMod(x * y - z * t, p) === 0n
    }
    fromAffine(x, y) { // edwards.ts:70-86 without the byte framing
        if (x < 0n || x >= this.p || y < 0n || y >= this.p) throw new Error('a not in range')
        const pt = new EPoint(this, x, y, posMod(x * y, this.p), 1n)
        if (!this.isOnGroup(pt)) throw new Error('point not on TEdwards group')
        return pt
    }
}
class EPoint extends Point { // edwards.ts:93-210
    constructor(g, x, y, t, z) { super(); this.group = g; this.x = x; this.y = y; this.t = t; this.z = z }
    isIdentity() { return this.x === 0n && this.y !== 0n && this.t === 0n && this.z !== 0n && this.y === this.z } // edwards.ts:117-125
    eq(o) { // edwards.ts:126-135
        const p = this.group.p
        return this.group === o.group && posMod(this.x * o.z - o.x * this.z, p) === 0n && posMod(this.y * o.z - o.y * this.z, p) === 0n
    }
    neg() { const p = this.group.p; return new EPoint(this.group, posMod(-this.x, p), this.y, posMod(-this.t, p), this.z) }
    dbl() { // edwards.ts:141-160
        const { x, y, z } = this, { p, a } = this.group
        const A = (x * x) % p, B = (y * y) % p, C = (2n * z * z) % p, D = (a * A) % p
        const E = posMod((x + y) * (x + y) - A - B, p), G = (D + B) % p, F = posMod(G - C, p), H = posMod(D - B, p)
        return new EPoint(this.group, (E * F) % p, (G * H) % p, (E * H) % p, (F * G) % p)
    }
    add(o) { // edwards.ts:161-183
        const { p, a, d } = this.group
        const A = (this.x * o.x) % p, B = (this.y * o.y) % p, C = (d * this.t * o.t) % p, D = (this.z * o.z) % p
        const E = posMod((this.x + this.y) * (o.x + o.y) - A - B, p), F = posMod(D - C, p), G = (D + C) % p, H = posMod(B - a * A, p)
        return new EPoint(this.group, (E * F) % p, (G * H) % p, (E * H) % p, (F * G) % p)
    }
    toAffine() { // edwards.ts:184-193
        const p = this.group.p, zi = invMod(this.z, p)
        this.x = posMod(this.x * zi, p); this.y = posMod(this.y * zi, p); this.t = posMod(this.x * this.y, p); this.z = 1n
        return [this.x, this.y]
    }
    toBytes() { // edwards.ts:194-203
        const c = this.toAffine(), cs = this.group.sizeFieldBytes()
        return Buffer.concat([Buffer.from([4]), toBytes(c[0], cs), toBytes(c[1], cs)])
    }
}
// instances.ts:22-54 (data)
const p256 = new WeierstrassGroup('p256',
    0xffffffff00000001000000000000000000000000ffffffffffffffffffffffffn, 0xffffffff00000001000000000000000000000000fffffffffffffffffffffffcn,
    0x5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604bn, 0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551n,
    0x6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296n, 0x4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5n)
const tom256 = new TEdwards('tomEdwards256',
    0x3fffffffc000000040000000000000002ae382c7957cc4ff9713c3d82bc47d3afn, 0x1abce3fd8e1d7a21252515332a512e09d4249bd5b1ec35e316c02254fe8cedf5dn,
    0x051781d9823abde00ec99295ba542c8b1401874bcbeb9e9c861174c7bca6a02aan, 0x0ffffffff00000001000000000000000000000000ffffffffffffffffffffffffn,
    0x7907055d0a7d4abc3eafdc25d431d9659fbe007ee2d8ddc4e906206ea9ba4fdbn, 0xbe231cb9f9bf18319c9f081141559b0a33dddccd2221f0464a9cd57081b01a01n)

const hashPoints = (pts) => fromBytes(sha256(...pts.map((q) => q.toBytes())).slice(0, 10)) // group.ts:221-233

// ---------------------------------------------------------------- curves/multimult.ts (verifier only)
function siftUp(h, i) { // multimult.ts:113-124, 1-based heap on scalar size
    while (i > 1) {
        const up = i >> 1
        if (h[up - 1].s.cmp(h[i - 1].s) >= 0) return
        const t = h[up - 1]; h[up - 1] = h[i - 1]; h[i - 1] = t
        i = up
    }
}
function siftDown(h, i) { // multimult.ts:126-145
    for (;;) {
        let c = 2 * i
        if (c > h.length) return
        if (c + 1 <= h.length && h[c].s.cmp(h[c - 1].s) > 0) c++
        if (h[i - 1].s.cmp(h[c - 1].s) >= 0) return
        const t = h[c - 1]; h[c - 1] = h[i - 1]; h[i - 1] = t
        i = c
    }
}
class MultiMult { // multimult.ts:31-90
    constructor(g) { this.group = g; this.pairs = []; this.known = [] }
    addKnown(pt) {
        if (this.known.some((k) => pt.eq(k.pt))) return
        this.pairs.push({ pt, s: this.group.newScalar(0n) })
        this.known.push({ pt, at: this.pairs.length - 1 })
    }
    insert(pt, s) {
        for (const k of this.known) if (pt.eq(k.pt)) { this.pairs[k.at].s = this.pairs[k.at].s.add(s); return }
        this.pairs.push({ pt, s })
    }
    evaluate() { // Bos-Coster, multimult.ts:61-89
        const h = this.pairs
        if (!h.length) return this.group.identity()
        for (let i = 1; i <= h.length; i++) siftUp(h, i)
        for (;;) {
            if (h.length === 1) return h[0].pt.mul(h[0].s)
            const t = h[0]; h[0] = h[h.length - 1]; h[h.length - 1] = t // multimult.ts:92-104
            const a = h.pop()
            siftDown(h, 1)
            const b = h[0]
            if (b.s.isZero()) return a.pt.mul(a.s)
            const rest = a.s.sub(b.s)
            h[0] = { pt: b.pt.add(a.pt), s: b.s }
            if (!rest.isZero()) { h.push({ pt: a.pt, s: rest }); siftUp(h, h.length) }
        }
    }
}
class Relation { // multimult.ts:147-174
    constructor(g) { this.group = g; this.terms = [] }
    insert(pt, s) { this.terms.push([pt, s]) }
    insertM(pts, ss) {
        if (pts.length !== ss.length) throw new Error('arrays are not the same length')
        pts.forEach((pt, i) => this.insert(pt, ss[i]))
    }
    drain(mm, vrng) {
        const rho = this.group.randomScalar(vrng)
        for (const [pt, s] of this.terms) mm.insert(pt, s.mul(rho))
    }
}

// ---------------------------------------------------------------- commit/pedersen.ts, equality.ts, mult.ts
class Commitment { // pedersen.ts:21-36
    constructor(p, r) { this.p = p; this.r = r }
    add(c) { return new Commitment(this.p.add(c.p), this.r.add(c.r)) }
    sub(c) { return new Commitment(this.p.sub(c.p), this.r.sub(c.r)) }
}
class PedersenParams { // pedersen.ts:40-58: the blinder is drawn first
    constructor(c, g, h) { this.c = c; this.g = g; this.h = h }
    commit(value, rng) {
        const r = this.c.randomScalar(rng)
        return new Commitment(this.h.dblmul(r, this.g, this.c.newScalar(value)), r)
    }
}
function proveEquality(pp, x, C1, C2, rng) { // equality.ts:60-78
    const S = (v) => pp.c.newScalar(v)
    const k = rnd(pp.c.order, rng)
    const A1 = pp.commit(k, rng), A2 = pp.commit(k, rng)
    const c = S(hashPoints([C1.p, C2.p, A1.p, A2.p]))
    return { A_1: A1.p, A_2: A2.p, t_x: S(k).sub(c.mul(S(x))), t_r1: A1.r.sub(c.mul(C1.r)), t_r2: A2.r.sub(c.mul(C2.r)) }
}
function aggregateEquality(pp, C1, C2, pi, mm, vrng) { // equality.ts:94-116
    const c = pp.c.newScalar(hashPoints([C1, C2, pi.A_1, pi.A_2])), one = pp.c.newScalar(1n)
    const r1 = new Relation(pp.c), r2 = new Relation(pp.c)
    r1.insertM([pp.g, pp.h, C1, pi.A_1.neg()], [pi.t_x, pi.t_r1, c, one])
    r2.insertM([pp.g, pp.h, C2, pi.A_2.neg()], [pi.t_x, pi.t_r2, c, one])
    r1.drain(mm, vrng)
    r2.drain(mm, vrng)
    return true
}
function proveMult(pp, x, y, z, Cx, Cy, Cz, rng) { // mult.ts:93-131
    const S = (v) => pp.c.newScalar(v), q = pp.c.order
    const xs = S(x), C4 = Cy.p.mul(xs), r4 = Cy.r.mul(xs)
    const kx = rnd(q, rng), ky = rnd(q, rng), kz = rnd(q, rng)
    const Ax = pp.commit(kx, rng), Ay = pp.commit(ky, rng), Az = pp.commit(kz, rng), A41 = pp.commit(kz, rng)
    const A42 = Cy.p.mul(S(kx))
    const c = S(hashPoints([Cx.p, Cy.p, Cz.p, C4, Ax.p, Ay.p, Az.p, A41.p, A42]))
    return {
        C_4: C4, A_x: Ax.p, A_y: Ay.p, A_z: Az.p, A_4_1: A41.p, A_4_2: A42,
        t_x: S(kx).sub(c.mul(xs)), t_y: S(ky).sub(c.mul(S(y))), t_z: S(kz).sub(c.mul(S(z))),
        t_rx: Ax.r.sub(c.mul(Cx.r)), t_ry: Ay.r.sub(c.mul(Cy.r)), t_rz: Az.r.sub(c.mul(Cz.r)), t_r4: A41.r.sub(c.mul(r4)),
    }
}
function aggregateMult(pp, Cx, Cy, Cz, pi, mm, vrng) { // mult.ts:148-175: all five relations are built, then drained in order
    const c = pp.c.newScalar(hashPoints([Cx, Cy, Cz, pi.C_4, pi.A_x, pi.A_y, pi.A_z, pi.A_4_1, pi.A_4_2])), one = pp.c.newScalar(1n)
    const rows = [
        [[pp.g, pp.h, Cx, pi.A_x.neg()], [pi.t_x, pi.t_rx, c, one]],
        [[pp.g, pp.h, Cy, pi.A_y.neg()], [pi.t_y, pi.t_ry, c, one]],
        [[pp.g, pp.h, Cz, pi.A_z.neg()], [pi.t_z, pi.t_rz, c, one]],
        [[pp.g, pp.h, pi.C_4, pi.A_4_1.neg()], [pi.t_z, pi.t_r4, c, one]],
        [[Cy, pi.C_4, pi.A_4_2.neg()], [pi.t_x, c, one]],
    ].map(([pts, ss]) => { const r = new Relation(pp.c); r.insertM(pts, ss); return r })
    rows.forEach((r) => r.drain(mm, vrng))
    return true
}

// ---------------------------------------------------------------- exp/pointAdd.ts
function provePointAdd(pp, P, Q, R, PX, PY, QX, QY, RX, RY, rng) { // pointAdd.ts:92-163
    if (!P.add(Q).eq(R)) throw new Error("Points don't add up!")
    const q = pp.c.order
    const cP = P.toAffine(), cQ = Q.toAffine(), cR = R.toAffine()
    if (!cP) throw new Error('P is at infinity')
    if (!cQ) throw new Error('Q is at infinity')
    if (!cR) throw new Error('R is at infinity')
    const [x1, y1] = cP, [x2, y2] = cQ, x3 = cR[0]
    const i7 = posMod(x2 - x1, q), i8 = invMod(i7, q), i9 = posMod(y2 - y1, q), i10 = posMod(i8 * i9, q)
    const i11 = posMod(i10 * i10, q), i12 = posMod(x1 - x3, q), i13 = posMod(i10 * i12, q)
    const C7 = QX.sub(PX)
    const C8 = pp.commit(i8, rng)
    const C9 = QY.sub(PY)
    const C10 = pp.commit(i10, rng), C11 = pp.commit(i11, rng)
    const C12 = PX.sub(RX)
    const C13 = pp.commit(i13, rng)
    const C14 = new Commitment(pp.g, pp.c.newScalar(0n)) // commits to 1 with blinder 0 (pointAdd.ts:144)
    const pi_8 = proveMult(pp, i7, i8, 1n, C7, C8, C14, rng)
    const pi_10 = proveMult(pp, i8, i9, i10, C8, C9, C10, rng)
    const pi_11 = proveMult(pp, i10, i10, i11, C10, C10, C11, rng)
    const pi_x = proveEquality(pp, i11, C11, new Commitment(RX.p.add(PX.p).add(QX.p), RX.r.add(PX.r).add(QX.r)), rng)
    const pi_13 = proveMult(pp, i10, i12, i13, C10, C12, C13, rng)
    const pi_y = proveEquality(pp, i13, C13, new Commitment(RY.p.add(PY.p), RY.r.add(PY.r)), rng)
    return { C_8: C8.p, C_10: C10.p, C_11: C11.p, C_13: C13.p, pi_8, pi_10, pi_11, pi_13, pi_x, pi_y }
}
function aggregatePointAdd(pp, PX, PY, QX, QY, RX, RY, pi, mm, vrng) { // pointAdd.ts:199-259
    const C7 = QX.sub(PX), C9 = QY.sub(PY), C12 = PX.sub(RX)
    return aggregateMult(pp, C7, pi.C_8, pp.g, pi.pi_8, mm, vrng) && aggregateMult(pp, pi.C_8, C9, pi.C_10, pi.pi_10, mm, vrng) &&
        aggregateMult(pp, pi.C_10, pi.C_10, pi.C_11, pi.pi_11, mm, vrng) && aggregateEquality(pp, pi.C_11, RX.add(PX).add(QX), pi.pi_x, mm, vrng) &&
        aggregateMult(pp, pi.C_10, C12, pi.C_13, pi.pi_13, mm, vrng) && aggregateEquality(pp, pi.C_13, PY.add(RY), pi.pi_y, mm, vrng)
}

// ---------------------------------------------------------------- exp/exp.ts
function generateIndices(limit, vrng) { // exp.ts:95-109 (the slice to indnum is discarded by the reference)
    const idx = Array.from({ length: limit }, (_, i) => i)
    for (let i = 0; i < limit - 2; i++) {
        const j = rndRange(i, limit - 1, vrng), t = idx[i]
        idx[i] = idx[j]; idx[j] = t
    }
    return idx
}
function proveExp(ppN, ppW, s, Cs, P, Px, Py, sec, rng, Q) { // exp.ts:126-231
    const alpha = [], r = [], T = [], A = [], Tx = [], Ty = []
    for (let i = 0; i < sec; i++) {
        alpha.push(ppN.c.randomScalar(rng))
        r.push(ppN.c.randomScalar(rng))
        T.push(ppN.g.mul(alpha[i]))
        A.push(T[i].add(ppN.h.mul(r[i])))
        const c = T[i].toAffine()
        if (!c) throw new Error('T[i] is at infinity')
        Tx.push(ppW.commit(c[0], rng))
        Ty.push(ppW.commit(c[1], rng))
    }
    const pts = [Px.p, Py.p]
    for (let i = 0; i < sec; i++) pts.push(A[i], Tx[i].p, Ty[i].p)
    let chal = hashPoints(pts)
    const out = []
    for (let i = 0; i < sec; i++, chal >>= 1n) {
        if (chal & 1n) { out.push({ A: A[i], Tx: Tx[i].p, Ty: Ty[i].p, alpha: alpha[i], beta1: r[i], beta2: Tx[i].r, beta3: Ty[i].r }); continue }
        const z = alpha[i].sub(ppN.c.newScalar(s))
        let T1 = ppN.g.mul(z)
        if (Q) T1 = T1.add(Q)
        const c1 = T1.toAffine()
        if (!c1) throw new Error('T1 is at infinity')
        const T1x = ppW.commit(c1[0], rng), T1y = ppW.commit(c1[1], rng)
        const proof = provePointAdd(ppW, T1, P, T[i], T1x, T1y, Px, Py, Tx[i], Ty[i], rng)
        out.push({ A: A[i], Tx: Tx[i].p, Ty: Ty[i].p, z, z2: r[i].sub(Cs.r), proof, r1: T1x.r, r2: T1y.r })
    }
    return out
}
function verifyExp(ppN, ppW, Cl, Px, Py, pi, sec, vrng, Q) { // exp.ts:233-349
    if (sec > pi.length) throw new Error('security level not achieved')
    const cN = ppN.c, cW = ppW.c, mW = new MultiMult(cW), mN = new MultiMult(cN)
    mW.addKnown(ppW.g); mW.addKnown(ppW.h)
    mN.addKnown(ppN.g); mN.addKnown(ppN.h); mN.addKnown(Cl)
    const pts = [Px, Py]
    for (const e of pi) pts.push(e.A, e.Tx, e.Ty)
    const chal = hashPoints(pts), idx = generateIndices(pi.length, vrng)
    const oneN = () => cN.newScalar(1n), oneW = () => cW.newScalar(1n)
    for (let j = 0; j < sec; j++) {
        const i = idx[j], e = pi[i]
        if ((chal >> BigInt(i)) & 1n) {
            if (!(e.alpha && e.beta1 && e.beta2 && e.beta3)) throw new Error('params not found')
            const T = ppN.g.mul(e.alpha), rA = new Relation(cN)
            rA.insertM([T, ppN.h, e.A.neg()], [oneN(), e.beta1, oneN()])
            rA.drain(mN, vrng)
            const c = T.toAffine()
            if (!c) throw new Error('T is at infinity')
            const rx = new Relation(cW), ry = new Relation(cW)
            rx.insertM([ppW.g, ppW.h, e.Tx.neg()], [cW.newScalar(c[0]), e.beta2, oneW()])
            ry.insertM([ppW.g, ppW.h, e.Ty.neg()], [cW.newScalar(c[1]), e.beta3, oneW()])
            rx.drain(mW, vrng)
            ry.drain(mW, vrng)
        } else {
            if (!(e.z && e.z2 && e.proof && e.r1 && e.r2)) throw new Error('params not found')
            let T1 = ppN.g.mul(e.z)
            const rA = new Relation(cN)
            rA.insertM([T1, Cl, e.A.neg(), ppN.h], [oneN(), oneN(), oneN(), e.z2])
            rA.drain(mN, vrng)
            if (Q) T1 = T1.add(Q)
            const c1 = T1.toAffine()
            if (!c1) throw new Error('T1 is at infinity')
            const T1x = ppW.g.dblmul(cW.newScalar(c1[0]), ppW.h, e.r1), T1y = ppW.g.dblmul(cW.newScalar(c1[1]), ppW.h, e.r2)
            if (!aggregatePointAdd(ppW, T1x, T1y, Px, Py, e.Tx, e.Ty, e.proof, mW, vrng)) return false
        }
    }
    return mW.evaluate().isIdentity() && mN.evaluate().isIdentity()
}

// ---------------------------------------------------------------- proofGK/interpolate.ts, gk.ts
function evalPoly(co, x, m) { // interpolate.ts:19-25
    let r = 0n
    for (let i = co.length - 1; i >= 0; i--) r = posMod(co[i] + x * r, m)
    return r
}
function interpolate(x, y, m) { // interpolate.ts:27-70; '%' here is the BigInt remainder (sign of the dividend), as in the reference
    if (x.length !== y.length) throw new Error('inconsistent args')
    const n = x.length, s = new Array(n + 1).fill(0n), co = new Array(n).fill(0n)
    s[n] = 1n
    s[n - 1] = -x[0] % m
    for (let i = 1; i < n; i++) {
        for (let j = n - i - 1; j < n - 1; j++) s[j] = (s[j] - x[i] * s[j + 1]) % m
        s[n - 1] = (s[n - 1] - x[i]) % m
    }
    for (let i = 0; i < n; i++) {
        let phi = 0n
        for (let j = n; j > 0; j--) phi = BigInt(j) * s[j] + x[i] * phi
        const ff = invMod(posMod(phi, m), m) % m
        let b = 1n
        for (let j = n - 1; j >= 0; j--) {
            co[j] = posMod(co[j] + b * ff * y[i], m)
            b = s[j] + x[i] * b
        }
    }
    for (let i = 0; i < n; i++) if (y[i] !== evalPoly(co, x[i], m)) throw new Error('incorrect interpolation')
    return co
}
const ceilLog2 = (n) => Math.ceil(Math.log2(n))
function pad(vals, c) { // gk.ts:75-86: up to the next power of two with copies of the first key
    const out = vals.map((v) => c.newScalar(v))
    for (let i = vals.length; i < 2 ** ceilLog2(vals.length); i++) out.push(out[0])
    return out
}
const gkCommit = (pp, v, blind) => pp.g.dblmul(pp.c.newScalar(posMod(v, pp.c.order)), pp.h, pp.c.newScalar(posMod(blind, pp.c.order))) // gk.ts:88-92
function proveMembership(pp, com, index, keys, rng) { // gk.ts:94-195
    const c = pp.c, q = c.order, vals = pad(keys, c), n = ceilLog2(vals.length)
    const l = []
    for (let i = 0, t = index; i < n; i++, t = Math.floor(t / 2)) l.push(BigInt(t % 2))
    const r = [], a = [], s = [], t = [], rho = []
    for (let i = 0; i < n; i++) { r.push(rnd(q, rng)); a.push(rnd(q, rng)); s.push(rnd(q, rng)); t.push(rnd(q, rng)); rho.push(rnd(q, rng)) }
    const cl = [], ca = [], cb = [], cd = []
    for (let i = 0; i < n; i++) { cl.push(gkCommit(pp, l[i], r[i])); ca.push(gkCommit(pp, a[i], s[i])); cb.push(gkCommit(pp, l[i] * a[i], t[i])) }
    const omegas = Array.from({ length: n }, (_, i) => BigInt(i)), dv = []
    for (const w of omegas) { // d(w) = sum_i (v_l - v_i) p_i(w), p_i built by repeated multiplication with f1/f0 (gk.ts:135-166)
        const f0 = [], ratio = []
        for (let j = 0; j < n; j++) {
            f0.push(posMod((1n - l[j]) * w - a[j], q))
            ratio.push(posMod(posMod(l[j] * w + a[j], q) * invMod(f0[j], q), q))
        }
        const p = [f0.reduce((acc, v) => posMod(acc * v, q), 1n)]
        for (let i = 0; i < n; i++) for (let j = 0, len = p.length; j < len; j++) p.push(posMod(ratio[i] * p[j], q))
        let d = 0n
        const vl = vals[index].k
        for (let i = 0; i < vals.length; i++) d = posMod(d + (vl - vals[i].k) * p[i], q)
        dv.push(d)
    }
    const di = interpolate(omegas, dv, q)
    for (let i = 0; i < n; i++) cd.push(gkCommit(pp, di[i], rho[i]))
    const x = hashPoints([...cl, ...ca, ...cb, ...cd])
    const f = [], za = [], zb = []
    let zd = (com.r.k * expMod(x, BigInt(n), q)) % q
    for (let i = 0; i < n; i++) {
        f.push(c.newScalar(posMod(l[i] * x + a[i], q)))
        za.push(c.newScalar(posMod(r[i] * x + s[i], q)))
        zb.push(c.newScalar(posMod(r[i] * (x - f[i].k) + t[i], q)))
    }
    for (let i = 0; i < n; i++) zd = posMod(zd - rho[i] * expMod(x, BigInt(i), q), q)
    return { cl, ca, cb, cd, f, za, zb, zd: c.newScalar(zd) }
}
function verifyMembership(pp, com, keys, pf, vrng) { // gk.ts:197-262
    const c = pp.c, q = c.order, mm = new MultiMult(c), vec = pad(keys, c), n = ceilLog2(vec.length)
    if (![pf.cl, pf.ca, pf.cb, pf.cd, pf.f, pf.za, pf.zb].every((v) => v.length === n)) return false
    const x = hashPoints([...pf.cl, ...pf.ca, ...pf.cb, ...pf.cd]), one = () => c.newScalar(1n)
    mm.addKnown(pp.g); mm.addKnown(pp.h)
    for (let i = 0; i < n; i++) {
        const r0 = new Relation(c), r1 = new Relation(c)
        r0.insertM([pf.cl[i], pf.ca[i], pp.g, pp.h], [c.newScalar(x), one(), pf.f[i].neg(), pf.za[i].neg()])
        r0.drain(mm, vrng)
        r1.insertM([pf.cl[i], pf.cb[i], pp.h], [c.newScalar(posMod(x - pf.f[i].k, q)), one(), pf.zb[i].neg()])
        r1.drain(mm, vrng)
    }
    let total = 0n // gk.ts:239-250 as written: sum_i v_i * prod_j (i_j ? f_j : x - f_j)
    for (let i = 0; i < vec.length; i++) {
        let pix = 1n
        for (let j = 0; j < n; j++) pix = posMod(pix * ((i >> j) & 1 ? pf.f[j].k : x - pf.f[j].k), q)
        total = posMod(total + vec[i].k * pix, q)
    }
    const fin = new Relation(c)
    for (let i = 0; i < n; i++) fin.insert(pf.cd[i], c.newScalar(posMod(-expMod(x, BigInt(i), q), q)))
    fin.insert(com, c.newScalar(expMod(x, BigInt(n), q)))
    fin.insertM([pp.g, pp.h], [c.newScalar(posMod(-total, q)), pf.zd.neg()])
    fin.drain(mm, vrng)
    return mm.evaluate().isIdentity()
}

// ---------------------------------------------------------------- zkpAttestList.ts
function truncateToN(msg, n) { // zkpAttestList.ts:80-86
    const d = bitLen(msg) - bitLen(n)
    return d > 0 ? msg >> BigInt(d) : msg
}
function ecdsaParts(msgHash, sig) { // zkpAttestList.ts:113-123
    const n = p256.order, half = sig.length >> 1
    return { z: truncateToN(fromBytes(msgHash), n), r: fromBytes(sig.slice(0, half)), s: fromBytes(sig.slice(half)), n }
}
function proveSignatureList(params, msgHash, sig, pkXY, which, keys, rng) { // zkpAttestList.ts:104-145; pkXY = raw key without 0x04
    const pk = p256.fromAffine(fromBytes(pkXY.slice(0, 32)), fromBytes(pkXY.slice(32)))
    const pkc = pk.toAffine()
    if (!pkc) throw new Error('invalid public key')
    const { z, r, s, n } = ecdsaParts(msgHash, sig), S = (v) => p256.newScalar(v)
    const sinv = invMod(s, n), rinv = invMod(r, n)
    const R = p256.generator().mul(S(posMod(sinv * z, n))).add(pk.mul(S(posMod(sinv * r, n))))
    const s1 = posMod(rinv * s, n), Q = p256.generator().mul(S(posMod(rinv * z, n)))
    const ppSig = new PedersenParams(p256, R, params.nistH)
    const comS1 = ppSig.commit(s1, rng)
    const pkX = params.proof.commit(pkc[0], rng), pkY = params.proof.commit(pkc[1], rng)
    const expProof = proveExp(ppSig, params.proof, s1, comS1, pk, pkX, pkY, params.sec, rng, Q)
    const membershipProof = proveMembership(params.proof, pkX, which, keys, rng)
    return { R, comS1: comS1.p, keyXcom: pkX.p, keyYcom: pkY.p, expProof, membershipProof }
}
function verifySignatureList(params, msgHash, keys, proof, vrng) { // zkpAttestList.ts:147-184
    vrng = vrng || new OsRng()
    const n = p256.order, z = truncateToN(fromBytes(msgHash), n)
    const cR = proof.R.toAffine()
    if (!cR) throw new Error('R is at infinity')
    const Q = p256.generator().mul(p256.newScalar(posMod(invMod(cR[0], n) * z, n)))
    const ppSig = new PedersenParams(p256, proof.R, params.nistH)
    return verifyMembership(params.proof, proof.keyXcom, keys, proof.membershipProof, vrng) &&
        verifyExp(ppSig, params.proof, proof.comS1, proof.keyXcom, proof.keyYcom, proof.expProof, 20, vrng, Q)
}

// ---------------------------------------------------------------- ZKA1 bytes (include/zkattest.h): P-256 coordinate 32 B, Tom coordinate 36 B, scalar 32 B
function proofToBytes(pf) {
    const pp = (pt) => { const c = pt.toAffine(); return Buffer.concat([toBytes(c[0], 32), toBytes(c[1], 32)]) }
    const tp = (pt) => { const c = pt.toAffine(); return Buffer.concat([toBytes(c[0], 36), toBytes(c[1], 36)]) }
    const sc = (s) => toBytes(s.k, 32)
    const mult = (m) => [m.C_4, m.A_x, m.A_y, m.A_z, m.A_4_1, m.A_4_2].map(tp).concat([m.t_x, m.t_y, m.t_z, m.t_rx, m.t_ry, m.t_rz, m.t_r4].map(sc))
    const eq = (e) => [tp(e.A_1), tp(e.A_2), sc(e.t_x), sc(e.t_r1), sc(e.t_r2)]
    const parts = [pp(pf.R), pp(pf.comS1), tp(pf.keyXcom), tp(pf.keyYcom)]
    let bits = 0n
    pf.expProof.forEach((e, i) => {
        parts.push(pp(e.A), tp(e.Tx), tp(e.Ty))
        if (e.alpha) { bits |= 1n << BigInt(i); parts.push(sc(e.alpha), sc(e.beta1), sc(e.beta2), sc(e.beta3)); return }
        const a = e.proof
        parts.push(sc(e.z), sc(e.z2), sc(e.r1), sc(e.r2), tp(a.C_8), tp(a.C_10), tp(a.C_11), tp(a.C_13),
            ...mult(a.pi_8), ...mult(a.pi_10), ...mult(a.pi_11), ...mult(a.pi_13), ...eq(a.pi_x), ...eq(a.pi_y))
    })
    const g = pf.membershipProof
    parts.push(...[...g.cl, ...g.ca, ...g.cb, ...g.cd].map(tp), ...[...g.f, ...g.za, ...g.zb].map(sc), sc(g.zd))
    const body = Buffer.concat(parts), head = Buffer.alloc(16)
    head.write('ZKA1', 0, 'latin1')
    head.writeUInt32BE(32 + body.length, 4)
    head.writeUInt32BE(pf.expProof.length, 8)
    head.writeUInt32BE(g.cl.length, 12)
    return Buffer.concat([head, toBytes(bits, 16), body])
}

// ---------------------------------------------------------------- driver: the committed golden vectors
function loadCase(c) {
    const H = (s) => Buffer.from(s, 'hex'), half = (b) => [fromBytes(b.slice(0, b.length / 2)), fromBytes(b.slice(b.length / 2))]
    const [hx, hy] = half(H(c.nist_h)), [gx, gy] = half(H(c.tom_g)), [tx, ty] = half(H(c.tom_h))
    const params = { nistH: p256.fromAffine(hx, hy), proof: new PedersenParams(tom256, tom256.fromAffine(gx, gy), tom256.fromAffine(tx, ty)), sec: c.sec }
    return { params, ring: c.ring.map((v) => BigInt('0x' + v)) }
}
function rngFor(rec) {
    if (rec.seed) return new SeedRng(Buffer.from(rec.seed, 'hex'))
    const seed = Buffer.from(rec.stream_seed, 'hex'), blocks = []
    for (let k = 0; k < rec.stream_blocks; k++) blocks.push(sha256(seed, be64(k)))
    for (const [i, v] of rec.plant) blocks[i] = toBytes(BigInt('0x' + v), 32)
    return new StreamRng(blocks)
}
function main(argv) {
    const mode = argv[2], gold = JSON.parse(fs.readFileSync(argv[3], 'utf8'))
    if (mode === 'kats') { // test/bignum/big.test.ts:19-21, test/proofGK/interpolate.test.ts:19-26
        const ok = invMod(3n, 5n) === 2n && invMod(7n, 41n) === 6n && interpolate([1n, 2n, 3n], [1n, 2n, 3n], 401n).join() === '0,1,0'
        console.log(JSON.stringify({ kats: ok }))
        process.exit(ok ? 0 : 1)
    }
    const names = argv.length > 4 ? argv.slice(4) : ['small_full']
    let bad = 0
    for (const name of names) {
        const c = gold[name], { params, ring } = loadCase(c)
        c.proofs.forEach((rec, b) => {
            if (mode === 'bench' && b > 0) return
            const H = (s) => Buffer.from(s, 'hex'), rng = rngFor(rec)
            let t0 = Date.now()
            const pf = proveSignatureList(params, H(rec.msg), H(rec.sig), H(rec.pk), rec.which, ring, rng)
            const proveMs = Date.now() - t0, raw = proofToBytes(pf)
            t0 = Date.now()
            const ok = verifySignatureList(params, H(rec.msg), ring, pf)
            const line = { case: name, proof: b, sec: c.sec, nkeys: c.nkeys, len: raw.length, sha256_ok: sha256(raw).toString('hex') === rec.sha256 && raw.length === rec.len,
                           fills_ok: rng.k === rec.fills_consumed, verified: ok, prove_ms: proveMs, verify_ms: Date.now() - t0, node: process.version }
            if (!(line.sha256_ok && line.fills_ok && ok)) bad++
            console.log(JSON.stringify(line))
        })
    }
    process.exit(bad ? 1 : 0)
}
if (require.main === module) main(process.argv)
module.exports = { invMod, interpolate, proveSignatureList, verifySignatureList, proofToBytes, p256, tom256 }
