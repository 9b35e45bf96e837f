// JavaScript façade over the N-API addon with the reference's public surface (src/index.ts:17-19):
//
//   generateParamsList(secLevel = 80) -> SystemParametersList                      src/zkpAttestList.ts:88-92
//   keyToInt(publicKey) -> Promise<bigint>                                         src/zkpAttestList.ts:94-102
//   proveSignatureList(params, msgHash, sigBytes, publicKey, which, keys)          src/zkpAttestList.ts:104-145
//        -> Promise<SignatureProofList>
//   verifySignatureList(params, msgHash, keys, proof) -> Promise<boolean>          src/zkpAttestList.ts:147-184
//   writeJson(Class, object) -> string, readJson(Class, text) -> object            src/serde.ts:21-36
//   SignatureProofList, SystemParametersList, PedersenParams (each with eq())      src/zkpAttestList.ts:27-78
//   p256, tomEdwards256, ALL_GROUPS                                                src/curves/instances.ts:22-56
//
// plus the batch calls the engine is built for (proveSignatureListBatch / verifySignatureListBatch) and the low-level
// Engine.  Same argument order, same Promise-returning functions, same error texts as the reference.  The GPU contexts are
// created on first use and CACHED by the identity of (params, ring): building the fixed-base tables costs 0.2-1 s and the
// per-ring table a few ms, so they are built once per SystemParametersList / key ring, not once per call.
// Runs on the Node 12 of this image (the TypeScript sources of the reference need Node >= 24 and tsc).
'use strict'
const crypto = require('crypto')
const path = require('path')
const native = require(process.env.ZKATTEST_NODE || path.join(__dirname, 'zkattest.node'))

const STATUS_TEXT = { 1: 'point not in group', 2: 'invalid public key', 3: 'T[i] is at infinity', 4: 'T1 is at infinity', 5: 'P/Q/R is at infinity',
    6: "Points don't add up!", 7: 'R is at infinity', 8: 'params not found', 9: 'security level not achieved', 10: 'error deserializing' }
// The two places where the reference dies with a TypeError of the JavaScript runtime instead of one of its own errors (include/zkattest.h):
// `which` past the padded ring reads values[index].k of undefined (src/proofGK/gk.ts:162, engine status 14), and a ring of ONE key pads to
// length 1, n = 0, where interpolate([], []) evaluates -x[0] % m with x[0] undefined (src/proofGK/interpolate.ts:40).
function statusError(st) {
    if (st === 14) return new TypeError("Cannot read properties of undefined (reading 'k')")
    return new Error(STATUS_TEXT[st] || ('status ' + st))
}
function checkRingSize(nKeys, proving) {
    if (nKeys === 1 && proving) throw new TypeError('Cannot mix BigInt and other types, use explicit conversions')
    if (nKeys < 2) throw new RangeError('the key ring needs at least two keys (the reference cannot prove over fewer: interpolate.ts:40)')
}

// ---------------------------------------------------------------- big numbers and the two groups (host side, BigInt)
const mod = (a, m) => { const r = a % m; return r < 0n ? r + m : r }
function invMod(a, m) { // extended Euclid; invMod(0) = 0 like src/bignum/big.ts:80-119
    let [r0, r1, s0, s1] = [mod(a, m), m, 1n, 0n]
    while (r1 !== 0n) { const q = r0 / r1; [r0, r1] = [r1, r0 - q * r1]; [s0, s1] = [s1, s0 - q * s1] }
    return r0 === 1n ? mod(s0, m) : 0n
}
const hex = (v) => (v < 0n ? '-0x' + (-v).toString(16) : '0x' + v.toString(16))   // src/bignum/big.ts:230-239
const unhex = (s) => { if (typeof s !== 'string' || !s) throw new Error('the field is required'); return s[0] === '-' ? -BigInt(s.slice(1)) : BigInt(s) }
function toBE(v, len) { return Buffer.from(v.toString(16).padStart(2 * len, '0'), 'hex') }
const fromBE = (buf) => (buf.length ? BigInt('0x' + Buffer.from(buf).toString('hex')) : 0n)
function rnd(n) { // src/bignum/big.ts:171-181: byteLen(n) random bytes, retry while >= n
    const len = Math.ceil(n.toString(2).length / 8)
    for (;;) { const v = fromBE(crypto.randomBytes(len)); if (v < n) return v }
}

class Group { // src/curves/group.ts:20-61 (name, p, order); the arithmetic here is affine and only serves the one-time calls
    constructor(name, p, order, gen) { this.name = name; this.p = p; this.order = order; this.gen = gen }
    eq(g) { return this.name === g.name }
    generator() { return new Point(this, this.gen[0], this.gen[1]) }
    sizeFieldBytes() { return Math.ceil(this.p.toString(2).length / 8) }
    newScalar(k) { return new Scalar(this, k) }
    randomScalar() { return new Scalar(this, rnd(this.order)) }
    toJSON() { return { name: this.name } }
}
class WeierstrassGroup extends Group { // y^2 = x^3 + ax + b, src/curves/weier.ts:25-89
    constructor(name, p, a, b, order, gen) { super(name, p, order, gen); this.a = a; this.b = b }
    isOnGroup(pt) { const { p } = this; return pt.inf || mod(pt.y * pt.y - (pt.x * pt.x * pt.x + this.a * pt.x + this.b), p) === 0n }
    add(P, Q) {
        const { p } = this
        if (P.inf) return Q
        if (Q.inf) return P
        let l
        if (P.x === Q.x) {
            if (mod(P.y + Q.y, p) === 0n) return new Point(this, 0n, 1n, true)
            l = mod((3n * P.x * P.x + this.a) * invMod(2n * P.y, p), p)
        } else l = mod((Q.y - P.y) * invMod(Q.x - P.x, p), p)
        const x = mod(l * l - P.x - Q.x, p)
        return new Point(this, x, mod(l * (P.x - x) - P.y, p))
    }
    identity() { return new Point(this, 0n, 1n, true) }
}
class TEdwards extends Group { // a x^2 + y^2 = 1 + d x^2 y^2, src/curves/edwards.ts:25-86; the addition law is complete
    constructor(name, p, a, d, order, gen) { super(name, p, order, gen); this.a = a; this.d = d }
    isOnGroup(pt) { const { p } = this, x2 = pt.x * pt.x, y2 = pt.y * pt.y; return mod(this.a * x2 + y2 - 1n - this.d * x2 % p * y2, p) === 0n }
    add(P, Q) {
        const { p } = this, t = mod(this.d * P.x * Q.x % p * P.y * Q.y, p)
        return new Point(this, mod((P.x * Q.y + P.y * Q.x) * invMod(1n + t, p), p), mod((P.y * Q.y - this.a * P.x * Q.x) * invMod(1n - t, p), p))
    }
    identity() { return new Point(this, 0n, 1n) }
}
class Point { // affine; src/curves/weier.ts:92-101 / edwards.ts:89-98 serialise group, x, y
    constructor(group, x, y, inf = false) { this.group = group; this.x = x; this.y = y; this.inf = inf }
    eq(o) { return this.group.eq(o.group) && this.inf === o.inf && (this.inf || (this.x === o.x && this.y === o.y)) }
    add(o) { return this.group.add(this, o) }
    mul(k) { // double-and-add, MSB first (a one-time call: generateParamsList)
        const s = typeof k === 'bigint' ? mod(k, this.group.order) : k.k
        let r = this.group.identity()
        for (const bit of s.toString(2)) { r = r.add(r); if (bit === '1') r = r.add(this) }
        return r
    }
    isIdentity() { return this.group instanceof TEdwards ? this.x === 0n && this.y === 1n : this.inf }
    toAffine() { return this.group instanceof TEdwards || !this.inf ? { x: this.x, y: this.y } : false }
    toJSON() { return { group: this.group.toJSON(), x: hex(this.x), y: hex(this.y) } }
}
class Scalar { // src/curves/group.ts:155-218
    constructor(group, k) { this.group = group; this.k = k ? mod(BigInt(k), group.order) : 0n }
    eq(o) { return this.group.eq(o.group) && this.k === o.k }
    toJSON() { return { group: this.group.toJSON(), k: hex(this.k) } }
}
const P = (s) => BigInt(s)
const p256 = new WeierstrassGroup('p256', P('0xffffffff00000001000000000000000000000000ffffffffffffffffffffffff'),
    P('0xffffffff00000001000000000000000000000000fffffffffffffffffffffffc'), P('0x5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b'),
    P('0xffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551'),
    [P('0x6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296'), P('0x4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5')])
const tomEdwards256 = new TEdwards('tomEdwards256', P('0x3fffffffc000000040000000000000002ae382c7957cc4ff9713c3d82bc47d3af'),
    P('0x1abce3fd8e1d7a21252515332a512e09d4249bd5b1ec35e316c02254fe8cedf5d'), P('0x051781d9823abde00ec99295ba542c8b1401874bcbeb9e9c861174c7bca6a02aa'),
    P('0xffffffff00000001000000000000000000000000ffffffffffffffffffffffff'),
    [P('0x7907055d0a7d4abc3eafdc25d431d9659fbe007ee2d8ddc4e906206ea9ba4fdb'), P('0xbe231cb9f9bf18319c9f081141559b0a33dddccd2221f0464a9cd57081b01a01')])
const ALL_GROUPS = [p256, tomEdwards256]   // the reference also lists war256, which the ZKAttest path never touches
function groupByName(name) { // src/curves/instances.ts:58-78
    for (const g of ALL_GROUPS) if (g.name === name) return g
    throw new Error('invalid group name: ' + name)
}
function pointFromJson(o, want) { // onDeserialized: 'afterJson' re-validates every point (weier.ts:256-260, edwards.ts:204-209)
    if (!o || typeof o !== 'object' || !o.group) throw new Error('error deserializing a point')
    const g = groupByName(o.group.name)
    if (want && !g.eq(want)) throw new Error('point not in group')
    const pt = new Point(g, unhex(o.x), unhex(o.y))
    if (!g.isOnGroup(pt)) throw new Error('point not in group')
    return pt
}

// ---------------------------------------------------------------- parameter and proof classes
class PedersenParams { // src/commit/pedersen.ts:38-58
    constructor(c, g, h) { this.c = c; this.g = g; this.h = h }
    eq(o) { return this.c.eq(o.c) && this.g.eq(o.g) && this.h.eq(o.h) }
    toJSON() { return { c: this.c.toJSON(), g: this.g.toJSON(), h: this.h.toJSON() } }
    static fromJson(o) {
        if (!o || !o.c) throw new Error('error deserializing PedersenParams')
        const c = groupByName(o.c.name)
        return new PedersenParams(c, pointFromJson(o.g, c), pointFromJson(o.h, c))
    }
}
function generatePedersenParams(c, g) { // src/commit/pedersen.ts:61-69 (h = g * rnd; the TODO about hashing to the curve is the reference's)
    g = g || c.generator()
    return new PedersenParams(c, g, g.mul(c.randomScalar()))
}
class SystemParametersList { // src/zkpAttestList.ts:62-78
    constructor(NistGroup, ProofGroup, SecLevel) { this.NistGroup = NistGroup; this.ProofGroup = ProofGroup; this.SecLevel = SecLevel }
    eq(o) { return this.NistGroup.eq(o.NistGroup) && this.ProofGroup.eq(o.ProofGroup) && this.SecLevel == o.SecLevel }
    toJSON() { return { NistGroup: this.NistGroup.toJSON(), ProofGroup: this.ProofGroup.toJSON(), SecLevel: this.SecLevel } }
    static fromJson(o) {
        if (!o || typeof o.SecLevel !== 'number') throw new Error('error deserializing SystemParametersList')
        return new SystemParametersList(PedersenParams.fromJson(o.NistGroup), PedersenParams.fromJson(o.ProofGroup), o.SecLevel)
    }
    // the engine's form of the parameters: affine big-endian coordinates (include/zkattest.h)
    engineParams() {
        if (!this.NistGroup.c.eq(p256) || !this.ProofGroup.c.eq(tomEdwards256)) throw new Error('params: NistGroup must be p256 and ProofGroup tomEdwards256')
        if (!this.NistGroup.g.eq(p256.generator())) throw new Error('params: NistGroup.g must be the P-256 generator')
        const h = this.NistGroup.h, g2 = this.ProofGroup.g, h2 = this.ProofGroup.h
        return { nistH: Buffer.concat([toBE(h.x, 32), toBE(h.y, 32)]), tomG: Buffer.concat([toBE(g2.x, 36), toBE(g2.y, 36)]),
            tomH: Buffer.concat([toBE(h2.x, 36), toBE(h2.y, 36)]), secLevel: this.SecLevel }
    }
}
function generateParamsList(secLevel = 80) { // src/zkpAttestList.ts:88-92
    return new SystemParametersList(generatePedersenParams(p256), generatePedersenParams(tomEdwards256), secLevel)
}
// Hardened mode (include/zkattest.h; NOT byte-compatible with the reference, both sides must opt in): h of both groups derived
// from SHA-256 (nobody knows log_g h: the TODO of src/commit/pedersen.ts:62) and the statement hashed into the membership
// challenge (the TODO of src/proofGK/gk.ts:178).  `params.hardened = true` selects the mode for proveSignatureList /
// verifySignatureList; the flag is not part of the JSON form (set it again after readJson).
function generateParamsListHardened(secLevel = 80, tag = Buffer.alloc(0)) {
    const h = native.hardenedH(Buffer.from(tag))
    const pt = (g, buf, w) => new Point(g, fromBE(buf.slice(0, w)), fromBE(buf.slice(w)))
    const params = new SystemParametersList(new PedersenParams(p256, p256.generator(), pt(p256, h.nistH, 32)),
        new PedersenParams(tomEdwards256, tomEdwards256.generator(), pt(tomEdwards256, h.tomH, 36)), secLevel)
    params.hardened = true
    return params
}

// Proof objects keep the engine's ZKA1 bytes (the binary equivalent of the reference's object graph, include/zkattest.h) and
// materialise the reference's members (R, comS1, keyXcom, keyYcom, expProof[], membershipProof: Points and Scalars) on demand.
// ZKA1 framing (include/zkattest.h): magic, big-endian total length at byte 4 equal to the buffer's length, 4-byte granular.  Checked
// before a proof may enter a packed batch: a proof with a wrong length would shift every later proof of the blob.
function wellFormedProof(b) { return Buffer.isBuffer(b) && b.length >= 32 && b.length % 4 === 0 && (b.slice(0, 4).toString('latin1') === 'ZKA1' || b.slice(0, 4).toString('latin1') === 'ZK1P') && b.readUInt32BE(4) === b.length }
function revive(v) {
    if (Array.isArray(v)) return v.map(revive)
    if (v && typeof v === 'object') {
        if (v.group && 'x' in v && 'y' in v) return new Point(groupByName(v.group.name), unhex(v.x), unhex(v.y))
        if (v.group && 'k' in v) return new Scalar(groupByName(v.group.name), unhex(v.k))
        const o = {}
        for (const k of Object.keys(v)) if (k !== '__type') o[k] = revive(v[k])
        return o
    }
    return v
}
class SignatureProofList { // src/zkpAttestList.ts:27-60
    constructor(bytes) { if (!wellFormedProof(bytes)) throw new Error('error deserializing'); this.bytes = bytes }
    // the encoding is canonical: equal members <=> equal bytes of ONE layout; a ZKA1 and a ZK1P image of the same proof are equal through their (layout-free) JSON text
    eq(o) {
        if (!(o instanceof SignatureProofList)) return false
        if (this.bytes.slice(0, 4).equals(o.bytes.slice(0, 4))) return this.bytes.equals(o.bytes)
        return this.toJson() === o.toJson()
    }
    toJson() { return native.proofToJson(this.bytes) }
    get members() { if (!this._m) Object.defineProperty(this, '_m', { value: revive(JSON.parse(this.toJson())) }); return this._m }
    get R() { return this.members.R }
    get comS1() { return this.members.comS1 }
    get keyXcom() { return this.members.keyXcom }
    get keyYcom() { return this.members.keyYcom }
    get expProof() { return this.members.expProof }
    get membershipProof() { return this.members.membershipProof }
}
// Whole batches on every host core, off the event loop (zk_proofs_to_json_batch / zk_proofs_from_json_batch): at ~600 KB of text per
// proof the per-proof writeJson / readJson below cannot keep up with a GPU that makes 300 000 proofs a second.
function blobOf(items) {
    const off = new BigUint64Array(items.length + 1)
    let o = 0n
    items.forEach((b, i) => { off[i] = o; o += BigInt(b.length) })
    off[items.length] = o
    return { blob: items.length === 1 ? items[0] : Buffer.concat(items), offsets: Buffer.from(off.buffer) }
}
function cut(r) {
    const off = u64(r.offsets), st = i32(r.status), out = []
    for (let i = 0; i < st.length; i++) out.push(st[i] === 0 ? r.blob.slice(Number(off[i]), Number(off[i + 1])) : null)
    return out
}
// proofs: SignatureProofList[] | Buffer[]  ->  Promise<string[]>   (null where a proof is malformed)
async function writeJsonBatch(proofs, threads = 0) {
    const { blob, offsets } = blobOf(proofs.map((p) => (p instanceof SignatureProofList ? p.bytes : p)))
    return cut(await native.proofsToJsonBatch(blob, offsets, threads)).map((b) => (b === null ? null : b.toString('latin1')))
}
// texts: (string | Buffer)[]  ->  Promise<(SignatureProofList | null)[]>   (null where readJson would throw)
async function readJsonBatch(texts, threads = 0) {
    const { blob, offsets } = blobOf(texts.map((t) => (Buffer.isBuffer(t) ? t : Buffer.from(t, 'utf8'))))
    return cut(await native.proofsFromJsonBatch(blob, offsets, threads)).map((b) => (b === null ? null : new SignatureProofList(b)))
}
function writeJson(type, object) { // src/serde.ts:34-36
    if (type === SignatureProofList) return (object instanceof SignatureProofList ? object : new SignatureProofList(object)).toJson()
    if (type === SystemParametersList || type === PedersenParams) return JSON.stringify(object.toJSON())
    throw new Error('writeJson: unsupported class')
}
function readJson(type, text) { // src/serde.ts:21-32 (throws on bad input)
    if (type === SignatureProofList) return new SignatureProofList(native.proofFromJson(text))
    if (type === SystemParametersList) return SystemParametersList.fromJson(JSON.parse(text))
    if (type === PedersenParams) return PedersenParams.fromJson(JSON.parse(text))
    throw new Error('readJson: unsupported class')
}

// ---------------------------------------------------------------- low-level engine (one handle = the listed GPUs)
function be32(v) { return Buffer.isBuffer(v) ? v : toBE(mod(BigInt(v), 1n << 256n), 32) }
const i32 = (buf) => new Int32Array(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.length))
const u64 = (buf) => new BigUint64Array(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.length))
function defaultDevices() { return (process.env.ZKATTEST_DEVICES || '0').split(',').map((s) => parseInt(s, 10)) }
function unpackProofs(r, B) {
    const st = i32(r.status), off = u64(r.offsets), len = u64(r.lengths), out = []
    for (let b = 0; b < B; b++) {
        if (st[b] !== 0) throw statusError(st[b])
        out.push(r.proofs.slice(Number(off[b]), Number(off[b] + len[b])))   // views of one (page-locked) buffer, no copies
    }
    return out
}
// Only well-formed proofs are packed (back to back, so every one starts 4-byte aligned); a malformed one never reaches the engine
// and cannot misalign its neighbours -- it is reported on its own as 'error deserializing', like the reference's readJson would.
function packProofs(proofs) {
    const good = [], slot = new Int32Array(proofs.length).fill(-1)
    for (let b = 0; b < proofs.length; b++) if (wellFormedProof(proofs[b])) { slot[b] = good.length; good.push(proofs[b]) }
    const B = good.length, off = new BigUint64Array(B), len = new BigUint64Array(B)
    let o = 0n
    for (let b = 0; b < B; b++) { off[b] = o; len[b] = BigInt(good[b].length); o += len[b] }
    return { blob: B === 1 ? good[0] : Buffer.concat(good), off: Buffer.from(off.buffer), len: Buffer.from(len.buffer), slot, n: B }
}
// -> booleans, one per submitted proof.  A proof for which the reference's verifier would THROW ('params not found', a point that
// does not deserialise, ...) is `false` here and its Error sits in the result's non-enumerable `errors[b]` (null otherwise): one bad
// proof must not deny the verdicts of the others.  verifySignatureList (one proof) rethrows, as the reference does.
function verdicts(r, slot) {
    const st = r ? i32(r.status) : null, out = [], errors = []
    for (let b = 0; b < slot.length; b++) {
        const k = slot[b]
        if (k < 0) { out.push(false); errors.push(new Error('error deserializing')); continue }
        errors.push(st[k] !== 0 ? statusError(st[k]) : null)
        out.push(st[k] === 0 && r.ok[k] === 1)
    }
    Object.defineProperty(out, 'errors', { value: errors })
    return out
}
function msgOf(msg, pk) {   // the message hashes of the packed proofs only
    if (pk.n === pk.slot.length) return msg
    const parts = []
    for (let b = 0; b < pk.slot.length; b++) if (pk.slot[b] >= 0) parts.push(msg.slice(32 * b, 32 * b + 32))
    return Buffer.concat(parts)
}
function seedsOf(seeds, pk) {
    if (!seeds || pk.n === pk.slot.length) return seeds || null
    const parts = []
    for (let b = 0; b < pk.slot.length; b++) if (pk.slot[b] >= 0) parts.push(seeds.slice(32 * b, 32 * b + 32))
    return Buffer.concat(parts)
}
class Engine {
    constructor(devices = 0) {
        const ids = Array.isArray(devices) ? devices : [devices]
        this.h = native.createPool(Int32Array.from(ids))
        this.tail = Promise.resolve()
    }
    close() { if (this.h) native.destroyPool(this.h); this.h = null }
    info() { return native.poolInfo(this.h) }
    setOption(name, value) { native.setOption(this.h, name, value) }   // chunk, lanes, combBits (before setParams), hostTaper, batchVerify, mode, slice, ringFold, verifyGroups, wire (0 ZKA1, 1 ZKA1P), inflight
    wipe() { native.setOption(this.h, 'wipe', 0) }   // zk_ctx_wipe on every device: prover workspaces and staged inputs zeroed (also done by close() and after a failed prove)
    // params: { nistH: 64 B, tomG: 72 B, tomH: 72 B, secLevel } -- SystemParametersList as affine big-endian coordinates
    setParams(p) { native.setParams(this.h, p.nistH, p.tomG, p.tomH, p.secLevel || 80); this.params = p }
    setRing(keys) { return native.setRing(this.h, Buffer.isBuffer(keys) ? keys : Buffer.concat(keys.map(be32))) }
    synthParams(seed) { return Object.assign(native.synthParams(this.h, seed), { secLevel: 80 }) }
    synthWorkload(seed, nKeys, B) { return native.synthWorkload(this.h, seed, nKeys, B) }
    keysToInts(pkxy) { return native.keysToInts(this.h, pkxy) }                       // keyToInt over a key set
    _proveArgs(msg, which, seeds) {
        const B = msg.length / 32
        return [B, Buffer.isBuffer(which) ? which : Buffer.from(Uint32Array.from(which).buffer), seeds || crypto.randomBytes(32 * B)]   // one fresh seed per proof
    }
    // -> array of proof Buffers; throws the reference's error text for the first failed proof
    proveBatch(msg, sig, pk, which, seeds) {
        const [B, w, s] = this._proveArgs(msg, which, seeds)
        return unpackProofs(native.proveBatch(this.h, msg, sig, pk, w, s), B)
    }
    // -> array of booleans; exceptions of the reference's verifier are thrown for the first proof that has one
    verifyBatch(msg, proofs, seeds) {
        const pk = packProofs(proofs)
        return verdicts(pk.n ? native.verifyBatch(this.h, msgOf(msg, pk), pk.blob, pk.off, pk.len, seedsOf(seeds, pk)) : null, pk.slot)
    }
    // Promise-returning variants: the batch runs on a libuv worker thread; jobs of one engine are chained (one batch at a time)
    _chain(run) { this.tail = this.tail.then(run, run); return this.tail }
    _proveNow(msg, sig, pk, which, seeds) {
        const [B, w, s] = this._proveArgs(msg, which, seeds)
        return native.proveBatchAsync(this.h, msg, sig, pk, w, s).then((r) => unpackProofs(r, B))
    }
    _verifyNow(msg, proofs, seeds) {
        const pk = packProofs(proofs)
        if (!pk.n) return Promise.resolve(verdicts(null, pk.slot))
        return native.verifyBatchAsync(this.h, msgOf(msg, pk), pk.blob, pk.off, pk.len, seedsOf(seeds, pk)).then((r) => verdicts(r, pk.slot))
    }
    proveBatchAsync(msg, sig, pk, which, seeds) { return this._chain(() => this._proveNow(msg, sig, pk, which, seeds)) }
    verifyBatchAsync(msg, proofs, seeds) { return this._chain(() => this._verifyNow(msg, proofs, seeds)) }
    run(f) { return this._chain(f) }   // any other call on the handle, queued behind the running batches
    // Streamed form (zk_pool_prove_submit / _wait, DESIGN.md 5c): every call returns its Promise at once and up to `inflight` batches are
    // inside the engine together (setOption('inflight', n), default 3), so the pipeline does not drain between batches.  `out` /
    // `blob` are page-locked Buffers (Engine.hostAlloc) owned by the job until its Promise settles -- one per batch in flight.
    // No other call on this engine until every streamed Promise has settled.
    static hostAlloc(bytes) { return native.hostAlloc(bytes) }
    proveStream(msg, sig, pk, which, seeds, out) {   // -> { proofs: Buffer[] (views into out), status: Int32Array }
        const [B, w, s] = this._proveArgs(msg, which, seeds)
        return native.proveSubmit(this.h, msg, sig, pk, w, s, out).then((r) => {
            const off = u64(r.offsets), len = u64(r.lengths)
            return { proofs: Array.from({ length: B }, (_, b) => r.proofs.slice(Number(off[b]), Number(off[b] + len[b]))), status: i32(r.status), used: r.used }
        })
    }
    verifyStream(msg, blob, offsets, lengths, seeds) {   // packed proofs in a page-locked Buffer -> { ok: boolean[], status: Int32Array }
        return native.verifySubmit(this.h, msg, blob, offsets, lengths, seeds || null).then((r) => ({ ok: Array.from(r.ok, (v) => v === 1), status: i32(r.status) }))
    }
}

// ---------------------------------------------------------------- context cache: (params) -> engine, (engine, ring) -> loaded
const engines = new Map()      // sha256(params bytes | secLevel | devices) -> { engine, ringTag }
const ringCache = new WeakMap() // keys array -> { buf, tag }: valid only while the array is FROZEN (nobody can change it in place)
// The loaded ring is identified by the SHA-256 of the whole key list (2 MiB at 2^16 keys: a few milliseconds).  A mutable bigint[]
// is serialised and hashed on EVERY call -- a cached copy validated by a sample would silently prove against a stale ring after an
// in-place change of an unsampled element; callers that want the serialisation cached pass a Buffer or Object.freeze(keys).
function ringOf(keys) {
    if (Buffer.isBuffer(keys)) return { buf: keys, tag: crypto.createHash('sha256').update(keys).digest('hex') }
    const frozen = Object.isFrozen(keys)
    let c = frozen ? ringCache.get(keys) : undefined
    if (!c) {
        const buf = Buffer.concat(keys.map(be32))
        c = { buf, tag: crypto.createHash('sha256').update(buf).digest('hex') }
        if (frozen) ringCache.set(keys, c)
    }
    return c
}
function engineFor(params, keys) {
    if (!(params instanceof SystemParametersList)) throw new TypeError('params: a SystemParametersList (generateParamsList / readJson)')
    if (!params._ep) {
        const ep = params.engineParams(), devices = defaultDevices()
        const tag = crypto.createHash('sha256').update(ep.nistH).update(ep.tomG).update(ep.tomH).update(String(ep.secLevel) + '|' + devices.join(',')).digest('hex')
        Object.defineProperty(params, '_tag0', { value: tag })
        Object.defineProperty(params, '_ep', { value: ep })
    }
    const key = params._tag0 + (params.hardened ? '|hardened' : '')
    let slot = engines.get(key)
    if (!slot) {
        const engine = new Engine(defaultDevices())
        if (process.env.ZKATTEST_COMB_BITS) engine.setOption('combBits', parseInt(process.env.ZKATTEST_COMB_BITS, 10))
        if (params.hardened) engine.setOption('mode', 1)
        engine.setParams(params._ep)           // builds the fixed-base tables: once per SystemParametersList
        slot = { engine, ringTag: null }
        engines.set(key, slot)
    }
    const ring = ringOf(keys)
    // one queued unit per call: (re)load the ring if the loaded one differs (table E: once per key ring), then run the batch;
    // units of one engine run strictly one after the other, so interleaved callers with different rings cannot mix them up
    const withRing = (job) => slot.engine.run(() => {
        if (slot.ringTag !== ring.tag) {
            slot.engine.setRing(ring.buf)
            slot.ringTag = ring.tag
        }
        return job(slot.engine)
    })
    return { engine: slot.engine, withRing }
}
function shutdown() { for (const s of engines.values()) s.engine.close(); engines.clear() }
// The wire layout of the proofs the reference-shaped calls below PRODUCE: 'zka1' (default; 36-byte Tom coordinates) or 'zka1p' (33-byte, 5.3 % fewer bytes
// to move: include/zkattest.h "packed wire layout"; ZKATTEST_WIRE sets the default).  Verification takes either layout, per proof, whatever this is set
// to; the JSON text of a proof does not depend on it.
let wireLayout = /^zka1p$/i.test(process.env.ZKATTEST_WIRE || '') ? 1 : 0
function setWireLayout(name) {
    if (!/^zka1p?$/i.test(name)) throw new TypeError("wire layout: 'zka1' or 'zka1p'")
    wireLayout = /p$/i.test(name) ? 1 : 0
}
function getWireLayout() { return wireLayout ? 'zka1p' : 'zka1' }
function useWire(slotEngine, wire) {   // inside a queued unit of the engine (nothing of it is in flight)
    if (slotEngine._wire !== wire) { slotEngine.setOption('wire', wire); slotEngine._wire = wire }
}
const isPacked = (b) => Buffer.isBuffer(b) && b.length >= 4 && b.slice(0, 4).toString('latin1') === 'ZK1P'

// ---------------------------------------------------------------- the reference's API
// publicKey: a WebCrypto CryptoKey (where crypto.subtle exists), a Node KeyObject, or the 65-byte 'raw' export 04 || X || Y
async function rawPublicKey(publicKey) {
    if (Buffer.isBuffer(publicKey) || publicKey instanceof Uint8Array) return Buffer.from(publicKey)
    if (publicKey && publicKey.type === 'public' && typeof publicKey.export === 'function') return publicKey.export({ type: 'spki', format: 'der' }).slice(-65)
    const subtle = (crypto.webcrypto && crypto.webcrypto.subtle) || (global.crypto && global.crypto.subtle)
    if (subtle && publicKey && publicKey.algorithm) return Buffer.from(await subtle.exportKey('raw', publicKey))
    throw new Error('invalid public key')
}
function checkedKeyPoint(raw) { // p256.deserializePoint + toAffine (zkpAttestList.ts:95-101, 113-118; weier.ts:74-89)
    if (raw.length !== 65 || raw[0] !== 4) throw new Error('error deserializing uncompressed point')
    const pt = new Point(p256, fromBE(raw.slice(1, 33)), fromBE(raw.slice(33)))
    if (!p256.isOnGroup(pt)) throw new Error('point not in group')
    return new Point(p256, mod(pt.x, p256.p), mod(pt.y, p256.p))
}
async function keyToInt(publicKey) { return checkedKeyPoint(await rawPublicKey(publicKey)).x }

async function proveSignatureList(params, msgHash, sigBytes, publicKey, which, keys) {
    return (await proveSignatureListBatch(params, [msgHash], [sigBytes], [publicKey], [which], keys))[0]
}
async function verifySignatureList(params, msgHash, keys, proof) {
    const r = await verifySignatureListBatch(params, [msgHash], keys, [proof])
    if (r.errors[0]) throw r.errors[0]   // the reference's verifier throws these (src/exp/exp.ts:244,270,302; deserialisation)
    return r[0]
}
// B statements over one ring in one call (what the engine is built for): arrays of the single-proof arguments
async function proveSignatureListBatch(params, msgHashes, sigs, publicKeys, whichs, keys) {
    const raws = await Promise.all(publicKeys.map(rawPublicKey))
    for (const r of raws) if (r.length !== 65 || r[0] !== 4) throw new Error('invalid public key')
    const msg = Buffer.concat(msgHashes.map((m) => Buffer.from(m))), sig = Buffer.concat(sigs.map((s) => Buffer.from(s)))
    checkRingSize(Buffer.isBuffer(keys) ? keys.length / 32 : keys.length, true)
    const wire = wireLayout
    const proofs = await engineFor(params, keys).withRing((engine) => { useWire(engine, wire); return engine._proveNow(msg, sig, Buffer.concat(raws.map((r) => r.slice(1))), whichs) })
    return proofs.map((b) => new SignatureProofList(b))
}
async function verifySignatureListBatch(params, msgHashes, keys, proofs) {
    const msg = Buffer.concat(msgHashes.map((m) => Buffer.from(m))), raw = proofs.map((p) => (p instanceof SignatureProofList ? p.bytes : p))
    checkRingSize(Buffer.isBuffer(keys) ? keys.length / 32 : keys.length, false)
    // a context verifies ONE layout at a time (zk_ctx_set_wire): the batch is split by the proofs' magic and the verdicts are put back in order
    const idx = [[], []]
    raw.forEach((b, i) => idx[isPacked(b) ? 1 : 0].push(i))
    const out = new Array(raw.length), errors = new Array(raw.length)
    for (const wire of [0, 1]) {
        const sel = idx[wire]
        if (!sel.length) continue
        const m = sel.length === raw.length ? msg : Buffer.concat(sel.map((i) => msg.slice(32 * i, 32 * i + 32)))
        const r = await engineFor(params, keys).withRing((engine) => { useWire(engine, wire); return engine._verifyNow(m, sel.map((i) => raw[i])) })
        sel.forEach((i, k) => { out[i] = r[k]; errors[i] = r.errors[k] })
    }
    Object.defineProperty(out, 'errors', { value: errors })
    return out
}

module.exports = { generateParamsList, generateParamsListHardened, keyToInt, proveSignatureList, verifySignatureList, proveSignatureListBatch, verifySignatureListBatch,
    writeJson, readJson, writeJsonBatch, readJsonBatch, SignatureProofList, SystemParametersList, PedersenParams, generatePedersenParams, p256, tomEdwards256, ALL_GROUPS,
    Group, Point, Scalar, Engine, shutdown, setWireLayout, getWireLayout, native }
