// JavaScript façade over the N-API addon with the reference's surface (src/index.ts:17-19, src/zkpAttestList.ts):
// proveSignatureList / verifySignatureList / keyToInt / writeJson / readJson, plus the batch calls the engine is built for.
// Proofs travel as ZKA1 byte strings (include/zkattest.h); writeJson / readJson convert to the typedjson wire format.
// Works on the Node 12 of this image (the TypeScript sources of the reference need Node >= 24 and tsc).
'use strict'
const crypto = require('crypto')
const path = require('path')
const native = require(process.env.ZKATTEST_NODE || path.join(__dirname, 'zkattest.node'))

const STATUS_TEXT = { 1: 'point not in group', 2: 'invalid public key', 3: 'T[i] is at infinity', 4: 'T1 is at infinity', 5: 'P/Q/R is at infinity',
    6: "Points don't add up!", 7: 'R is at infinity', 8: 'params not found', 9: 'security level not achieved', 10: 'error deserializing' }

function be32(v) { // bigint | Buffer -> 32 bytes big-endian
    if (Buffer.isBuffer(v)) return v
    let h = BigInt(v).toString(16)
    return Buffer.from(h.padStart(64, '0'), 'hex')
}
function i32(buf) { return new Int32Array(buf.buffer, buf.byteOffset, buf.length / 4) }
function u64(buf) { return new BigUint64Array(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.length)) }

class Engine {
    constructor(device = 0) { this.ctx = native.createContext(device) }
    close() { if (this.ctx) native.destroyContext(this.ctx); this.ctx = null }
    // params: { nistH: 64 B, tomG: 72 B, tomH: 72 B, secLevel } -- SystemParametersList as affine big-endian coordinates
    setParams(p) { native.setParams(this.ctx, p.nistH, p.tomG, p.tomH, p.secLevel || 80); this.params = p }
    setRing(keys) { native.setRing(this.ctx, Buffer.isBuffer(keys) ? keys : Buffer.concat(keys.map(be32))) }
    synthParams(seed) { return Object.assign(native.synthParams(this.ctx, seed), { secLevel: 80 }) }
    synthWorkload(seed, nKeys, B) { return native.synthWorkload(this.ctx, seed, nKeys, B) }
    keysToInts(pkxy) { return native.keysToInts(this.ctx, pkxy) }                       // keyToInt over a key set
    // -> array of proof Buffers; throws the reference's error text for the first failed proof
    proveBatch(msg, sig, pk, which, seeds) {
        const B = msg.length / 32
        seeds = seeds || crypto.randomBytes(32 * B)                                      // RNG contract: one fresh seed per proof
        const w = Buffer.isBuffer(which) ? which : Buffer.from(Uint32Array.from(which).buffer)
        const r = native.proveBatch(this.ctx, msg, sig, pk, w, seeds)
        const st = i32(r.status), off = u64(r.offsets)
        const out = []
        for (let b = 0; b < B; b++) {
            if (st[b] !== 0) throw new Error(STATUS_TEXT[st[b]] || ('status ' + st[b]))
            out.push(r.proofs.slice(Number(off[b]), Number(off[b + 1])))
        }
        return out
    }
    // -> array of booleans; exceptions of the reference's verifier are thrown for the first proof that has one
    verifyBatch(msg, proofs, seeds) {
        const B = proofs.length
        const off = new BigUint64Array(B + 1)
        for (let b = 0; b < B; b++) off[b + 1] = off[b] + BigInt(proofs[b].length)
        const r = native.verifyBatch(this.ctx, msg, Buffer.concat(proofs), Buffer.from(off.buffer), seeds || null)
        const st = i32(r.status)
        for (let b = 0; b < B; b++) if (st[b] !== 0) throw new Error(STATUS_TEXT[st[b]] || ('status ' + st[b]))
        return Array.from(r.ok).map((v) => v === 1)
    }
}

// Promise-returning batch calls (the batch runs on a libuv worker thread); jobs of one engine are chained because a
// context runs one batch at a time
Engine.prototype.proveBatchAsync = function (msg, sig, pk, which, seeds) {
    const B = msg.length / 32
    seeds = seeds || crypto.randomBytes(32 * B)
    const w = Buffer.isBuffer(which) ? which : Buffer.from(Uint32Array.from(which).buffer)
    const run = () => native.proveBatchAsync(this.ctx, msg, sig, pk, w, seeds).then((r) => {
        const st = i32(r.status), off = u64(r.offsets), out = []
        for (let b = 0; b < B; b++) {
            if (st[b] !== 0) throw new Error(STATUS_TEXT[st[b]] || ('status ' + st[b]))
            out.push(r.proofs.slice(Number(off[b]), Number(off[b + 1])))
        }
        return out
    })
    this.tail = (this.tail || Promise.resolve()).then(run, run)
    return this.tail
}
Engine.prototype.verifyBatchAsync = function (msg, proofs, seeds) {
    const B = proofs.length
    const off = new BigUint64Array(B + 1)
    for (let b = 0; b < B; b++) off[b + 1] = off[b] + BigInt(proofs[b].length)
    const run = () => native.verifyBatchAsync(this.ctx, msg, Buffer.concat(proofs), Buffer.from(off.buffer), seeds || null).then((r) => {
        const st = i32(r.status)
        for (let b = 0; b < B; b++) if (st[b] !== 0) throw new Error(STATUS_TEXT[st[b]] || ('status ' + st[b]))
        return Array.from(r.ok).map((v) => v === 1)
    })
    this.tail = (this.tail || Promise.resolve()).then(run, run)
    return this.tail
}

// ---- the reference's single-proof API on top (zkpAttestList.ts:104-184); `engine` carries params and ring
async function proveSignatureList(engine, msgHash, sigBytes, publicKeyRaw, which) {
    const pk = publicKeyRaw.length === 65 ? publicKeyRaw.slice(1) : publicKeyRaw        // WebCrypto 'raw' export: 04 || X || Y
    return (await engine.proveBatchAsync(msgHash, sigBytes, pk, [which]))[0]
}
async function verifySignatureList(engine, msgHash, proof) { return (await engine.verifyBatchAsync(msgHash, [proof]))[0] }
const writeJson = (proof) => native.proofToJson(proof)       // src/serde.ts:34-36
const readJson = (text) => native.proofFromJson(text)        // src/serde.ts:21-32

module.exports = { Engine, proveSignatureList, verifySignatureList, writeJson, readJson, native }
