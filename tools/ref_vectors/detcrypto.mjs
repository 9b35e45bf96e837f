// Deterministic stand-in for crypto.getRandomValues implementing this build's RNG contract (include/zkattest.h):
//   fill k of a proof = SHA-256(seed || be64(k)); a request for 32 bytes consumes one fill (randomScalar -> rnd(order),
//   src/bignum/big.ts:171-181: byteLen(order) = 32 bytes, retry while >= order), a request for n < 32 bytes the first n bytes
//   of one fill (rndRange on small bounds, src/exp/exp.ts:95-109).
// Plugged in where the reference's own test/mockCrypto.js:18-22 installs webcrypto: everything else (subtle.digest, subtle.sign,
// exportKey) stays the platform's.  With it, proveSignatureList of the REAL TypeScript reference consumes exactly the fills the
// engine consumes, so its output is comparable byte for byte (make_reference_vectors.mjs).
//   node detcrypto.mjs --selftest      (plain Node >= 12; checks the fill contract against known digests)
import nodeCrypto from 'crypto'
const { createHash } = nodeCrypto

export class DeterministicCrypto {
    constructor(base) {
        this.base = base                    // the platform's crypto (subtle etc.)
        this.seed = null
        this.k = 0
        this.planted = new Map()            // fill index -> 32 bytes (the rejection-path vectors plant fills >= the modulus)
    }
    get subtle() { return this.base.subtle }
    reseed(seed, planted) {
        this.seed = Buffer.from(seed)
        this.k = 0
        this.planted = new Map(planted || [])
    }
    fill(k) {
        if (this.planted.has(k)) return this.planted.get(k)
        const ctr = Buffer.alloc(8)
        ctr.writeBigUInt64BE(BigInt(k))
        return createHash('sha256').update(this.seed).update(ctr).digest()
    }
    getRandomValues(arr) {
        if (!this.seed) return this.base.getRandomValues(arr)      // outside a seeded region: the platform's randomness
        const out = new Uint8Array(arr.buffer, arr.byteOffset, arr.byteLength)
        if (out.length > 32) throw new Error('detcrypto: the reference never asks for more than 32 random bytes at once')
        out.set(this.fill(this.k++).subarray(0, out.length))
        return arr
    }
}
export function install(globalObj) {   // like test/mockCrypto.js, but deterministic once reseed() has been called
    const det = new DeterministicCrypto(globalObj.crypto || nodeCrypto.webcrypto)
    Object.defineProperty(globalObj, 'crypto', { value: det, configurable: true, writable: true })
    return det
}

if (process.argv[2] === '--selftest') {
    const det = new DeterministicCrypto({ getRandomValues: () => { throw new Error('unseeded') }, subtle: null })
    const seed = Buffer.alloc(32, 7)
    det.reseed(seed, [[2, Buffer.alloc(32, 0xff)]])
    const a = det.getRandomValues(new Uint8Array(32)), b = det.getRandomValues(new Uint8Array(1)), c = det.getRandomValues(new Uint8Array(32))
    const h = (k) => createHash('sha256').update(seed).update(Buffer.from([0, 0, 0, 0, 0, 0, 0, k])).digest()
    const ok = Buffer.from(a).equals(h(0)) && b[0] === h(1)[0] && Buffer.from(c).equals(Buffer.alloc(32, 0xff)) && det.k === 3
    console.log(ok ? 'detcrypto selftest ok ' + Buffer.from(a).toString('hex').slice(0, 16) : 'detcrypto selftest FAILED')
    process.exit(ok ? 0 : 1)
}
