// Node's OpenSSL as an implementation of P-256 that is NOT this build's: driven by tests/test_openssl_pin.py.
// stdin: JSON {mulG: [k hex], ecdh: [{k, pk}], verify: [{pk, msg, sig_rs}], sign: [{d, msg}]}  -> stdout: JSON with the results
const crypto = require('crypto')
const inp = JSON.parse(require('fs').readFileSync(0, 'utf8'))
const hex = (h, n) => Buffer.from(h.padStart(2 * n, '0'), 'hex')
function derInt(b) { let i = 0; while (i < b.length - 1 && b[i] === 0) i++; b = b.slice(i); if (b[0] & 0x80) b = Buffer.concat([Buffer.from([0]), b]); return Buffer.concat([Buffer.from([2, b.length]), b]) }
function derSig(rs) { const r = derInt(rs.slice(0, 32)), s = derInt(rs.slice(32)); return Buffer.concat([Buffer.from([0x30, r.length + s.length]), r, s]) }
function rsOf(der) { // SEQUENCE { INTEGER r, INTEGER s } -> 64 bytes
    let o = 2; if (der[1] & 0x80) o = 2 + (der[1] & 0x7f)
    const rl = der[o + 1], r = der.slice(o + 2, o + 2 + rl), so = o + 2 + rl, sl = der[so + 1], s = der.slice(so + 2, so + 2 + sl)
    const fix = (b) => { while (b.length > 32) b = b.slice(1); return Buffer.concat([Buffer.alloc(32 - b.length), b]) }
    return Buffer.concat([fix(r), fix(s)])
}
const SPKI = Buffer.from('3059301306072a8648ce3d020106082a8648ce3d030107034200', 'hex')   // id-ecPublicKey, prime256v1, BIT STRING
const pubKey = (raw65) => crypto.createPublicKey({ key: Buffer.concat([SPKI, raw65]), format: 'der', type: 'spki' })
function privKey(d32) { // RFC 5915 ECPrivateKey inside PKCS#8 is more than needed: SEC1 with the curve parameter does
    const sec1 = Buffer.concat([Buffer.from('30310201010420', 'hex'), d32, Buffer.from('a00a06082a8648ce3d030107', 'hex')])
    return crypto.createPrivateKey({ key: sec1, format: 'der', type: 'sec1' })
}
const out = { mulG: [], ecdh: [], verify: [], sign: [] }
for (const k of inp.mulG || []) { const e = crypto.createECDH('prime256v1'); e.setPrivateKey(hex(k, 32)); out.mulG.push(e.getPublicKey('hex')) }
for (const q of inp.ecdh || []) { const e = crypto.createECDH('prime256v1'); e.setPrivateKey(hex(q.k, 32)); out.ecdh.push(e.computeSecret(Buffer.from(q.pk, 'hex')).toString('hex')) }
for (const v of inp.verify || []) out.verify.push(crypto.verify('sha256', Buffer.from(v.msg, 'hex'), pubKey(Buffer.from(v.pk, 'hex')), derSig(Buffer.from(v.sig, 'hex'))))
for (const s of inp.sign || []) out.sign.push(rsOf(crypto.sign('sha256', Buffer.from(s.msg, 'hex'), privKey(hex(s.d, 32)))).toString('hex'))
console.log(JSON.stringify(out))
