// node test_addon.js json <golden.json>      CPU-only: JSON wire format through the addon against the committed digest
// node test_addon.js gpu                       on an MI355X: synthetic workload -> prove -> writeJson/readJson -> verify
'use strict'
const assert = require('assert')
const crypto = require('crypto')
const fs = require('fs')
const zk = require('./zkattest.js')
const i32 = (buf) => new Int32Array(buf.buffer, buf.byteOffset, buf.length / 4)

async function main() {
    const mode = process.argv[2]
    if (mode === 'json') {
        const gold = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'))
        const rec = gold.small_full.proofs[0]
        const proof = Buffer.from(rec.proof, 'hex')
        const text = zk.writeJson(proof)
        assert.strictEqual(text.length, rec.json_len)
        assert.strictEqual(crypto.createHash('sha256').update(text).digest('hex'), rec.json_sha256)
        assert.ok(zk.readJson(text).equals(proof))
        const obj = JSON.parse(text)
        assert.deepStrictEqual(Object.keys(obj), ['R', 'comS1', 'keyXcom', 'keyYcom', 'expProof', 'membershipProof'])
        assert.throws(() => zk.readJson(text.slice(0, -1)), /error deserializing/)
        console.log('json ok', text.length)
        return
    }
    const eng = new zk.Engine(0)
    const params = eng.synthParams(7)
    eng.setParams(params)
    const B = 6, nKeys = 16
    const wl = eng.synthWorkload(7, nKeys, B)
    eng.setRing(wl.ring)
    const proofs = eng.proveBatch(wl.msg, wl.sig, wl.pk, wl.which, wl.seeds)
    assert.strictEqual(proofs.length, B)
    const again = eng.proveBatch(wl.msg, wl.sig, wl.pk, wl.which, wl.seeds)             // deterministic under the RNG contract
    assert.ok(again.every((p, i) => p.equals(proofs[i])))
    const viaJson = proofs.map((p) => zk.readJson(zk.writeJson(p)))                      // test/zkpAttestList.test.ts:55-60
    assert.deepStrictEqual(eng.verifyBatch(wl.msg, viaJson), Array(B).fill(true))
    const forged = Buffer.from(proofs[2])
    forged[forged.length - 1] ^= 1
    const mixed = proofs.slice()
    mixed[2] = forged
    assert.deepStrictEqual(eng.verifyBatch(wl.msg, mixed), [true, true, false, true, true, true])
    const one = await zk.proveSignatureList(eng, wl.msg.slice(0, 32), wl.sig.slice(0, 64), Buffer.concat([Buffer.from([4]), wl.pk.slice(0, 64)]), 0)
    assert.strictEqual(await zk.verifySignatureList(eng, wl.msg.slice(0, 32), one), true)
    const bad = Buffer.from(wl.pk.slice(0, 64))
    bad[63] ^= 1
    assert.throws(() => eng.proveBatch(wl.msg.slice(0, 32), wl.sig.slice(0, 64), bad, [0], wl.seeds.slice(0, 32)), /point not in group/)
    // Promise-based calls: two batches queued back to back on one engine, results in order
    const [pa, pb] = await Promise.all([eng.proveBatchAsync(wl.msg, wl.sig, wl.pk, wl.which, wl.seeds), eng.proveBatchAsync(wl.msg.slice(0, 64), wl.sig.slice(0, 128), wl.pk.slice(0, 128), [0, 1], wl.seeds.slice(0, 64))])
    assert.ok(pa.every((p, i) => p.equals(proofs[i])) && pb.length === 2 && pb[1].equals(proofs[1]))
    assert.deepStrictEqual(await eng.verifyBatchAsync(wl.msg, mixed), [true, true, false, true, true, true])
    await assert.rejects(eng.proveBatchAsync(wl.msg.slice(0, 32), wl.sig.slice(0, 64), bad, [0], wl.seeds.slice(0, 32)), /point not in group/)
    // the reference's own test flow (test/zkpAttestList.test.ts:25-63) with REAL keys and signatures made by Node's OpenSSL:
    // ECDSA P-256 / SHA-256 key pairs, ring = keyToInt of every public key, prove for one of them, JSON round trip, verify
    {
        const n = 5, mine = 3, message = Buffer.from('ZKAttest: the signer is one of the ring, nobody learns which one')
        const pairs = Array.from({ length: n }, () => crypto.generateKeyPairSync('ec', { namedCurve: 'P-256' }))
        const raw = pairs.map((kp) => kp.publicKey.export({ type: 'spki', format: 'der' }).slice(-65))   // 04 || X || Y
        const ring = eng.keysToInts(Buffer.concat(raw.map((r) => r.slice(1))))
        assert.ok(i32(ring.status).every((v) => v === 0))
        eng.setRing(ring.keys)
        const sig = crypto.sign('sha256', message, { key: pairs[mine].privateKey, dsaEncoding: 'ieee-p1363' })   // r || s
        const msgHash = crypto.createHash('sha256').update(message).digest()
        const proof = await zk.proveSignatureList(eng, msgHash, sig, raw[mine], mine)
        assert.strictEqual(await zk.verifySignatureList(eng, msgHash, zk.readJson(zk.writeJson(proof))), true)
        const otherHash = crypto.createHash('sha256').update('another message').digest()
        assert.strictEqual(await zk.verifySignatureList(eng, otherHash, proof), false)
        const stranger = crypto.generateKeyPairSync('ec', { namedCurve: 'P-256' })                       // not in the ring
        const sig2 = crypto.sign('sha256', message, { key: stranger.privateKey, dsaEncoding: 'ieee-p1363' })
        const p2 = await zk.proveSignatureList(eng, msgHash, sig2, stranger.publicKey.export({ type: 'spki', format: 'der' }).slice(-65), mine)
        assert.strictEqual(await zk.verifySignatureList(eng, msgHash, p2), false)
        eng.setRing(wl.ring)
    }
    const k = eng.keysToInts(wl.pk)
    assert.ok(k.keys.slice(0, 32).equals(wl.pk.slice(0, 32)))
    eng.close()
    console.log('gpu ok', proofs[0].length)
}
main().catch((e) => { console.error(e); process.exit(1) })
