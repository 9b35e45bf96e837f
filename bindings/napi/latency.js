// node latency.js [reps]: what ONE proveSignatureList / verifySignatureList call costs through the facade -- the reference's only shape
// (src/zkpAttestList.ts:104-145 proves one signature per call) -- on the reference test's ring of 6 keys and on a ring of 1024 keys.
// The first call of a (params, ring) pair builds the fixed-base tables and the ring's tables; the timed calls reuse them.
'use strict'
const crypto = require('crypto')
const zk = require('./zkattest.js')
function keyAndSignature(text) {
    const keyPair = crypto.generateKeyPairSync('ec', { namedCurve: 'P-256' }), msg = Buffer.from(text)
    return { keyPair, msgHash: crypto.createHash('sha256').update(msg).digest(), signature: crypto.sign('sha256', msg, { key: keyPair.privateKey, dsaEncoding: 'ieee-p1363' }) }
}
const median = (a) => a.slice().sort((x, y) => x - y)[a.length >> 1]
async function main() {
    const reps = parseInt(process.argv[2] || '7', 10)
    const params = zk.generateParamsList()
    const out = { unit: 'ms per call (median of ' + reps + ')', rings: {} }
    for (const nKeys of [6, 1024]) {
        const { keyPair, msgHash, signature } = keyAndSignature('kilroy was here')
        const ring = [await zk.keyToInt(keyPair.publicKey)]
        while (ring.length < nKeys) ring.push(BigInt('0x' + crypto.randomBytes(31).toString('hex')))
        Object.freeze(ring)
        let t0 = Date.now()
        let proof = await zk.proveSignatureList(params, msgHash, signature, keyPair.publicKey, 0, ring)
        const first = Date.now() - t0
        const tp = [], tv = [], tj = []
        for (let k = 0; k < reps; k++) {
            t0 = process.hrtime.bigint()
            proof = await zk.proveSignatureList(params, msgHash, signature, keyPair.publicKey, 0, ring)
            const t1 = process.hrtime.bigint()
            const ok = await zk.verifySignatureList(params, msgHash, ring, proof)
            const t2 = process.hrtime.bigint()
            const text = zk.writeJson(zk.SignatureProofList, proof)
            zk.readJson(zk.SignatureProofList, text)
            const t3 = process.hrtime.bigint()
            if (!ok) throw new Error('proof rejected')
            tp.push(Number(t1 - t0) / 1e6), tv.push(Number(t2 - t1) / 1e6), tj.push(Number(t3 - t2) / 1e6)
        }
        out.rings[nKeys] = { first_call_ms: first, prove_ms: +median(tp).toFixed(2), verify_ms: +median(tv).toFixed(2), json_round_trip_ms: +median(tj).toFixed(2) }
    }
    console.log(JSON.stringify(out))
    zk.shutdown()
}
main().catch((e) => { console.error(e); process.exit(1) })
