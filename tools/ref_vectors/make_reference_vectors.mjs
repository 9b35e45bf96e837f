// Emits tests/golden/reference_vectors.json from the REAL TypeScript reference, for the inputs of tests/golden/golden.json.
// Needs what this build's container lacks: Node >= 24 and the reference built with its own toolchain --
//
//   git clone https://github.com/cloudflare/zkp-ecdsa && cd zkp-ecdsa && git checkout v0.2.6 && npm ci && npm run build
//   node <this repo>/tools/ref_vectors/make_reference_vectors.mjs "$PWD" <this repo>/tests/golden/golden.json \
//        <this repo>/tests/golden/reference_vectors.json
//
// For every golden case it rebuilds SystemParametersList from the recorded points, imports the recorded public key with
// WebCrypto, seeds the deterministic getRandomValues (detcrypto.mjs: the engine's RNG contract) with the recorded per-proof
// seed (and planted fills), and records writeJson(SignatureProofList, await proveSignatureList(...)) plus the reference
// verifier's verdict.  tests/test_reference_vectors.py then requires: engine / oracle ZKA1 bytes == readJson(text) of the
// reference, and (strict mode) this build's JSON writer == the reference's typedjson text.  Nothing here runs in the build
// container; the committed golden.json is produced by the Python restatement (tests/golden/make_golden.py).
import { readFileSync, writeFileSync } from 'fs'
import { pathToFileURL } from 'url'
import { join } from 'path'
import { install } from './detcrypto.mjs'

async function main() {
    const [refDir, goldenPath, outPath] = process.argv.slice(2)
    if (!refDir || !goldenPath || !outPath) throw new Error('usage: make_reference_vectors.mjs <reference checkout> <golden.json> <out.json>')
    const det = install(globalThis)
    const ref = await import(pathToFileURL(join(refDir, 'lib', 'src', 'index.js')).href)
    const { proveSignatureList, verifySignatureList, readJson, writeJson, SignatureProofList, SystemParametersList } = ref
    const golden = JSON.parse(readFileSync(goldenPath, 'utf8'))
    const hx = (s) => '0x' + (s.replace(/^0+/, '') || '0')
    const point = (group, bytesHex, w) => ({ group: { name: group }, x: hx(bytesHex.slice(0, 2 * w)), y: hx(bytesHex.slice(2 * w)) })
    const G = { x: '0x6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296', y: '0x4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5' }
    const out = { generator: 'tools/ref_vectors/make_reference_vectors.mjs', node: process.version, cases: {} }
    for (const [name, c] of Object.entries(golden)) {
        if (!c.proofs) continue
        const paramsJson = JSON.stringify({
            NistGroup: { c: { name: 'p256' }, g: { group: { name: 'p256' }, ...G }, h: point('p256', c.nist_h, 32) },
            ProofGroup: { c: { name: 'tomEdwards256' }, g: point('tomEdwards256', c.tom_g, 36), h: point('tomEdwards256', c.tom_h, 36) },
            SecLevel: c.sec,
        })
        const params = readJson(SystemParametersList, paramsJson)
        const keys = c.ring.map((v) => BigInt('0x' + v))
        const recs = []
        for (const p of c.proofs) {
            const raw = Buffer.concat([Buffer.from([4]), Buffer.from(p.pk, 'hex')])
            const publicKey = await det.subtle.importKey('raw', raw, { name: 'ECDSA', namedCurve: 'P-256' }, true, ['verify'])
            const planted = (p.plant || []).map(([i, v]) => [i, Buffer.from(v.padStart(64, '0'), 'hex')])
            det.reseed(Buffer.from(p.seed || p.stream_seed, 'hex'), planted)
            const proof = await proveSignatureList(params, Buffer.from(p.msg, 'hex'), Buffer.from(p.sig, 'hex'), publicKey, p.which, keys)
            const fills = det.k
            const json = writeJson(SignatureProofList, proof)
            det.seed = null                                           // the verifier draws platform randomness, like in production
            const verdict = await verifySignatureList(params, Buffer.from(p.msg, 'hex'), keys, readJson(SignatureProofList, json))
            recs.push({ msg: p.msg, sig: p.sig, pk: p.pk, which: p.which, seed: p.seed, stream_seed: p.stream_seed, plant: p.plant, fills_consumed: fills,
                json, reference_verifies: verdict })
        }
        out.cases[name] = { nist_h: c.nist_h, tom_g: c.tom_g, tom_h: c.tom_h, sec: c.sec, nkeys: c.nkeys, ring: c.ring, params_json: writeJson(SystemParametersList, params), proofs: recs }
    }
    writeFileSync(outPath, JSON.stringify(out))
    console.log('wrote', outPath)
}
main().catch((e) => { console.error(e); process.exit(1) })
