// node test_addon.js cpu <golden.json>   CPU-only: params / keys / serde of the façade, JSON wire format against the committed digest
// node test_addon.js gpu                  on an MI355X: the reference's own tests restated against the façade + the batch calls
'use strict'
const assert = require('assert')
const crypto = require('crypto')
const fs = require('fs')
const zk = require('./zkattest.js')
const { generateParamsList, keyToInt, proveSignatureList, verifySignatureList, writeJson, readJson, SignatureProofList, SystemParametersList } = zk
const i32 = (buf) => new Int32Array(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.length))

// test/serde.test.ts:21-32
function serdeTest(type, object) {
    const json = writeJson(type, object), object2 = readJson(type, json)
    return object.eq(object2)
}
// what crypto.subtle.generateKey / sign / digest give the reference's test, from Node's OpenSSL (Node 12 has no WebCrypto)
function keyAndSignature(text) {
    const keyPair = crypto.generateKeyPairSync('ec', { namedCurve: 'P-256' }), msg = Buffer.from(text)
    return { keyPair, msgHash: crypto.createHash('sha256').update(msg).digest(),
        signature: crypto.sign('sha256', msg, { key: keyPair.privateKey, dsaEncoding: 'ieee-p1363' }) }   // r || s, like WebCrypto ECDSA
}

async function cpu(goldenPath) {
    // generateParamsList (zkpAttestList.ts:88-92): h = g * rnd on both groups, default secLevel 80; serde round trip with eq()
    const params = generateParamsList()
    assert.strictEqual(params.SecLevel, 80)
    assert.ok(params.NistGroup.c.eq(zk.p256) && params.ProofGroup.c.eq(zk.tomEdwards256))
    assert.ok(params.NistGroup.g.eq(zk.p256.generator()) && zk.p256.isOnGroup(params.NistGroup.h) && zk.tomEdwards256.isOnGroup(params.ProofGroup.h))
    assert.ok(!generateParamsList(40).eq(params) && generateParamsList(40).SecLevel === 40)
    assert.ok(serdeTest(SystemParametersList, params))
    const obj = JSON.parse(writeJson(SystemParametersList, params))
    assert.deepStrictEqual(Object.keys(obj), ['NistGroup', 'ProofGroup', 'SecLevel'])
    assert.deepStrictEqual(Object.keys(obj.NistGroup), ['c', 'g', 'h'])
    assert.strictEqual(obj.ProofGroup.c.name, 'tomEdwards256')
    assert.ok(/^0x[0-9a-f]+$/.test(obj.ProofGroup.h.x))
    const tampered = JSON.parse(JSON.stringify(obj))
    tampered.ProofGroup.h.x = '0x5'
    assert.throws(() => readJson(SystemParametersList, JSON.stringify(tampered)), /point not in group/)
    tampered.ProofGroup.h.group.name = 'war256'
    assert.throws(() => readJson(SystemParametersList, JSON.stringify(tampered)), /invalid group name/)
    const ep = params.engineParams()
    assert.ok(ep.nistH.length === 64 && ep.tomG.length === 72 && ep.tomH.length === 72 && ep.secLevel === 80)
    // group sanity (test/curves/ec.test.ts:21-40): order * G = identity, P + (order - 1) P = identity
    for (const g of zk.ALL_GROUPS) {
        const G = g.generator()
        assert.ok(G.mul(g.order).isIdentity() && G.add(G.mul(g.order - 1n)).isIdentity() && g.isOnGroup(G.mul(12345678901234567890n)))
    }
    // keyToInt (zkpAttestList.ts:94-102): x coordinate of a KeyObject / raw export; off-curve keys are refused
    const { keyPair } = keyAndSignature('x')
    const raw = keyPair.publicKey.export({ type: 'spki', format: 'der' }).slice(-65)
    assert.strictEqual(await keyToInt(keyPair.publicKey), BigInt('0x' + raw.slice(1, 33).toString('hex')))
    assert.strictEqual(await keyToInt(raw), await keyToInt(keyPair.publicKey))
    const off = Buffer.from(raw)
    off[64] ^= 1
    await assert.rejects(keyToInt(off), /point not in group/)
    // JSON wire format of a proof against the committed digest; object model with eq()
    const gold = JSON.parse(fs.readFileSync(goldenPath, 'utf8'))
    const rec = gold.small_full.proofs[0]
    const proof = new SignatureProofList(Buffer.from(rec.proof, 'hex'))
    const text = writeJson(SignatureProofList, proof)
    assert.strictEqual(text.length, rec.json_len)
    assert.strictEqual(crypto.createHash('sha256').update(text).digest('hex'), rec.json_sha256)
    assert.ok(serdeTest(SignatureProofList, proof))
    assert.deepStrictEqual(Object.keys(JSON.parse(text)), ['R', 'comS1', 'keyXcom', 'keyYcom', 'expProof', 'membershipProof'])
    assert.ok(proof.R.group.eq(zk.p256) && zk.p256.isOnGroup(proof.R) && proof.keyXcom.group.eq(zk.tomEdwards256) && zk.tomEdwards256.isOnGroup(proof.keyXcom))
    assert.strictEqual(proof.expProof.length, gold.small_full.sec)
    assert.ok(proof.expProof.every((e) => ('alpha' in e) !== ('proof' in e)) && proof.membershipProof.zd.group.eq(zk.tomEdwards256))
    const other = Buffer.from(proof.bytes)
    other[other.length - 1] ^= 1
    assert.ok(!proof.eq(new SignatureProofList(other)))
    assert.throws(() => readJson(SignatureProofList, text.slice(0, -1)), /error deserializing/)
    // the batch converters: the same texts, the same proofs, a malformed item does not disturb its neighbours
    const texts = await zk.writeJsonBatch([proof, proof.bytes.slice(0, -4), proof], 2)
    assert.deepStrictEqual(texts, [text, null, text])
    const back = await zk.readJsonBatch([text, text.slice(0, -1), Buffer.from(text)])
    assert.ok(back[0].eq(proof) && back[1] === null && back[2].eq(proof))
    // hardened parameters: derived, not drawn -- the same every time, on the curves, different per tag
    const hp = zk.generateParamsListHardened(), hp2 = zk.generateParamsListHardened(80)
    assert.ok(hp.eq(hp2) && hp.hardened && !hp.eq(zk.generateParamsListHardened(80, Buffer.from('x'))))
    assert.ok(zk.p256.isOnGroup(hp.NistGroup.h) && zk.tomEdwards256.isOnGroup(hp.ProofGroup.h) && hp.ProofGroup.h.mul(zk.tomEdwards256.order).isIdentity())
    assert.ok(serdeTest(SystemParametersList, hp))
    console.log('cpu ok', text.length)
}

async function gpu() {
    // ---- test/zkpAttestList.test.ts:28-54, argument for argument
    {
        const { keyPair, msgHash, signature } = keyAndSignature('kilroy was here'),
            testKey = await keyToInt(keyPair.publicKey),
            testArray = [testKey, BigInt(4), BigInt(5), BigInt(6), BigInt(7), BigInt(8)],
            params = generateParamsList(),
            proof = await proveSignatureList(params, msgHash, signature, keyPair.publicKey, 0, testArray),
            res = await verifySignatureList(params, msgHash, testArray, proof)
        assert.strictEqual(res, true)
        assert.ok(serdeTest(SignatureProofList, proof))
        assert.ok(serdeTest(SystemParametersList, params))
        // example/usage.ts:43-57: JSON out, verify
        const proofJSON = writeJson(SignatureProofList, proof)
        assert.ok(proofJSON.length > 100000)
        assert.strictEqual(await verifySignatureList(params, msgHash, testArray, readJson(SignatureProofList, proofJSON)), true)
        // negative cases the reference has no test for: another message, another ring, a signer outside the ring
        assert.strictEqual(await verifySignatureList(params, crypto.createHash('sha256').update('x').digest(), testArray, proof), false)
        const otherRing = testArray.slice()
        otherRing[0] = BigInt(9)
        assert.strictEqual(await verifySignatureList(params, msgHash, otherRing, proof), false)
        assert.strictEqual(await verifySignatureList(params, msgHash, testArray, proof), true)   // and back: the ring cache follows the argument
        // an IN-PLACE change of the caller's array (same identity, same length, an element no sample would look at) is seen too
        const big = testArray.concat(Array.from({ length: 58 }, (_, i) => BigInt(100 + i)))
        const pbig = await proveSignatureList(params, msgHash, signature, keyPair.publicKey, 0, big)
        assert.strictEqual(await verifySignatureList(params, msgHash, big, pbig), true)
        const was = big[37]
        big[37] = BigInt(3)
        assert.strictEqual(await verifySignatureList(params, msgHash, big, pbig), false)
        big[37] = was
        assert.strictEqual(await verifySignatureList(params, msgHash, Object.freeze(big), pbig), true)
        const stranger = keyAndSignature('kilroy was here')
        const p2 = await proveSignatureList(params, msgHash, stranger.signature, stranger.keyPair.publicKey, 0, testArray)
        assert.strictEqual(await verifySignatureList(params, msgHash, testArray, p2), false)
        // params are cached by identity: a second SystemParametersList with the same content reuses the engine
        const again = readJson(SystemParametersList, writeJson(SystemParametersList, params))
        const t0 = Date.now()
        assert.strictEqual(await verifySignatureList(again, msgHash, testArray, proof), true)
        assert.ok(Date.now() - t0 < 2000)
        // interleaved callers with different rings on one engine
        const [a, b] = await Promise.all([verifySignatureList(params, msgHash, testArray, proof), verifySignatureList(params, msgHash, otherRing, proof)])
        assert.deepStrictEqual([a, b], [true, false])
        const badKey = keyPair.publicKey.export({ type: 'spki', format: 'der' }).slice(-65)
        badKey[64] ^= 1
        await assert.rejects(proveSignatureList(params, msgHash, signature, badKey, 0, testArray), /point not in group/)
        // where the reference dies with a TypeError of the runtime: `which` past the padded ring (gk.ts:162), a one-key ring (interpolate.ts:40)
        await assert.rejects(proveSignatureList(params, msgHash, signature, keyPair.publicKey, testArray.length + 2, testArray), TypeError)
        await assert.rejects(proveSignatureList(params, msgHash, signature, keyPair.publicKey, 0, [testArray[0]]), /Cannot mix BigInt/)
        const pad = await proveSignatureList(params, msgHash, signature, keyPair.publicKey, testArray.length, testArray)   // 6 keys pad to 8: index 6 is keys[0]
        assert.strictEqual(await verifySignatureList(params, msgHash, testArray, pad), true)
        // the batch form over the same ring
        const B = 5, ks = Array.from({ length: B }, (_, i) => keyAndSignature('message ' + i))
        const ring = await Promise.all(ks.map((k) => keyToInt(k.keyPair.publicKey)))
        const proofs = await zk.proveSignatureListBatch(params, ks.map((k) => k.msgHash), ks.map((k) => k.signature), ks.map((k) => k.keyPair.publicKey), [0, 1, 2, 3, 4], ring)
        const swapped = [proofs[1], proofs[0], proofs[2], proofs[3], proofs[4]]
        assert.deepStrictEqual(await zk.verifySignatureListBatch(params, ks.map((k) => k.msgHash), ring, proofs), Array(B).fill(true))
        assert.deepStrictEqual(await zk.verifySignatureListBatch(params, ks.map((k) => k.msgHash), ring, swapped), [false, false, true, true, true])
        // the packed wire layout through the reference-shaped calls: setWireLayout('zka1p') makes proveSignatureList return ZK1P proofs (5.3 % shorter), the
        // verifier takes either layout per proof -- mixed in one batch too --, the object model and the JSON text do not see the difference
        zk.setWireLayout('zka1p')
        assert.strictEqual(zk.getWireLayout(), 'zka1p')
        const packedProof = await proveSignatureList(params, msgHash, signature, keyPair.publicKey, 0, testArray)
        assert.strictEqual(packedProof.bytes.slice(0, 4).toString('latin1'), 'ZK1P')
        const sameAsZka1 = readJson(SignatureProofList, writeJson(SignatureProofList, packedProof))      // the reader emits ZKA1: the same proof in the other layout
        assert.strictEqual(sameAsZka1.bytes.slice(0, 4).toString('latin1'), 'ZKA1')
        assert.ok(packedProof.bytes.length < sameAsZka1.bytes.length && packedProof.bytes.length > 0.9 * sameAsZka1.bytes.length)
        assert.ok(packedProof.eq(sameAsZka1) && sameAsZka1.eq(packedProof) && !packedProof.eq(proof))                // eq() is layout-free, not proof-free
        assert.strictEqual(await verifySignatureList(params, msgHash, testArray, sameAsZka1), true)
        assert.strictEqual(await verifySignatureList(params, msgHash, testArray, packedProof), true)
        assert.strictEqual(await verifySignatureList(params, msgHash, otherRing, packedProof), false)
        assert.strictEqual(packedProof.expProof.length, proof.expProof.length)
        assert.ok(serdeTest(SignatureProofList, packedProof))
        const packedBatch = await zk.proveSignatureListBatch(params, ks.map((k) => k.msgHash), ks.map((k) => k.signature), ks.map((k) => k.keyPair.publicKey), [0, 1, 2, 3, 4], ring)
        assert.ok(packedBatch.every((p) => p.bytes.slice(0, 4).toString('latin1') === 'ZK1P'))
        const mixedLayouts = [proofs[0], packedBatch[1], packedBatch[2], proofs[3], packedBatch[4]]
        assert.deepStrictEqual(await zk.verifySignatureListBatch(params, ks.map((k) => k.msgHash), ring, mixedLayouts), Array(B).fill(true))
        const mixedSwapped = [packedBatch[1], proofs[0], packedBatch[2], proofs[3], packedBatch[4]]
        assert.deepStrictEqual(await zk.verifySignatureListBatch(params, ks.map((k) => k.msgHash), ring, mixedSwapped), [false, false, true, true, true])
        zk.setWireLayout('zka1')
        assert.throws(() => zk.setWireLayout('json'), TypeError)
        assert.strictEqual((await proveSignatureList(params, msgHash, signature, keyPair.publicKey, 0, testArray)).bytes.slice(0, 4).toString('latin1'), 'ZKA1')
        // hardened mode end to end: proofs of one mode do not verify in the other
        const hparams = zk.generateParamsListHardened()
        const hproof = await proveSignatureList(hparams, msgHash, signature, keyPair.publicKey, 0, testArray)
        assert.strictEqual(await verifySignatureList(hparams, msgHash, testArray, hproof), true)
        const soft = readJson(SystemParametersList, writeJson(SystemParametersList, hparams))      // same generators, reference mode
        assert.strictEqual(await verifySignatureList(soft, msgHash, testArray, hproof), false)
        assert.strictEqual(await verifySignatureList(hparams, msgHash, otherRing, hproof), false)
        zk.shutdown()
    }
    // ---- the low-level engine: synthetic workload, determinism under the RNG contract, misuse of handles
    const eng = new zk.Engine(0)
    const params = eng.synthParams(7)
    eng.setParams(params)
    const B = 6, nKeys = 16
    const wl = eng.synthWorkload(7, nKeys, B)
    assert.strictEqual(eng.setRing(wl.ring), 'single')
    const proofs = eng.proveBatch(wl.msg, wl.sig, wl.pk, wl.which, wl.seeds)
    assert.strictEqual(proofs.length, B)
    const again = eng.proveBatch(wl.msg, wl.sig, wl.pk, wl.which, wl.seeds)             // deterministic under the RNG contract
    assert.ok(again.every((p, i) => p.equals(proofs[i])))
    const viaJson = proofs.map((p) => readJson(SignatureProofList, writeJson(SignatureProofList, p)).bytes)
    assert.deepStrictEqual(eng.verifyBatch(wl.msg, viaJson), Array(B).fill(true))
    // the packed wire form (ZKA1P, include/zkattest.h): same proofs, 33-byte Tom coordinates; the JSON text is the same text
    eng.setOption('wire', 1)
    const packed = eng.proveBatch(wl.msg, wl.sig, wl.pk, wl.which, wl.seeds)
    assert.ok(packed.every((p, i) => p.slice(0, 4).toString('latin1') === 'ZK1P' && p.length < proofs[i].length))
    assert.ok(packed.every((p, i) => writeJson(SignatureProofList, p) === writeJson(SignatureProofList, proofs[i])))
    assert.deepStrictEqual(eng.verifyBatch(wl.msg, packed), Array(B).fill(true))
    eng.setOption('wire', 0)                                                            // a context verifies the layout it is set to
    assert.deepStrictEqual(eng.verifyBatch(wl.msg, packed), Array(B).fill(false))
    const forged = Buffer.from(proofs[2])
    forged[forged.length - 1] ^= 1
    const mixed = proofs.slice()
    mixed[2] = forged
    assert.deepStrictEqual(eng.verifyBatch(wl.msg, mixed), [true, true, false, true, true, true])
    // one malformed proof (truncated: its length no longer matches its header) must not deny the verdicts of the honest ones
    const broken = proofs.slice()
    broken[1] = proofs[1].slice(0, proofs[1].length - 5)
    broken[4] = Buffer.from('not a proof')
    const bv = eng.verifyBatch(wl.msg, broken)
    assert.deepStrictEqual(bv, [true, false, true, true, false, true])
    assert.ok(/deserializ/.test(bv.errors[1].message) && /deserializ/.test(bv.errors[4].message) && bv.errors[0] === null)
    assert.throws(() => new SignatureProofList(broken[1]), /deserializ/)
    const otherSec = Buffer.from(proofs[3])
    otherSec.writeUInt32BE(7, 8)                                                       // header claims another secLevel than its structure holds
    const sv = eng.verifyBatch(Buffer.concat([wl.msg.slice(0, 32), wl.msg.slice(96, 128)]), [proofs[0], otherSec])
    assert.deepStrictEqual(sv, [true, false])
    assert.ok(/deserializ/.test(sv.errors[1].message))                                 // exact codes: include/zkattest.h, tests/test_gpu_mutants.py
    const bad = Buffer.from(wl.pk.slice(0, 64))
    bad[63] ^= 1
    assert.throws(() => eng.proveBatch(wl.msg.slice(0, 32), wl.sig.slice(0, 64), bad, [0], wl.seeds.slice(0, 32)), /point not in group/)
    const [pa, pb] = await Promise.all([eng.proveBatchAsync(wl.msg, wl.sig, wl.pk, wl.which, wl.seeds), eng.proveBatchAsync(wl.msg.slice(0, 64), wl.sig.slice(0, 128), wl.pk.slice(0, 128), [0, 1], wl.seeds.slice(0, 64))])
    assert.ok(pa.every((p, i) => p.equals(proofs[i])) && pb.length === 2 && pb[1].equals(proofs[1]))
    assert.deepStrictEqual(await eng.verifyBatchAsync(wl.msg, mixed), [true, true, false, true, true, true])
    await assert.rejects(eng.proveBatchAsync(wl.msg.slice(0, 32), wl.sig.slice(0, 64), bad, [0], wl.seeds.slice(0, 32)), /point not in group/)
    // a handle is busy while an asynchronous batch runs: direct calls and destroy are refused instead of racing the worker thread
    const running = zk.native.proveBatchAsync(eng.h, wl.msg, wl.sig, wl.pk, wl.which, wl.seeds)
    assert.throws(() => zk.native.setRing(eng.h, wl.ring), /busy/)
    assert.throws(() => zk.native.destroyPool(eng.h), /while an asynchronous batch/)
    await running
    // streamed form: five batches through three page-locked buffers, three inside the engine at a time; bytes = the synchronous call
    {
        const nb = 5, bufs = [0, 1, 2].map(() => zk.Engine.hostAlloc(4 << 20))
        eng.setOption('inflight', 3)
        const sub = (k) => eng.proveStream(wl.msg, wl.sig, wl.pk, wl.which, wl.seeds, bufs[k % 3])
        const pend = [sub(0), sub(1), sub(2)]
        assert.throws(() => zk.native.setRing(eng.h, wl.ring), /busy/)                   // exclusive calls wait for the stream to drain
        assert.throws(() => zk.native.destroyPool(eng.h), /while an asynchronous batch/)
        for (let k = 0; k < nb; k++) {
            const r = await pend[k]
            assert.ok(r.status.every((v) => v === 0) && r.proofs.every((p, i) => p.equals(proofs[i])), 'streamed batch ' + k)
            if (k + 3 < nb) pend.push(sub(k + 3))                                        // its buffer is free again
        }
        // verify straight out of a page-locked buffer; a forged proof in the middle batch
        const packed = zk.Engine.hostAlloc(4 << 20), off = new BigUint64Array(B), len = new BigUint64Array(B)
        let at = 0n
        proofs.forEach((p, i) => { p.copy(packed, Number(at)); off[i] = at; len[i] = BigInt(p.length); at += BigInt(p.length) })
        const forgedPacked = zk.Engine.hostAlloc(4 << 20)
        packed.copy(forgedPacked)
        forgedPacked[Number(off[2] + len[2]) - 1] ^= 1
        const ob = Buffer.from(off.buffer), lb = Buffer.from(len.buffer)
        const vs = await Promise.all([eng.verifyStream(wl.msg, packed, ob, lb), eng.verifyStream(wl.msg, forgedPacked, ob, lb), eng.verifyStream(wl.msg, packed, ob, lb)])
        assert.deepStrictEqual(vs.map((v) => v.ok), [Array(B).fill(true), [true, true, false, true, true, true], Array(B).fill(true)])
        await assert.rejects(eng.proveStream(wl.msg, wl.sig, wl.pk, wl.which, wl.seeds, Buffer.alloc(64)), /buffer|page-locked|argument/i)   // not page-locked, too small
        assert.strictEqual(eng.setRing(wl.ring), 'single')                               // drained: exclusive calls work again
    }
    // two contexts on this GPU behind one handle: sharded batch, same bytes; the ring went device to device
    const duo = new zk.Engine([0, 0])
    duo.setParams(params)
    assert.strictEqual(duo.setRing(wl.ring), 'peer-copy')
    assert.strictEqual(duo.info().devices, 2)
    const sharded = await duo.proveBatchAsync(wl.msg, wl.sig, wl.pk, wl.which, wl.seeds)
    assert.ok(sharded.every((p, i) => p.equals(proofs[i])))
    assert.deepStrictEqual(await duo.verifyBatchAsync(wl.msg, mixed), [true, true, false, true, true, true])
    duo.close()
    const k = eng.keysToInts(wl.pk)
    assert.ok(i32(k.status).every((v) => v === 0) && k.keys.slice(0, 32).equals(wl.pk.slice(0, 32)))
    const h = eng.h
    eng.wipe()                                                                           // zk_ctx_wipe on every device
    eng.close()
    eng.close()                                                                          // idempotent
    assert.throws(() => zk.native.setRing(h, wl.ring), /destroyed/)
    console.log('gpu ok', proofs[0].length)
}
const mode = process.argv[2]
;(mode === 'cpu' || mode === 'json' ? cpu(process.argv[3]) : gpu()).catch((e) => { console.error(e); process.exit(1) })
