#!/usr/bin/env python3
"""Generates tests/golden/*.json with the Python restatement (oracle/zkattest_ref.py).  The reference TypeScript
cannot run in the build container, so these vectors pin the C restatement and the HIP engine to the Python
restatement under the RNG contract; public vectors (RFC 6979 A.2.5, FIPS 180-4) and the reference's own KATs
pin the primitives.  Re-run: python tests/golden/make_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'oracle'))
import zkattest_ref as R  # noqa: E402


def xy(pt, w):
    x, y = pt.toAffine()
    return (x.to_bytes(w, 'big') + y.to_bytes(w, 'big')).hex()


def case(S, nkeys, B, sec, full=False, stream_plant=None):
    params = R.synth_params(S, sec)
    ring = R.synth_ring_fast(S, nkeys)
    ins = [R.synth_proof_input(S, b, nkeys) for b in range(B)]
    for m, s, p, w, d, seed in ins:
        ring[w] = R.keyToInt(p)
    out = {'S': S, 'nkeys': nkeys, 'sec': sec, 'nist_h': xy(params.NistGroup.h, 32), 'tom_g': xy(params.ProofGroup.g, 36),
           'tom_h': xy(params.ProofGroup.h, 36), 'ring': [format(v, 'x') for v in ring], 'proofs': []}
    for b, (m, s, p, w, d, seed) in enumerate(ins):
        if stream_plant is None:
            rng = R.SeedRng(seed)
            rec = {'seed': seed.hex()}
        else:
            nblk = 3 + 44 * sec + 5 * 8 + 16
            blocks = [hashlib.sha256(seed + k.to_bytes(8, 'big')).digest() for k in range(nblk)]
            for idx, val in stream_plant:
                blocks[idx] = val.to_bytes(32, 'big')
            rng = R.StreamRng(blocks)
            rec = {'stream_seed': seed.hex(), 'stream_blocks': nblk, 'plant': [[i, format(v, 'x')] for i, v in stream_plant]}
        proof = R.proveSignatureList(params, m, s, p, w, ring, rng)
        assert R.verifySignatureList(params, m, ring, proof)
        raw = R.proof_to_bytes(proof)
        rec.update({'msg': m.hex(), 'sig': s.hex(), 'pk': p[1:].hex(), 'which': w, 'len': len(raw), 'sha256': hashlib.sha256(raw).hexdigest(),
                    'fills_consumed': rng.k})
        if full:
            rec['proof'] = raw.hex()
            text = R.proof_to_json(proof)   # writeJson(SignatureProofList, proof), src/serde.ts:34-36
            rec['json_len'], rec['json_sha256'] = len(text), hashlib.sha256(text.encode()).hexdigest()
        out['proofs'].append(rec)
    return out


def main():
    n, q = R.p256.order, R.p256.p
    golden = {
        'kats': {
            'invMod': [[3, 5, R.invMod(3, 5)], [7, 41, R.invMod(7, 41)]],          # test/bignum/big.test.ts:19-21
            'interpolate': [[1, 2, 3], [1, 2, 3], 401, R.interpolate([1, 2, 3], [1, 2, 3], 401)],  # test/proofGK/interpolate.test.ts:19-26
        },
        'small_full': case(3, 5, 1, 20, full=True),       # secLevel 20 (the least the verifier accepts, zkpAttestList.ts:177): a complete ZKA1 proof
        'ring6_sec80': case(1, 6, 2, 80),                 # the reference test's shape: 6 keys padded to 8 (test/zkpAttestList.test.ts:37)
        'ring37_sec80': case(1037, 37, 1, 80),
        # rejection path of rnd() (big.ts:171-181): fills >= the modulus are planted at the draws for comS1.r (mod n),
        # pkX.r (mod q), alpha_0 (mod n: value in [n, q) is rejected), Tx_0.r (mod q: same value is accepted)
        'rejection_stream': case(9, 4, 1, 20, full=False, stream_plant=[(0, n + 12345), (2, q + 1), (3, (1 << 256) - 1), (7, n + 99), (11, n + 5)]),
    }
    with open(os.path.join(HERE, 'golden.json'), 'w') as f:
        json.dump(golden, f, indent=0, separators=(',', ':'))
    print('wrote golden.json', os.path.getsize(os.path.join(HERE, 'golden.json')), 'bytes')


if __name__ == '__main__':
    main()
