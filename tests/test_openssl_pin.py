"""not-gpu: P-256 arithmetic and the ECDSA front end of the restatements against an implementation that is not this build's --
the OpenSSL inside this image's Node (crypto.createECDH('prime256v1') for k*G and k*P, crypto.sign / crypto.verify for ECDSA).
Every parity claim of the engine runs through oracle/zkattest_ref.py (and the C / JS restatements that are byte-identical to it);
this pins its curve layer (src/curves/weier.ts:133-260, group.ts:97-152) and the signature algebra of
src/zkpAttestList.ts:119-135 to an independent code base.  Proof-level parity stays unpinned."""
import hashlib
import json
import os
import random
import shutil
import subprocess

import pytest

import zkattest_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(shutil.which('node') is None, reason='no node in this environment')


def _openssl(req):
    p = subprocess.run(['node', os.path.join(HERE, 'openssl_p256.js')], input=json.dumps(req).encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return json.loads(p.stdout.decode())


def _raw(P):
    x, y = P.toAffine()
    return '04' + format(x, '064x') + format(y, '064x')


def test_fixed_and_variable_base_multiplication_match_openssl():
    n = R.p256.order
    rnd = random.Random(256)
    ks = [1, 2, 3, n - 1, n - 2, (1 << 255) % n, (1 << 128) - 1] + [rnd.randrange(1, n) for _ in range(24)]
    G = R.p256.generator()
    bases = [G.mul(R.p256.newScalar(rnd.randrange(1, n))) for _ in range(6)]
    pairs = [(rnd.randrange(1, n), P) for P in bases for _ in range(3)] + [(n - 1, bases[0]), (1, bases[1]), (2, bases[2])]
    got = _openssl({'mulG': [format(k, 'x') for k in ks], 'ecdh': [{'k': format(k, 'x'), 'pk': _raw(P)} for k, P in pairs]})
    assert got['mulG'] == [_raw(G.mul(R.p256.newScalar(k))) for k in ks]                      # window-4 mul (group.ts:133-152)
    assert got['ecdh'] == [format(P.mul(R.p256.newScalar(k)).toAffine()[0], '064x') for k, P in pairs]
    # the Straus double multiplication the verifier's front end uses (group.ts:97-132): u1*G + u2*P = (u1 + u2*k)*G for P = k*G
    for _ in range(6):
        k, u1, u2 = (rnd.randrange(1, n) for _ in range(3))
        P = G.mul(R.p256.newScalar(k))
        assert _raw(G.dblmul(R.p256.newScalar(u1), P, R.p256.newScalar(u2))) == _openssl({'mulG': [format((u1 + u2 * k) % n, 'x')]})['mulG'][0]


def test_ecdsa_front_end_matches_openssl_both_ways():
    """(1) signatures made by the restatement (what the synthetic workloads consist of) verify under OpenSSL; (2) signatures made by
    OpenSSL verify under the restatement's ecdsa_verify -- the same u1 = z/s, u2 = r/s algebra the prover's front end starts
    from (zkpAttestList.ts:119-135); (3) a flipped bit is rejected by both."""
    n = R.p256.order
    rnd = random.Random(6979)
    ver, sign, keys = [], [], []
    for i in range(8):
        d = rnd.randrange(1, n)
        msg = b'message %d' % i
        h = hashlib.sha256(msg).digest()
        pk = R.ecdsa_pubkey(d)
        sig = R.ecdsa_sign(d, h)                     # RFC 6979 nonce
        assert R.ecdsa_verify(pk, h, sig)
        bad = bytes([sig[0] ^ 1]) + sig[1:]
        ver += [{'pk': pk.hex(), 'msg': msg.hex(), 'sig': sig.hex()}, {'pk': pk.hex(), 'msg': msg.hex(), 'sig': bad.hex()}]
        sign.append({'d': format(d, 'x'), 'msg': msg.hex()})
        keys.append((pk, h))
    got = _openssl({'verify': ver, 'sign': sign})
    assert got['verify'] == [True, False] * 8
    for (pk, h), rs in zip(keys, got['sign']):
        sig = bytes.fromhex(rs)
        assert R.ecdsa_verify(pk, h, sig)
        assert not R.ecdsa_verify(pk, h, sig[:40] + bytes([sig[40] ^ 4]) + sig[41:])


def test_synthetic_workload_keys_are_real_p256_keys():
    """The public keys planted into the synthetic rings (tools, bench.py, every parity test) are d*G for the workload's d: OpenSSL
    derives the same points."""
    ds, pks = [], []
    for b in range(6):
        msgHash, sig, pk, which, d, seed = R.synth_proof_input(2024, b, 8)
        ds.append(format(d, 'x'))
        pks.append(pk.hex())
        assert R.ecdsa_verify(pk, msgHash, sig)
    assert _openssl({'mulG': ds})['mulG'] == pks
