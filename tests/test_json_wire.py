"""JSON wire format of SignatureProofList: the C-ABI converters zk_proof_to_json /
zk_proof_from_json against the oracle's restatement of writeJson/readJson (src/serde.ts:21-36).  Host-only code:
runs without a GPU.  The reference exercises this in test/zkpAttestList.test.ts:55-60 (writeJson -> readJson ->
verify); the typedjson text itself is unpinned (package not vendored), so the pins here are (1) the committed golden
proof's JSON digest, (2) engine == oracle text, (3) round trips, (4) readJson-style tolerance and rejections."""
import hashlib
import json
import os

import pytest

import zkattest_ref as R
import zkp_ecdsa_amd as Z

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'golden.json')))


def _golden_proofs():
    out = []
    for name in ('small_full', 'ring6_sec80'):
        for rec in GOLD[name]['proofs']:
            if 'proof' in rec:
                out.append((name, bytes.fromhex(rec['proof'])))
    return out


def test_engine_json_equals_oracle_json_and_round_trips():
    cases = _golden_proofs()
    assert cases
    for name, raw in cases:
        text = Z.write_json(raw)
        assert text == R.proof_to_json(R.proof_from_bytes(raw)), name
        assert Z.read_json(text) == raw
        assert R.proof_to_bytes(R.proof_from_json(text)) == raw
        # canonical: serialising the parsed object again gives the same text (writeJson o readJson = id)
        assert Z.write_json(Z.read_json(text)) == text


def test_json_shape_follows_the_decorators():
    raw = _golden_proofs()[0][1]
    top = json.loads(Z.write_json(raw))
    assert list(top.keys()) == ['R', 'comS1', 'keyXcom', 'keyYcom', 'expProof', 'membershipProof']   # zkpAttestList.ts:30-35
    assert top['R']['group']['name'] == 'p256' and top['keyXcom']['group']['name'] == 'tomEdwards256'
    assert list(top['R'].keys())[:3] == ['group', 'x', 'y']
    proof = R.proof_from_bytes(raw)
    for e, j in zip(proof.expProof, top['expProof']):
        keys = [k for k in j.keys()]
        if e.alpha is not None:   # response1 (exp.ts:30-34)
            assert keys == ['A', 'Tx', 'Ty', 'alpha', 'beta1', 'beta2', 'beta3']
            assert j['alpha']['group']['name'] == 'p256' and j['beta2']['group']['name'] == 'tomEdwards256'
            assert int(j['alpha']['k'], 16) == e.alpha.k
        else:                     # response0 (exp.ts:35-40)
            assert keys == ['A', 'Tx', 'Ty', 'z', 'z2', 'proof', 'r1', 'r2']
            assert list(j['proof'].keys()) == ['C_8', 'C_10', 'C_11', 'C_13', 'pi_8', 'pi_10', 'pi_11', 'pi_13', 'pi_x', 'pi_y']
            assert list(j['proof']['pi_8'].keys()) == ['C_4', 'A_x', 'A_y', 'A_z', 'A_4_1', 'A_4_2', 't_x', 't_y', 't_z', 't_rx', 't_ry', 't_rz', 't_r4']
            assert list(j['proof']['pi_x'].keys()) == ['A_1', 'A_2', 't_x', 't_r1', 't_r2']
    gk = top['membershipProof']
    assert list(gk.keys()) == ['cl', 'ca', 'cb', 'cd', 'f', 'za', 'zb', 'zd']                              # gk.ts:32-39
    n = len(proof.membershipProof.cl)
    assert all(len(gk[k]) == n for k in ('cl', 'ca', 'cb', 'cd', 'f', 'za', 'zb'))
    # bigint encoding: '0x' + lowercase hex, no leading zeros (big.ts:230-239)
    assert all(v == '0x%x' % int(v, 16) for v in (top['R']['x'], gk['zd']['k'], gk['f'][0]['k']))


def test_golden_json_digest():
    """The JSON text of the committed golden proof is itself pinned (tests/golden/golden.json, key json_sha256)."""
    rec = GOLD['small_full']['proofs'][0]
    text = Z.write_json(bytes.fromhex(rec['proof']))
    assert hashlib.sha256(text.encode()).hexdigest() == rec['json_sha256']
    assert len(text) == rec['json_len']


def test_reader_is_order_tolerant_and_ignores_type_hints():
    raw = _golden_proofs()[0][1]
    top = json.loads(Z.write_json(raw))

    def strip(v):   # drop every "__type" hint and reverse member order
        if isinstance(v, dict):
            return {k: strip(x) for k, x in reversed(list(v.items())) if k != '__type'}
        if isinstance(v, list):
            return [strip(x) for x in v]
        return v
    loose = json.dumps(strip(top), indent=1)
    assert Z.read_json(loose) == raw
    # leading zeros and upper-case digits are valid BigInt() input
    top['R']['x'] = '0x000' + top['R']['x'][2:].upper()
    assert Z.read_json(json.dumps(top)) == raw


def test_reader_rejects_what_readjson_rejects():
    raw = _golden_proofs()[0][1]
    text = Z.write_json(raw)
    top = json.loads(text)

    def bad(mut):
        t = json.loads(text)
        mut(t)
        with pytest.raises(Z.ZkError) as e:
            Z.read_json(json.dumps(t))
        assert e.value.status == 10   # ZK_E_BAD_ENCODING

    bad(lambda t: t.pop('comS1'))                                        # isRequired member missing
    bad(lambda t: t['membershipProof'].pop('zd'))
    bad(lambda t: t['R']['group'].__setitem__('name', 'secp256k1'))      # instances.ts:58-78 'invalid group name'
    bad(lambda t: t['keyXcom']['group'].__setitem__('name', 'p256'))     # wrong group for this member
    bad(lambda t: t['R'].__setitem__('x', ''))                           # serdeBigInt: 'the field x is required'
    bad(lambda t: t['R'].__setitem__('x', '0xzz'))
    bad(lambda t: t['R'].__setitem__('x', '0x1' + '0' * 64))             # wider than the field
    bad(lambda t: t['membershipProof']['ca'].pop())                      # ragged GK arrays
    bad(lambda t: t['expProof'][0].pop('Tx'))
    idx = next(i for i, e in enumerate(top['expProof']) if 'proof' in e)
    bad(lambda t: t['expProof'][idx]['proof'].pop('pi_13'))
    bad(lambda t: t['expProof'][idx].pop('r1'))
    for junk in ('', '[]', '{', text[:-1], text + 'x', '{"R":'):
        with pytest.raises(Z.ZkError):
            Z.read_json(junk)
    with pytest.raises(Z.ZkError):
        Z.write_json(raw[:-1])          # truncated ZKA1
    with pytest.raises(Z.ZkError):
        Z.write_json(b'ZKA2' + raw[4:])


def test_sizing_call_reports_required_length():
    import ctypes as C
    raw = _golden_proofs()[0][1]
    L = Z.lib()
    n = C.c_uint64()
    assert L.zk_proof_to_json(raw, len(raw), None, 0, C.byref(n)) == 12   # ZK_E_BUFFER, n = required size
    assert n.value == len(Z.write_json(raw))
    small = C.create_string_buffer(16)
    assert L.zk_proof_to_json(raw, len(raw), small, 16, C.byref(n)) == 12


def test_reader_follows_json_parse_on_the_corner_cases():
    """Parser differentials on attacker-controlled text (the same text must not decode to another proof here than under
    JSON.parse + typedjson): a duplicated member -- the LAST one counts; escapes are decoded, \\uXXXX included; stray tokens, raw
    control characters and unknown escapes are syntax errors; numbers / literals in unknown members are valid JSON and ignored."""
    raw = _golden_proofs()[0][1]
    text = Z.write_json(raw)
    top = json.loads(text)
    x = top['R']['x']
    other = '0x' + format(int(x, 16) ^ 1, 'x')
    # duplicate key: python's json.loads keeps the last one too
    dup = text.replace('"x":"%s"' % x, '"x":"%s","x":"%s"' % (other, x), 1)
    assert json.loads(dup) == top and Z.read_json(dup) == raw
    dup2 = text.replace('"x":"%s"' % x, '"x":"%s","x":"%s"' % (x, other), 1)
    assert Z.read_json(dup2) != raw and Z.read_json(dup2) == Z.read_json(json.dumps(json.loads(dup2)))
    # escapes: "0x..." is "0x...", "name" is "name"
    esc = text.replace('"x":"0x', '"x":"\\u0030x', 1).replace('"name"', '"na\\u006de"', 1).replace('"R"', '"\\u0052"', 1)
    assert json.loads(esc) == top and Z.read_json(esc) == raw
    for bad_text in (text.replace('"x":"0x', '"x":"\\q0x', 1),            # unknown escape
                     text.replace('"x":"0x', '"x":"\\u00zz', 1),          # bad \u
                     text.replace('"x":"0x', '"x":"0\tx', 1),             # raw control character inside a string
                     text.replace('{"R":', '{"extra":tru,"R":', 1),       # not a literal
                     text.replace('{"R":', '{"extra":01,"R":', 1),        # not a number
                     text.replace('{"R":', '{"extra":1.,"R":', 1),
                     text.replace('{"R":', '{"extra":-,"R":', 1)):
        with pytest.raises(Z.ZkError):
            Z.read_json(bad_text)
        with pytest.raises(ValueError):
            json.loads(bad_text)
    for fine in ('{"extra":true,"R":', '{"extra":-1.5e+3,"R":', '{"extra":null,"e2":[0,false,{"a":1E2}],"R":'):
        assert Z.read_json(text.replace('{"R":', fine, 1)) == raw


def test_batch_converters_equal_the_single_proof_ones():
    cases = [raw for _, raw in _golden_proofs()]
    proofs = (cases * 9)[:40]
    single = [Z.write_json(p).encode() for p in proofs]
    for threads in (1, 3, 0):
        texts, st = Z.write_json_batch(proofs, threads)
        assert st == [0] * len(proofs) and texts == single
        back, st = Z.read_json_batch(texts, threads)
        assert st == [0] * len(proofs) and back == proofs
    # a bad item gets its status and an empty result; its neighbours are converted
    broken = list(proofs[:5])
    broken[2] = broken[2][:-4]
    texts, st = Z.write_json_batch(broken, 2)
    assert st == [0, 0, 10, 0, 0] and texts[2] == b'' and texts[3] == single[3]
    bad_texts = list(single[:5])
    bad_texts[1] = bad_texts[1][:-1]
    bad_texts[4] = b'[]'
    back, st = Z.read_json_batch(bad_texts, 2)
    assert st == [0, 10, 0, 0, 10] and back[1] == b'' and back[0] == proofs[0] and back[3] == proofs[3]
    assert Z.write_json_batch([], 0) == ([], []) and Z.read_json_batch([], 0) == ([], [])
