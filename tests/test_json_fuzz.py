"""Mutation fuzzing of the JSON reader (zk_proof_from_json, host-only code in csrc/api_json.hip): whatever text arrives, the
call must return either a proof or a ZkError -- no crash, no hang, no out-of-bounds write -- and everything it accepts must
serialise back to a fixed point.  The reference's readJson (src/serde.ts:21-33) faces the same untrusted input."""
import json
import os
import random


import zkp_ecdsa_amd as Z

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'golden.json')))


def _texts():
    out = []
    for name in ('small_full', 'ring6_sec80'):
        for rec in GOLD[name]['proofs']:
            if 'proof' in rec:
                out.append(Z.write_json(bytes.fromhex(rec['proof'])))
    return out[:3]


def _try(text):
    try:
        raw = Z.read_json(text)
    except Z.ZkError as e:
        assert e.status in (10, 12, 14), e.status     # 'error deserializing' / size / argument
        return None
    # accepted: the binary form must serialise and parse back to itself
    again = Z.write_json(raw)
    assert Z.read_json(again) == raw
    return raw


def test_byte_level_mutations_never_crash_the_reader():
    rnd = random.Random(20240926)
    texts = _texts()
    assert texts
    accepted = 0
    for it in range(1500):
        t = bytearray(rnd.choice(texts).encode())
        for _ in range(rnd.choice((1, 1, 2, 5, 20))):
            op = rnd.randrange(6)
            pos = rnd.randrange(len(t))
            if op == 0:
                t[pos] = rnd.randrange(256)
            elif op == 1:
                del t[pos:pos + rnd.choice((1, 2, 17, 400))]
            elif op == 2:
                t[pos:pos] = bytes(rnd.randrange(256) for _ in range(rnd.choice((1, 3, 40))))
            elif op == 3:
                t[pos:pos] = rnd.choice((b'{', b'}', b'[', b']', b'"', b',', b':', b'\\', b'0x', b'-0x', b'null', b'1e999', b'\x00'))
            elif op == 4:
                del t[pos:]
            else:
                a = rnd.randrange(len(t))
                t[pos:pos] = t[a:a + rnd.choice((10, 200, 3000))]
            if not t:
                t = bytearray(b' ')
        try:
            s = t.decode('utf-8')
        except UnicodeDecodeError:
            s = t.decode('latin-1')
        if _try(s) is not None:
            accepted += 1
    assert accepted < 1500          # the mutations do bite


def test_structural_mutations_never_crash_the_reader():
    rnd = random.Random(7)
    base = [json.loads(t) for t in _texts()]

    def paths(o, p=()):
        yield p
        if isinstance(o, dict):
            for k, v in o.items():
                yield from paths(v, p + (k,))
        elif isinstance(o, list):
            for i, v in enumerate(o):
                yield from paths(v, p + (i,))

    junk = [None, True, 0, -1, 2 ** 70, 1.5, '', '0x', '0xzz', '-0x1', '0x' + 'f' * 70, '0x' + 'f' * 5000, [], {}, [[]], {'group': {}},
            {'group': {'name': 'p256'}}, {'group': {'name': 'nope'}, 'x': '0x1', 'y': '0x2'}, 'tomEdwards256', ['0x1'] * 100]
    for it in range(600):
        doc = json.loads(json.dumps(rnd.choice(base)))
        for _ in range(rnd.choice((1, 1, 3))):
            ps = [p for p in paths(doc) if p]
            p = rnd.choice(ps)
            parent = doc
            for k in p[:-1]:
                parent = parent[k]
            op = rnd.randrange(4)
            if op == 0:
                parent[p[-1]] = rnd.choice(junk)
            elif op == 1:
                if isinstance(parent, dict):
                    del parent[p[-1]]
                else:
                    parent.pop(p[-1])
            elif op == 2 and isinstance(parent, list):
                parent.extend(parent[:] * rnd.choice((1, 30)))
            elif isinstance(parent, dict):
                parent[rnd.choice(('x', 'k', 'zz', '__type', 'proof', 'expProof'))] = rnd.choice(junk)
        _try(json.dumps(doc))
    # deep nesting and huge inputs are refused, not recursed into without bound
    for s in ('[' * 100000, '{"R":' * 50000, '{"expProof":[' + '{},' * 200000 + '{}]}', '"' + 'a' * (1 << 22), '0x' + 'f' * (1 << 22)):
        assert _try(s) is None


def test_output_capacity_is_respected():
    import ctypes as C
    t = _texts()[0].encode()
    L = Z.lib()
    need = C.c_uint64(0)
    for cap in (0, 1, 100, 4096):
        buf = C.create_string_buffer(cap + 64)
        C.memset(buf, 0x5A, cap + 64)
        rc = L.zk_proof_from_json(t, len(t), buf, cap, C.byref(need))
        assert rc != 0 and need.value > cap
        assert buf.raw[cap:] == b'\x5a' * 64      # nothing written past the capacity
    raw = Z.read_json(t.decode())
    for cap in (0, 10, len(t) // 2):
        buf = C.create_string_buffer(cap + 64)
        C.memset(buf, 0x5A, cap + 64)
        rc = L.zk_proof_to_json(raw, len(raw), buf, cap, C.byref(need))
        assert rc != 0 and need.value > cap and buf.raw[cap:] == b'\x5a' * 64


def test_binary_side_mutations_never_crash_the_writer():
    """zk_proof_to_json walks an untrusted ZKA1 buffer (header lengths, per-rep layout bits): truncations and bit flips must end
    in a ZkError or in a text that reads back to the same bytes."""
    rnd = random.Random(99)
    raws = []
    for name in ('small_full', 'ring6_sec80'):
        for rec in GOLD[name]['proofs']:
            if 'proof' in rec:
                raws.append(bytes.fromhex(rec['proof']))
    raws = raws[:3]
    ok = 0
    for it in range(1200):
        b = bytearray(rnd.choice(raws))
        op = rnd.randrange(5)
        if op == 0:
            del b[rnd.randrange(len(b)):]
        elif op == 1:
            for _ in range(rnd.choice((1, 4, 64))):
                b[rnd.randrange(min(len(b), rnd.choice((32, 400, len(b)))))] ^= 1 << rnd.randrange(8)
        elif op == 2:
            b[4:8] = rnd.randrange(1 << 32).to_bytes(4, 'big')        # total length field
        elif op == 3:
            b[8:16] = bytes(rnd.randrange(256) for _ in range(8))      # secLevel, n
        else:
            b += bytes(rnd.randrange(256) for _ in range(rnd.choice((1, 4, 1000))))
        try:
            text = Z.write_json(bytes(b))
        except Z.ZkError as e:
            assert e.status in (10, 12, 14), e.status
            continue
        ok += 1
        back = Z.read_json(text)
        assert Z.write_json(back) == text
    assert 0 < ok < 1200


def test_fast_path_and_tree_reader_agree_on_every_mutation(monkeypatch):
    """The reader has a fast path for text in exactly the writer's shape (csrc/api_json.hip: literal-by-literal matching, no node
    tree) that must never decide on its own: for every mutated text the outcome -- the proof bytes or the status -- with the fast
    path (default) and without it (ZKATTEST_JSON_NO_FAST=1: the tolerant tree reader alone) is the same."""
    rnd = random.Random(33)
    texts = _texts()

    def outcome(s):
        try:
            return Z.read_json(s)
        except Z.ZkError as e:
            return e.status

    checked = accepted = 0
    for it in range(600):
        t = bytearray(rnd.choice(texts).encode())
        for _ in range(rnd.choice((0, 1, 1, 2, 4))):
            pos = rnd.randrange(len(t))
            op = rnd.randrange(5)
            if op == 0:
                t[pos] = rnd.choice(b'0123456789abcdefx",:{}[] ')
            elif op == 1:
                del t[pos:pos + rnd.choice((1, 2, 64))]
            elif op == 2:
                t[pos:pos] = rnd.choice((b' ', b'0', b'00', b'"', b',', b'\n', b'A', b'{"a":1},'))
            elif op == 3:   # swap two members of the top-level object: only the tree reader takes it, and it must still give the proof
                s0 = t.decode('latin-1')
                i = s0.find(',"comS1":')
                j = s0.find(',"keyXcom":')
                if 0 < i < j:
                    t = bytearray((s0[:1] + s0[i + 1:j] + ',' + s0[1:i] + s0[j:]).encode('latin-1'))
            else:
                t[pos:pos + 1] = bytes([t[pos]]).upper()
        s = t.decode('latin-1')
        monkeypatch.delenv('ZKATTEST_JSON_NO_FAST', raising=False)
        a = outcome(s)
        monkeypatch.setenv('ZKATTEST_JSON_NO_FAST', '1')
        b = outcome(s)
        assert a == b, (it, a if isinstance(a, int) else len(a), b if isinstance(b, int) else len(b))
        checked += 1
        accepted += not isinstance(a, int)
    assert checked == 600 and 0 < accepted < 600
