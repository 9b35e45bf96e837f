"""The packed wire layout ZKA1P (include/zkattest.h: 33-byte Tom-256 coordinates, magic "ZK1P").  CPU tier: the host converters
zk_proof_pack / zk_proof_unpack against a restatement of the layout in Python over the golden proofs, round trips, refusals, the JSON writers
on packed input.  GPU tier (-m gpu): the prover's writers emit the packed form natively -- byte for byte pack(oracle proof) through every entry
point -- and the verifier, fed packed proofs, returns the (ok, status) pairs of the oracle on the expanded form, mutants included."""
import hashlib
import json
import os

import pytest

import zkp_ecdsa_amd as Z
from zka1_mutants import Layout, mutants

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'golden.json')))


def pack_py(raw, n, sec):
    """ZKA1 -> ZKA1P, written down from the layout table: every 36-byte coordinate loses its three leading zero bytes."""
    lay = Layout(raw, n, sec)
    tom = set()
    for o, sz in lay.top_points()[2:]:
        tom.update((o, o + 36))
    for i in range(sec):
        for o, sz in lay.rep_points(i)[1:]:
            tom.update((o, o + 36))
        if not (lay.bits >> i) & 1:
            for o in lay.padd_points(i):
                tom.update((o, o + 36))
    for o in lay.gk_points():
        tom.update((o, o + 36))
    out, pos = bytearray(), 0
    for o in sorted(tom):
        out += raw[pos:o]
        assert raw[o:o + 3] == b'\0\0\0'
        pos = o + 3
    out += raw[pos:]
    out[0:4] = b'ZK1P'
    out[4:8] = len(out).to_bytes(4, 'big')
    return bytes(out)


def test_host_converters_follow_the_layout_table():
    rec = GOLD['small_full']['proofs'][0]
    raw, sec = bytes.fromhex(rec['proof']), GOLD['small_full']['sec']
    n = int.from_bytes(raw[12:16], 'big')
    packed = Z.pack_proof(raw)
    assert packed == pack_py(raw, n, sec)
    z = sum(1 for i in range(sec) if not (int.from_bytes(raw[16:32], 'big') >> i) & 1)
    assert len(packed) == 292 + 324 * sec + 3200 * z + 360 * n + 32 and len(raw) == 304 + 336 * sec + 3392 * z + 384 * n + 32
    assert Z.unpack_proof(packed) == raw
    assert Z.write_json(packed) == Z.write_json(raw)                       # the text does not know the layout
    texts, tst = Z.write_json_batch([packed, raw, packed[:-4]], threads=2)
    as_text = lambda t: t.decode() if isinstance(t, (bytes, bytearray)) else t
    assert as_text(texts[0]) == as_text(texts[1]) == Z.write_json(raw) and list(tst) == [0, 0, 10]
    # refusals: the other layout's magic, a length that is not the header's, challenge bits above secLevel, a coordinate that has no 33-byte form
    for bad in (raw[:-4], raw[:4] + (len(raw) + 4).to_bytes(4, 'big') + raw[8:] + bytes(4), raw[:16] + b'\x80' + raw[17:], raw[:160] + b'\x01' + raw[161:]):
        with pytest.raises(Z.ZkError) as e:
            Z.pack_proof(bad)
        assert e.value.status == 10
    with pytest.raises(Z.ZkError):
        Z.pack_proof(packed)
    with pytest.raises(Z.ZkError):
        Z.unpack_proof(raw)
    for bad in (packed[:-1], packed[:31], b'ZK1P', packed[:4] + bytes(4) + packed[8:]):
        with pytest.raises(Z.ZkError):
            Z.unpack_proof(bad)


def _vseeds(n, tag):
    return b''.join(hashlib.sha256(tag + i.to_bytes(4, 'big')).digest() for i in range(n))


@pytest.mark.gpu
@pytest.mark.parametrize('nkeys,B,chunk,lanes', [(8, 5, 4096, 2), (1024, 300, 128, 3)])
def test_the_prover_emits_the_packed_layout_natively(nkeys, B, chunk, lanes):
    import coracle as CO
    eng = Z.Engine(0)
    eng.set_comb_bits(16)
    nh, tg, th = eng.synth_params(71)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(71, nkeys, B)
    eng.set_ring(ring, nkeys)
    eng.set_chunk(chunk), eng.set_lanes(lanes)
    plain, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert not any(st)
    n = (nkeys - 1).bit_length()
    eng.set_wire(True)
    assert eng.proof_max_size() == 292 + 324 * 80 + 3200 * 80 + 360 * n + 32
    packed, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert not any(st)
    assert packed == [pack_py(p, n, 80) for p in plain] == [Z.pack_proof(p) for p in plain]
    # page-locked sink (sliced D2H) and the streamed form give the same packed bytes
    pin = Z.PinnedBuffer(eng.proof_max_size() * B)
    _, _, off, st = eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=pin)
    assert [bytes(pin.view[off[b]:off[b + 1]]) for b in range(B)] == packed
    t = eng.prove_submit(msg, sig, pk, which, seeds, pin)
    off, st = eng.prove_wait(t)
    assert [bytes(pin.view[off[b]:off[b + 1]]) for b in range(B)] == packed
    # the oracle made the same proofs (ZKA1), the engine verifies its packed ones, streamed too
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, nkeys)
    k = min(B, 8)
    exp, _ = octx.prove_batch(msg[:32 * k], sig[:64 * k], pk[:64 * k], which[:k], seeds=seeds[:32 * k], nthreads=8)
    assert [Z.unpack_proof(p) for p in packed[:k]] == exp
    vs = _vseeds(B, b'p')
    assert eng.verify_batch(msg, packed, vseeds=vs) == ([1] * B, [0] * B)
    tv = eng.verify_submit(msg, pin, off, B, vs)
    ok, vst = eng.verify_wait(tv)
    assert list(ok) == [1] * B and not any(vst)
    # handed the other layout, either verifier refuses every proof as malformed
    assert eng.verify_batch(msg[:64], plain[:2], vseeds=vs[:64]) == ([0, 0], [10, 10])
    eng.set_wire(False)
    assert eng.verify_batch(msg[:64], packed[:2], vseeds=vs[:64]) == ([0, 0], [10, 10])
    assert eng.verify_batch(msg[:64], plain[:2], vseeds=vs[:64]) == ([1, 1], [0, 0])
    pin.free()
    eng.close()


@pytest.mark.gpu
def test_device_side_offsets_of_packed_proofs_are_checked_before_the_expansion_is_sized():
    """zk_verify_batch_device with ZKA1P proofs: the offsets live in HBM and the staging of the expanded proofs is sized from off[B] -- overlapping ranges
    ([0, L, 0, L]: every proof expands again) would run past it.  A non-monotonic or misaligned device array is refused (ZK_E_ARG) before anything is
    expanded; the well-formed array verifies."""
    import torch
    eng = Z.Engine(0)
    eng.set_comb_bits(16)
    eng.set_params(*eng.synth_params(72), 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(72, 8, 4)
    eng.set_ring(ring, 8)
    eng.set_wire(True)
    packed, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert not any(st)
    dev = torch.device('cuda', 0)
    blob = b''.join(packed)
    d_pr = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    d_msg = torch.frombuffer(bytearray(msg), dtype=torch.uint8).to(dev)
    d_vs = torch.frombuffer(bytearray(_vseeds(4, b'o')), dtype=torch.uint8).to(dev)
    d_ok, d_st = torch.zeros(4, dtype=torch.uint8, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)
    offs, o = [0], 0
    for p_ in packed:
        o += len(p_)
        offs.append(o)

    def run(off_list):
        d_off = torch.tensor(off_list, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        eng.verify_batch_device(4, d_msg.data_ptr(), d_pr.data_ptr(), d_off.data_ptr(), d_vs.data_ptr(), d_ok.data_ptr(), d_st.data_ptr())
        return d_ok.cpu().tolist(), d_st.cpu().tolist()
    assert run(offs) == ([1] * 4, [0] * 4)
    L = len(packed[0])
    for bad in ([0, L, 0, L, 2 * L], [0, L, 2 * L, L, 4 * L], [0, L + 2, 2 * L, 3 * L, 4 * L]):
        with pytest.raises(Z.ZkError) as e:
            run(bad)
        assert e.value.status == 14 and 'offsets' in str(e.value)
    assert run(offs) == ([1] * 4, [0] * 4)   # the context is usable afterwards
    eng.close()


@pytest.mark.gpu
def test_packed_mutants_get_the_verdicts_of_their_expanded_form():
    """Every mutant of the sweep that still has a packed form (structure intact, coordinates below 2^264), packed: the engine on the packed bytes
    == the oracle on the ZKA1 bytes, exact status codes; plus damage done to the PACKED bytes themselves (length, magic, truncation)."""
    import coracle as CO
    S, nkeys = 9001, 8
    eng = Z.Engine(0)
    nh, tg, th = eng.synth_params(S)
    eng.set_params(nh, tg, th, 80)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(S, nkeys, 2)
    eng.set_ring(ring, nkeys)
    octx = CO.OracleCtx(nh, tg, th, 80)
    octx.set_ring(ring, nkeys)
    proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    names, msgs, plain, packed = [], [], [], []
    for name, j, raw in mutants(proofs, 3, S, S):
        try:
            pk_ = Z.pack_proof(raw)
        except Z.ZkError:
            continue
        names.append(name), msgs.append(msg[32 * j:32 * j + 32]), plain.append(raw), packed.append(pk_)
    assert len(packed) >= 300, len(packed)
    eng.set_wire(True)
    for tag in (b'm0', b'm1'):
        vs = _vseeds(len(packed), tag)
        g = eng.verify_batch(b''.join(msgs), packed, vseeds=vs)
        o = octx.verify_batch(b''.join(msgs), plain, nthreads=16, vseeds=vs)
        bad = [(names[i], (g[0][i], g[1][i]), (o[0][i], o[1][i])) for i in range(len(packed)) if (g[0][i], g[1][i]) != (o[0][i], o[1][i])]
        assert not bad, (tag, len(bad), bad[:10])
    assert {(1, 0), (0, 0), (0, 3), (0, 4), (0, 8), (0, 10)} <= set(zip(*g))
    good = Z.pack_proof(proofs[0])
    broken = [good, good[:-4], good + bytes(4), b'ZKA1' + good[4:], good[:4] + (len(good) - 4).to_bytes(4, 'big') + good[8:], good[:32], good[:12] + (4).to_bytes(4, 'big') + good[16:],
              good[:16] + b'\x01' + good[17:], good]
    g = eng.verify_batch(msg[:32] * len(broken), broken, vseeds=_vseeds(len(broken), b'b'))
    assert g == ([1] + [0] * (len(broken) - 2) + [1], [0] + [10] * (len(broken) - 2) + [0])
    eng.close()
