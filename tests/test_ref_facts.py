"""not-gpu: this build against FACTS extracted from the reference tree (tests/golden/ref_facts.json, written by
tools/ref_constants.py in the build container: curve constants, JSON member names in declaration order, the point lists of the
Fiat-Shamir hashes, challenge width, repetition counts).  The restatements in oracle/, the generated device constants, the
JavaScript facade and the JSON writer of the product are all checked against the same extracted data -- a transcription slip in
one of them (a swapped hash argument, a renamed member, a wrong constant) shows up here without running the reference."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FACTS = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'ref_facts.json')))


def _consts(group):
    return [int(c, 16) for c in FACTS['groups'][group]['constants']]


def test_fixture_is_current_when_the_reference_tree_is_present():
    if not os.path.isdir('/root/reference/src'):
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'ref_constants.py'), '/root/reference', '/dev/stdout'], capture_output=True, text=True, check=True).stdout
    assert json.loads(out[:out.rindex('}') + 1]) == FACTS


def test_curve_constants_everywhere():
    p, a, b, n, gx, gy = _consts('p256')
    t, ta, td, tq, tgx, tgy = _consts('tomEdwards256')
    assert tq == p and a == p - 3
    # python restatement
    import zkattest_ref as R
    assert (R.p256.p, R.p256.a, R.p256.b, R.p256.order) == (p, a, b, n) and R.p256.name == FACTS['groups']['p256']['name']
    assert (R.tomEdwards256.p, R.tomEdwards256.a, R.tomEdwards256.d, R.tomEdwards256.order) == (t, ta, td, tq)
    assert R.tomEdwards256.name == FACTS['groups']['tomEdwards256']['name']
    g1, g2 = R.p256.generator().toAffine(), R.tomEdwards256.generator().toAffine()
    assert tuple(g1) == (gx, gy) and tuple(g2) == (tgx, tgy)
    # generator of the device constants (tools/gen_consts.py)
    src = open(os.path.join(ROOT, 'tools', 'gen_consts.py')).read()
    vals = {m.group(1): int(m.group(2), 16) for m in re.finditer(r'^(\w+) = (0x[0-9a-f]+)$', src, re.M)}
    assert (vals['t'], vals['q'], vals['n'], vals['tom_a'], vals['tom_d'], vals['tom_gx'], vals['tom_gy'], vals['p256_b'], vals['p256_gx'], vals['p256_gy']) == \
        (t, p, n, ta, td, tgx, tgy, b, gx, gy)
    # the C and JavaScript restatements and the JavaScript facade carry the same literals
    for rel in ('oracle/zkattest_oracle.c', 'oracle/js/zkattest_ref.js', 'bindings/napi/zkattest.js'):
        text = open(os.path.join(ROOT, rel)).read().lower().replace('_', '')
        lits = {int(h, 16) for h in re.findall(r'(?:0x|")([0-9a-f]{40,})', text)}
        words = re.findall(r'0x([0-9a-f]{8,16})u?l*', text)
        for v in (p, n, b, gx, gy, t, ta, td, tgx, tgy):
            hexv = '%x' % v
            if v in lits:
                continue
            # limb-wise tables (C): every 32-bit word of the value appears
            ws = {hexv[max(0, i - 8):i].lstrip('0') or '0' for i in range(len(hexv), 0, -8)}
            assert all(any(w.lstrip('0') == x for w in words) for x in ws), (rel, hex(v))


def test_json_member_names_and_order():
    import zkp_ecdsa_amd as Z
    gold = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'golden.json')))
    proof = bytes.fromhex(gold['small_full']['proofs'][0]['proof'])
    obj = json.loads(Z.write_json(proof))
    M = {k: [m['name'] for m in v] for k, v in FACTS['json_members'].items()}

    def keys(o):
        return [k for k in o if k != '__type']

    def in_order(sub, full):
        it = iter(full)
        return all(k in it for k in sub)
    assert keys(obj) == M['SignatureProofList']
    pts = {'WeierstrassPoint': M['WeierstrassPoint'], 'TEdwardsPoint': M['TEdwardsPoint']}
    assert keys(obj['R']) == pts['WeierstrassPoint'] and keys(obj['keyXcom']) == pts['TEdwardsPoint'] and keys(obj['R']['group']) == M['Group']
    assert obj['R']['group']['name'] == FACTS['groups']['p256']['name'] and obj['keyXcom']['group']['name'] == FACTS['groups']['tomEdwards256']['name']
    seen = set()
    for e in obj['expProof']:
        assert in_order(keys(e), M['ExpProof']) and set(keys(e)) <= set(M['ExpProof'])
        required = [m['name'] for m in FACTS['json_members']['ExpProof'] if not m['optional']]
        assert all(k in e for k in required)
        seen.add('proof' in e)
        if 'proof' in e:
            pa = e['proof']
            assert keys(pa) == M['PointAddProof']
            for k in ('pi_8', 'pi_10', 'pi_11', 'pi_13'):
                assert keys(pa[k]) == M['MultProof']
            for k in ('pi_x', 'pi_y'):
                assert keys(pa[k]) == M['EqualityProof']
            assert keys(pa['pi_x']['t_x']) == M['Scalar'] and pa['pi_x']['t_x']['k'].startswith(FACTS['misc']['bigint_prefix'])
    assert seen == {True, False}
    gk = obj['membershipProof']
    assert keys(gk) == M['GKProof']
    for m in FACTS['json_members']['GKProof']:
        assert isinstance(gk[m['name']], list) == m['array']
    # SystemParametersList / PedersenParams of the JavaScript facade
    js = open(os.path.join(ROOT, 'bindings', 'napi', 'zkattest.js')).read()
    assert re.search(r"toJSON\(\) \{ return \{ NistGroup: .*ProofGroup: .*SecLevel: ", js) and re.search(r"toJSON\(\) \{ return \{ c: .*g: .*h: ", js)
    assert M['SystemParametersList'] == ['NistGroup', 'ProofGroup', 'SecLevel'] and M['PedersenParams'] == ['c', 'g', 'h']


def test_fiat_shamir_transcripts_of_the_restatements():
    """Every literal hashPoints list of the reference appears, identifier for identifier, in the Python restatement (a line-by-line
    transliteration keeps the names), and the two array-fed hashes are filled in the reference's order."""
    src = open(os.path.join(ROOT, 'oracle', 'zkattest_ref.py')).read()
    lists = [re.sub(r'\s+', '', m) for m in re.findall(r'hashPoints\(\[([^\]]*)\]\)', src)]
    for call in FACTS['hash_points_calls']:
        if 'points' in call:
            assert ','.join(call['points']) in lists, call
    fills = FACTS['hash_points_arrays']
    assert fills['exp_arr'][:4] == ['Px.p, Py.p', 'A[i as number]', 'Tx[i as number].p', 'Ty[i as number].p']
    assert re.search(r'arr = \[Px\.p, Py\.p\]\s+for i in range\(secparam\):\s+arr \+= \[A\[i\], Tx\[i\]\.p, Ty\[i\]\.p\]', src)
    assert fills['exp_arr'][4:] == ['Px, Py', 'pi[i as number].A', 'pi[i as number].Tx', 'pi[i as number].Ty']
    assert re.search(r'arr = \[Px, Py\]\s+for e in pi:\s+arr \+= \[e\.A, e\.Tx, e\.Ty\]', src)
    assert fills['gk_commitments'] == 'cl.concat(ca).concat(cb).concat(cd)' and 'hashPoints(cl + ca + cb + cd, statement)' in src   # statement = b'' outside the hardened mode
    assert 'hashPoints(proof.cl + proof.ca + proof.cb + proof.cd, statement)' in src
    # challenge width, bit order, repetition counts
    import zkattest_ref as R
    mi = FACTS['misc']
    pts = [R.p256.generator()]
    import hashlib
    assert R.hashPoints(pts) == int.from_bytes(hashlib.sha256(pts[0].toBytes()).digest()[:mi['challenge_bytes']], 'big')
    assert mi['challenge_bytes'] == 10 and mi['verify_reps'] == 20 and mi['default_sec_level'] == 80 and mi['exp_challenge_lsb_first'] and mi['point_prefix_byte'] == 4
    hdr = open(os.path.join(ROOT, 'zkp-ecdsa_amd', 'csrc', 'engine.h')).read()
    assert re.search(r'#define VK %d\b' % mi['verify_reps'], hdr)
