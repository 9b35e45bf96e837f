"""not-gpu: the plain-JavaScript (V8 BigInt) restatement oracle/js/zkattest_ref.js against the committed golden vectors -- the
same vectors that pin the Python and C restatements and the HIP engine.  It runs on the Node 12 of this image (the reference
itself needs Node >= 24 + tsc + typedjson); skipped where there is no `node`."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, 'oracle', 'js', 'zkattest_ref.js')
GOLD = os.path.join(ROOT, 'tests', 'golden', 'golden.json')

pytestmark = pytest.mark.skipif(shutil.which('node') is None, reason='no node in this environment')


def _run(*args):
    p = subprocess.run(['node', JS] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [json.loads(l) for l in p.stdout.decode().splitlines() if l.startswith('{')]
    return p.returncode, lines, p.stderr.decode()


def test_reference_kats_in_v8_bigint():
    rc, lines, err = _run('kats', GOLD)     # test/bignum/big.test.ts:19-21, test/proofGK/interpolate.test.ts:19-26
    assert rc == 0 and lines == [{'kats': True}], err


def test_js_restatement_reproduces_the_golden_proofs_byte_for_byte():
    """secLevel 20 full proof, the planted-rejection RNG stream, and one secLevel-80 proof over a 37-key ring (padded to 64):
    SHA-256 and length of the ZKA1 bytes, the number of RNG fills consumed, and the JS verifier's verdict."""
    rc, lines, err = _run('golden', GOLD, 'small_full', 'rejection_stream', 'ring37_sec80')
    assert rc == 0, (lines, err)
    assert [l['case'] for l in lines] == ['small_full', 'rejection_stream', 'ring37_sec80']
    for l in lines:
        assert l['sha256_ok'] and l['fills_ok'] and l['verified'], l


def test_js_verifier_rejects_what_the_other_restatements_reject(tmp_path):
    """A golden file whose expected digest is wrong must make the driver fail (the comparison is not vacuous)."""
    g = json.load(open(GOLD))
    g['small_full']['proofs'][0]['sha256'] = '00' * 32
    bad = tmp_path / 'bad.json'
    bad.write_text(json.dumps(g))
    rc, lines, err = _run('golden', str(bad), 'small_full')
    assert rc == 1 and lines and not lines[0]['sha256_ok'] and lines[0]['verified']
