"""The N-API addon (bindings/napi): the binding the reference's TypeScript host would call.
CPU: builds with gcc against the local Node's headers and serves the JSON wire format from JavaScript (golden digest).
GPU: proveSignatureList / verifySignatureList from JavaScript (bindings/napi/test_addon.js)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAPI = os.path.join(ROOT, 'bindings', 'napi')


def _build(tmp_path):
    if not (shutil.which('node') and shutil.which('gcc') and os.path.exists('/usr/include/node/node_api.h')):
        pytest.skip('node / gcc / node_api.h not available')
    out = str(tmp_path / 'zkattest.node')
    subprocess.check_call(['make', '-s', '-C', NAPI, 'OUT=' + out])
    return out


def test_addon_builds_and_serves_the_json_wire_format(tmp_path):
    out = _build(tmp_path)
    env = dict(os.environ, ZKATTEST_NODE=out)
    res = subprocess.run(['node', 'test_addon.js', 'cpu', os.path.join(ROOT, 'tests', 'golden', 'golden.json')], cwd=NAPI, env=env,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and 'cpu ok' in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_prove_and_verify_from_javascript(tmp_path):
    out = _build(tmp_path)
    env = dict(os.environ, ZKATTEST_NODE=out)
    res = subprocess.run(['node', 'test_addon.js', 'gpu'], cwd=NAPI, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'gpu ok' in res.stdout, res.stdout + res.stderr
