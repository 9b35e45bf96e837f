"""not-gpu: the C-ABI library loads and exports every symbol include/zkattest.h declares (no compute calls), and the
host-side helpers of bench.py behave."""
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import zkp_ecdsa_amd as Z
    hdr = open(os.path.join(ROOT, 'include', 'zkattest.h')).read()
    declared = sorted(set(re.findall(r'\b(zk_[a-z0-9_]+)\s*\(', hdr)))
    assert set(declared) == set(Z.SYMBOLS), (set(declared) ^ set(Z.SYMBOLS))
    if not os.path.exists(Z.LIB_PATH):
        Z.build()
    lib = ctypes.CDLL(Z.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s
    lib.zk_strerror.restype = ctypes.c_char_p
    assert lib.zk_strerror(3) == b'T[i] is at infinity'
    assert Z.STATUS_TEXT[6] == "Points don't add up!"


def test_product_never_imports_oracle():
    """The product path (package + csrc) must not reference oracle/ in any form."""
    pkg = os.path.join(ROOT, 'zkp-ecdsa_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')) or f == 'Makefile':
                txt = open(os.path.join(dirpath, f), errors='ignore').read()
                assert 'coracle' not in txt and 'zkattest_ref' not in txt and 'zkattest_oracle' not in txt, os.path.join(dirpath, f)


def test_bench_helpers():
    import bench
    seeds = bytes(range(64))
    assert bench.rank_seeds(seeds, 0) == seeds
    s1, s2 = bench.rank_seeds(seeds, 1), bench.rank_seeds(seeds, 2)
    assert len(s1) == 64 and s1 != seeds and s1 != s2
    wt, wq, ring = bench.nominal_modmuls(16)
    assert (wt, wq, ring) == ((162 + 26 * 40 + 64) * 4064 + 320 * 3184, (163 + 40) * 4448 + 5568, 2 * 65536 * 16)
    assert bench.host_cores() >= 1


def test_c_program_compiles_and_links_against_the_abi(tmp_path):
    """include/zkattest.h is plain C and the shared library is all a C host needs (examples/c_abi_demo.c)."""
    import shutil
    import subprocess
    import zkp_ecdsa_amd as Z
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which('gcc'):
        pytest.skip('no gcc')
    out = tmp_path / 'zk_demo'
    libdir = os.path.dirname(Z.LIB_PATH)
    subprocess.check_call(['gcc', '-std=c99', '-O1', '-Wall', '-Werror', '-I' + os.path.join(root, 'include'),
                           os.path.join(root, 'examples', 'c_abi_demo.c'), '-o', str(out), '-L' + libdir, '-lzkattest_hip',
                           '-Wl,-rpath,' + libdir])
    assert out.exists()
