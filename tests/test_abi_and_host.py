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


def test_pool_reads_device_locality_from_sysfs(tmp_path, monkeypatch):
    """zk_pool binds every shard's host thread to the CPUs next to its device and places the shard's output region on that NUMA
    node: both come from /sys/bus/pci/devices/<bus id>/{numa_node, local_cpulist}.  The parser, on a fake tree (no GPU needed)."""
    import ctypes as C
    import zkp_ecdsa_amd as Z
    d = tmp_path / 'bus' / 'pci' / 'devices' / '0000:c1:00.0'
    d.mkdir(parents=True)
    (d / 'numa_node').write_text('3\n')
    (d / 'local_cpulist').write_text('48-63,176-191\n')
    d2 = tmp_path / 'bus' / 'pci' / 'devices' / '0000:05:00.0'
    d2.mkdir(parents=True)
    (d2 / 'numa_node').write_text('-1\n')
    (d2 / 'local_cpulist').write_text('7\n')
    monkeypatch.setenv('ZKATTEST_SYSFS_ROOT', str(tmp_path))
    L = Z.lib()
    numa, cpus = C.c_int(-5), (C.c_int * 64)()
    n = L.zk_pool_test_locality(b'0000:C1:00.0', C.byref(numa), cpus, 64)   # HIP prints upper-case hex, sysfs is lower-case
    assert (n, numa.value, list(cpus[:n])) == (32, 3, list(range(48, 64)) + list(range(176, 192)))
    n = L.zk_pool_test_locality(b'0000:05:00.0', C.byref(numa), cpus, 64)
    assert (n, numa.value, cpus[0]) == (1, -1, 7)
    n = L.zk_pool_test_locality(b'0000:ff:00.0', C.byref(numa), cpus, 64)     # unknown device: no affinity, no placement
    assert (n, numa.value) == (0, -1)
    # a failed create leaves NULL behind and says why
    h = C.c_void_p(1)
    assert L.zk_pool_create(None, 0, C.byref(h)) == 14 and not h.value
