#!/usr/bin/env python3
"""Per-kernel resource usage of the gfx950 code object inside libzkattest_hip.so (AMDGPU metadata notes): VGPRs (incl. AGPRs),
SGPRs, spills, scratch bytes, LDS, and the resulting waves per SIMD (512 VGPRs per SIMD lane on gfx950).

    python tools/kernel_meta.py [lib.so] [--csv]

Used by tests/test_abi_and_host.py (no kernel of the product may use scratch memory unless listed) and for DESIGN.md."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUNDLER = '/opt/rocm/llvm/bin/clang-offload-bundler'
READELF = '/opt/rocm/llvm/bin/llvm-readelf'


def code_object(lib):
    """Extract the gfx950 code object of the fat binary section (.hip_fatbin, clang offload bundle) into a temp file."""
    tmp = tempfile.mkdtemp(prefix='zkmeta')
    fat = os.path.join(tmp, 'fat.bin')
    subprocess.check_call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', lib, fat])
    data = open(fat, 'rb').read()
    outs = []
    pos = 0
    while True:   # one bundle per translation unit
        i = data.find(b'__CLANG_OFFLOAD_BUNDLE__', pos)
        if i < 0:
            break
        j = data.find(b'__CLANG_OFFLOAD_BUNDLE__', i + 24)
        blob = data[i:j if j > 0 else len(data)]
        b = os.path.join(tmp, 'b%d.bin' % len(outs))
        open(b, 'wb').write(blob)
        co = os.path.join(tmp, 'co%d.o' % len(outs))
        r = subprocess.run([BUNDLER, '--type=o', '--unbundle', '--input=' + b, '--output=' + co, '--targets=hipv4-amdgcn-amd-amdhsa--gfx950'],
                           capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            outs.append(co)
        pos = i + 24
    return outs


def kernels(lib):
    res = {}
    for co in code_object(lib):
        txt = subprocess.run([READELF, '--notes', co], capture_output=True, text=True).stdout
        for blk in re.split(r'\n\s+- \.agpr_count:', txt)[1:]:
            blk = '.agpr_count:' + blk
            g = lambda k, d=0: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, d])[1]
            name = g('name', '?')
            vg = int(g('vgpr_count'))
            res[name] = {'vgpr': vg, 'agpr': int(g('agpr_count')), 'sgpr': int(g('sgpr_count')), 'vgpr_spill': int(g('vgpr_spill_count')),
                         'sgpr_spill': int(g('sgpr_spill_count')), 'scratch': int(g('private_segment_fixed_size')), 'lds': int(g('group_segment_fixed_size')),
                         'max_wg': int(g('max_flat_workgroup_size')), 'waves_per_simd': min(8, 512 // max(1, (vg + 7) // 8 * 8))}
    return res


def main():
    lib = next((a for a in sys.argv[1:] if not a.startswith('--')), os.path.join(ROOT, 'zkp-ecdsa_amd', 'lib', 'libzkattest_hip.so'))
    ks = kernels(lib)
    if '--csv' in sys.argv:
        print('kernel,vgpr,agpr,sgpr,vgpr_spill,sgpr_spill,scratch_bytes,lds_bytes,waves_per_simd')
        for n, k in sorted(ks.items()):
            print('%s,%d,%d,%d,%d,%d,%d,%d,%d' % (n, k['vgpr'], k['agpr'], k['sgpr'], k['vgpr_spill'], k['sgpr_spill'], k['scratch'], k['lds'], k['waves_per_simd']))
        return
    for n, k in sorted(ks.items(), key=lambda kv: -kv[1]['vgpr']):
        print('%-70s vgpr %3d (agpr %3d) sgpr %3d spill v%d s%d scratch %4d lds %6d waves/SIMD %d' % (n[:70], k['vgpr'], k['agpr'], k['sgpr'], k['vgpr_spill'],
                                                                                                       k['sgpr_spill'], k['scratch'], k['lds'], k['waves_per_simd']))


if __name__ == '__main__':
    main()
