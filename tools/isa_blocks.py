#!/usr/bin/env python3
"""Static instruction mix of one kernel, per basic block, from hipcc's assembly listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only zkp-ecdsa_amd/csrc/k_tom.hip -o /tmp/k_tom.s
    python tools/isa_blocks.py /tmp/k_tom.s _Z12k_tom_commitILi2ELb0 [min_valu_per_block]

Used for the issue-slot model of k_tom_commit in DESIGN.md section 8 (VALU instructions per table addition, how many of
them are v_mad_u64_u32, stray v_mov / 64 x 32-bit products the optimiser introduced)."""
import collections
import sys


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    blocks, cur = [], ['entry', collections.Counter()]
    for l in lines[start:end]:
        t = l.strip()
        if t.startswith('.LBB'):
            blocks.append(cur)
            cur = [t.rstrip(':'), collections.Counter()]
        elif t and t[0] in 'vsgdb' and not t.startswith(';'):
            cur[1][t.split()[0]] += 1
    blocks.append(cur)
    total = collections.Counter()
    for name, c in blocks:
        total.update(c)
        valu = sum(n for op, n in c.items() if op.startswith('v_'))
        if valu >= min_valu:
            top = ', '.join('%s %d' % kv for kv in c.most_common(12))
            print('%-10s VALU %5d  v_mad_u64_u32 %5d  v_mov %4d  s_nop %3d | %s' % (name, valu, c['v_mad_u64_u32'], c['v_mov_b32_e32'] + c['v_mov_b64_e32'], c['s_nop'], top))
    valu = sum(n for op, n in total.items() if op.startswith('v_'))
    print('kernel     VALU %5d  v_mad_u64_u32 %5d  (lines %d-%d)' % (valu, total['v_mad_u64_u32'], start + 1, end + 1))


if __name__ == '__main__':
    main()
