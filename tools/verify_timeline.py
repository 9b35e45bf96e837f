#!/usr/bin/env python3
"""Idle time inside ONE zk_verify_batch_device call, from a rocprofv3 --kernel-trace database (rocpd).

    python tools/verify_timeline.py r_results.db [call_index_from_end=0]

A call begins with k_v_header and ends with its last k_v_final.  Prints the call's wall time, the time during which no kernel of any stream ran, the
longest idle windows with the kernels on either side, and the time during which ONLY dependent-chain kernels (the bucket reductions, the P-256 pass's
reductions) ran -- the windows another chunk's throughput-bound kernels could fill."""
import re
import sqlite3
import sys

CHAINS = ('k_msm_red', 'k_msm_final', 'k_msm_coef', 'k_pm_reduce', 'k_pm_final', 'k_pm_big', 'k_words_to_host', 'k_tom_commit<1')


def union(iv):
    out = []
    for a, b in sorted(iv):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = db.execute('select name, stream, start, end from kernels order by start').fetchall()
    rows = [(re.sub(r'\(.*\)$', '', n).replace('void ', ''), st, s, e) for n, st, s, e in rows]
    calls, cur = [], None
    for r in rows:
        if r[0].startswith('k_v_header') and (cur is None or any(x[0].startswith('k_v_final') for x in cur)) and (cur is None or r[2] >= max(x[3] for x in cur)):
            if cur:
                calls.append(cur)
            cur = []
        if cur is not None:
            cur.append(r)
    if cur:
        calls.append(cur)
    calls = [c for c in calls if any(x[0].startswith('k_v_final') for x in c)]
    call = calls[-1 - back]
    end = max(x[3] for x in call if x[0].startswith('k_v_final'))
    call = [x for x in call if x[2] < end]
    t0 = call[0][2]
    busy = union([(x[2], x[3]) for x in call])
    wall = (end - t0) / 1e6
    idle = wall - sum(b - a for a, b in busy) / 1e6
    print(f'call: {len(call)} kernels, wall {wall:.2f} ms, no kernel running {idle:.2f} ms')
    gaps = [(busy[i + 1][0] - busy[i][1], busy[i][1], busy[i + 1][0]) for i in range(len(busy) - 1)]
    for g, a, b in sorted(gaps, reverse=True)[:8]:
        before = max((x for x in call if x[3] <= a + 1), key=lambda x: x[3])
        after = min((x for x in call if x[2] >= b - 1), key=lambda x: x[2])
        print(f'  idle {g / 1e3:8.1f} us at {(a - t0) / 1e6:7.2f} ms   after {before[0][:36]:36s} before {after[0][:36]}')
    heavy = union([(x[2], x[3]) for x in call if not x[0].startswith(CHAINS)])
    only_chain = sum(b - a for a, b in busy) / 1e6 - sum(b - a for a, b in heavy) / 1e6
    print(f'only dependent-chain kernels running: {only_chain:.2f} ms')
    by = {}
    for x in call:
        by[x[0]] = by.get(x[0], 0) + (x[3] - x[2]) / 1e6
    print('largest kernels (ms summed):', ', '.join(f'{k[:28]} {v:.2f}' for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]))


if __name__ == '__main__':
    main()
