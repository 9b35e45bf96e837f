#!/usr/bin/env python3
"""Kernel timeline of ONE proof per call (the reference's only call shape: src/zkpAttestList.ts:104-184, one proof per proveSignatureList /
verifySignatureList call, timed by bench/zkpAttestList.bench.ts:48-62).

    cd /tmp && rocprofv3 --kernel-trace -d $OUT -o r -- python $ROOT/tools/b1_timeline.py run
    python tools/b1_timeline.py parse $OUT/r_results.db [> profiles/rNN_b1_timeline.txt]

`run`: ring of 1 024 keys, 16-bit combs, B = 1, six prove + verify calls.  `parse`: the kernels of the LAST prove call and the LAST verify call in
start order -- stream, start offset, duration, gap to the previous kernel's end on any stream -- and the per-kernel totals of each call."""
import os
import re
import sqlite3
import sys


def run():
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    eng.set_comb_bits(16)
    eng.set_params(*eng.synth_params(2024), 80)
    nk = int(os.environ.get('B1_RING', '1024'))   # B1_RING=65536: the bench's ring
    ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, nk, 4)
    eng.set_ring(ring, nk)
    eng.set_lanes(1)
    eng.set_chunk(1)
    pin = Z.PinnedBuffer(8 << 20)
    a = (msg[:32], sig[:64], pk[:64], which[:1], seeds[:32])
    for k in range(int(os.environ.get('B1_CALLS', '6'))):
        dt, hout, hoff, hst = eng.prove_batch_host_raw(*a, out=pin)
        vdt, vok, vst = eng.verify_batch_host_raw(a[0], hout, hoff, 1)
        assert sum(vok) == 1
        print('call %d: prove %.3f ms, verify %.3f ms' % (k, 1e3 * dt, 1e3 * vdt))


def short(n):
    n = re.sub(r'\(.*\)$', '', n).replace('void ', '')
    return re.sub(r'<.*>', lambda m: m.group(0) if len(m.group(0)) < 24 else '<..>', n)


def parse(path):
    db = sqlite3.connect(path)
    rows = db.execute('select name, stream, start, end from kernels order by start').fetchall()
    rows = [(short(n), st, s, e) for n, st, s, e in rows]
    # a verify call starts at k_v_header (or k_v_unpack_scan), a prove call at the first kernel after a verify call's k_v_final
    calls, cur, kind = [], [], None
    for r in rows:
        is_vstart = r[0].startswith('k_v_header') and not (cur and kind == 'verify' and not any(x[0].startswith('k_v_final') for x in cur))
        ended = cur and kind == 'verify' and any(x[0].startswith('k_v_final') for x in cur) and not r[0].startswith('k_words_to_host')
        if is_vstart or ended:
            if cur:
                calls.append((kind, cur))
            cur, kind = [], 'verify' if is_vstart else 'prove'
        if kind is None:
            kind = 'setup'
        cur.append(r)
    if cur:
        calls.append((kind, cur))
    for want in ('prove', 'verify'):
        sel = [c for k, c in calls if k == want and len(c) >= 10]
        if not sel:
            continue
        call = sel[-1]
        t0 = call[0][2]
        end_prev = t0
        print('== last %s call: %d kernels, first start -> last end %.3f ms, sum of kernel durations %.3f ms'
              % (want, len(call), (max(x[3] for x in call) - t0) / 1e6, sum(x[3] - x[2] for x in call) / 1e6))
        print('%-44s %6s %10s %9s %8s' % ('kernel', 'stream', 'start us', 'dur us', 'gap us'))
        for n, st, s, e in call:
            print('%-44s %6s %10.1f %9.1f %8.1f' % (n[:44], st, (s - t0) / 1e3, (e - s) / 1e3, (s - end_prev) / 1e3))
            end_prev = max(end_prev, e)
        by = {}
        for n, st, s, e in call:
            v = by.setdefault(n, [0, 0])
            v[0] += 1
            v[1] += e - s
        print('-- totals')
        for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            print('%-44s x%-3d %9.1f us' % (n[:44], c, d / 1e3))


def modes(path):
    """The shortest and the longest prove call of a trace side by side (start offset and duration of every kernel): what differs between the calls of one process."""
    db = sqlite3.connect(path)
    rows = [(short(n), st, s, e) for n, st, s, e in db.execute('select name, stream, start, end from kernels order by start').fetchall()]
    calls, cur = [], []
    for r in rows:   # a prove call: k_rng_prepass (its first kernel behind the input copy) .. k_status_out
        if r[0].startswith('k_rng_prepass') and not any(x[0].startswith('k_scan') for x in cur):
            cur = [r]
        elif cur:
            cur.append(r)
            if r[0].startswith('k_status_out'):
                calls.append(cur)
                cur = []
    calls = [c for c in calls[2:] if not any(x[0].startswith('k_v_') for x in c)]
    span = lambda c: (c[-1][3] - c[0][2]) / 1e3
    calls.sort(key=span)
    print('%d prove calls, first kernel -> k_status_out: %s us' % (len(calls), ' '.join('%.0f' % span(c) for c in calls)))
    a, b = calls[0], calls[-1]
    print('%-28s %8s | %9s %8s | %9s %8s' % ('kernel', 'stream', 'start', 'dur', 'start', 'dur'))
    for x, y in zip(a, b):
        flag = '  <--' if abs((y[3] - y[2]) - (x[3] - x[2])) > 8e3 or abs((y[2] - b[0][2]) - (x[2] - a[0][2])) > 30e3 else ''
        print('%-28s %8s | %9.1f %8.1f | %9.1f %8.1f   %s%s' % (x[0][:28], x[1], (x[2] - a[0][2]) / 1e3, (x[3] - x[2]) / 1e3, (y[2] - b[0][2]) / 1e3, (y[3] - y[2]) / 1e3, y[0][:20] if y[0] != x[0] else '', flag))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run()
    elif sys.argv[1] == 'modes':
        modes(sys.argv[2])
    else:
        parse(sys.argv[2])
