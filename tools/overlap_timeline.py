#!/usr/bin/env python3
"""Timeline of ONE prove step from a rocprofv3 --kernel-trace database (rocpd): which kernels run beside which.

    python tools/overlap_timeline.py gpurun_out/r05/trace_base/r_results.db [step_index_from_end=0]

Splits the dispatches into steps at gaps of the host (a step = one zk_prove_batch_device call: its kernels are separated from the next call's by the
host's synchronisation), takes the chosen step and prints: wall time, the time during which at least one GPU-filling ("heavy": k_tom_commit*,
k_exp_commit*) kernel was running, the time during which only light kernels ran, the time nothing ran, and per kernel name the summed duration
and how much of it ran beside a heavy kernel of ANOTHER stream."""
import re
import sqlite3
import sys

HEAVY = ('k_tom_commit', 'k_exp_commit')


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def intersect(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if lo < hi:
            out.append([lo, hi])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = db.execute('select name, stream, start, end from kernels order by start').fetchall()
    rows = [(re.sub(r'\(.*\)$', '', n).replace('void ', ''), st, s, e) for n, st, s, e in rows]
    # steps: a k_rng_prepass that starts after everything before it has ended and follows a gap begins a new call; simpler: cut at idle gaps > 150 us
    steps, cur, last_end, seen_out = [], [], None, False
    for r in rows:   # a new call begins with a k_rng_prepass launched after everything before it -- a k_status_out included -- has finished
        if last_end is not None and r[2] >= last_end and r[0].startswith('k_rng_prepass') and seen_out:
            steps.append(cur)
            cur, seen_out = [], False
        if r[0].startswith('k_rng_prepass') and not cur:
            pass
        cur.append(r)
        seen_out = seen_out or r[0].startswith('k_status_out')
        last_end = r[3] if last_end is None else max(last_end, r[3])
    if cur:   # the last step ends with its last k_status_out (what follows is the harness)
        k = max(i for i, r in enumerate(cur) if r[0].startswith('k_status_out')) if any(r[0].startswith('k_status_out') for r in cur) else len(cur) - 1
        steps.append(cur[:k + 1])
    steps = [s for s in steps if sum(1 for r in s if r[0].startswith('k_tom_commit')) >= 6]   # prove steps only
    print('%d prove steps found: wall ms %s' % (len(steps), ' '.join('%.1f' % ((max(r[3] for r in s) - s[0][2]) / 1e6) for s in steps)))
    S = steps[-1 - back]
    t0, t1 = S[0][2], max(r[3] for r in S)
    heavy = union([(s, e) for n, st, s, e in S if n.startswith(HEAVY)])
    anyk = union([(s, e) for n, st, s, e in S])
    wall = t1 - t0
    print('step wall %.2f ms; heavy running %.2f ms (%.1f %%); only light %.2f ms; idle %.2f ms; streams %d; dispatches %d' %
          (wall / 1e6, length(heavy) / 1e6, 100.0 * length(heavy) / wall, (length(anyk) - length(heavy)) / 1e6, (wall - length(anyk)) / 1e6, len({r[1] for r in S}), len(S)))
    # degree of heavy concurrency
    ev = []
    for n, st, s, e in S:
        if n.startswith(HEAVY):
            ev.append((s, 1)), ev.append((e, -1))
    ev.sort()
    deg, last, hist = 0, t0, {}
    for t, d in ev:
        hist[deg] = hist.get(deg, 0) + (t - last)
        deg += d
        last = t
    print('heavy kernels in flight: ' + ', '.join('%d: %.1f ms' % (k, v / 1e6) for k, v in sorted(hist.items()) if v))
    per = {}
    for n, st, s, e in S:
        others = union([(a, b) for m, st2, a, b in S if m.startswith(HEAVY) and st2 != st and b > s and a < e])
        ov = length(intersect([[s, e]], others))
        p = per.setdefault(n, [0, 0, 0])
        p[0] += 1
        p[1] += e - s
        p[2] += ov
    print('%-28s %5s %10s %10s %8s' % ('kernel', 'calls', 'sum ms', 'avg us', 'beside-heavy'))
    for n, (c, d, ov) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
        print('%-28s %5d %10.2f %10.1f %7.0f %%' % (n[:28], c, d / 1e6, d / c / 1e3, 100.0 * ov / d if d else 0))
    if len(sys.argv) > 3:   # dump the step's dispatches
        for n, st, s, e in S:
            print('%9.3f %9.3f  s%-3d %s' % ((s - t0) / 1e6, (e - t0) / 1e6, st, n))


if __name__ == '__main__':
    main()
