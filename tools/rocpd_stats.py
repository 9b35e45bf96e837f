"""Per-kernel summary of a rocprofv3 `--kernel-trace --stats` run whose output is the rocpd SQLite database
(`<prefix>_results.db`, the default output format of this ROCm's rocprofv3).  Prints the same columns as the
`kernel_stats.csv` of the csv output format (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs) and,
per kernel, how many launches ran while another kernel of this process was in flight on a different stream
(Overlapped) -- with two lanes the per-launch durations of overlapped launches are time-shared, not kernel time.

    python tools/rocpd_stats.py gpurun_out/prof/r_results.db > profiles/rNN_rocprofv3_kernel_stats.csv
"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute('select name, stream, start, end from kernels order by start').fetchall()
    stats = {}
    total = 0
    # overlap: a launch overlaps if any launch on another stream intersects [start, end)
    active = []
    flags = [False] * len(rows)
    for i, (name, stream, s, e) in enumerate(rows):
        active = [(j, st, ee) for (j, st, ee) in active if ee > s]
        for j, st, ee in active:
            if st != stream:
                flags[i] = flags[j] = True
        active.append((i, stream, e))
    for i, (name, stream, s, e) in enumerate(rows):
        short = re.sub(r'\(.*\)$', '', name)
        d = e - s
        st = stats.setdefault(short, [0, 0, 1 << 62, 0, 0])
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
        st[4] += 1 if flags[i] else 0
        total += d
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","Overlapped"')
    for name, (n, t, mn, mx, ov) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%d,%.1f,%.2f,%d,%d,%d' % (name, n, t, t / n, 100.0 * t / total, mn, mx, ov))


if __name__ == '__main__':
    main(sys.argv[1])
