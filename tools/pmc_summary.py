#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py <dir> [<dir> ...]"""
import csv, glob, os, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get('Kernel_Name', r.get('Kernel Name', '?')).split('(')[0]
            c = r.get('Counter_Name'); v = float(r.get('Counter_Value', 0))
            agg[name][c] += v
            calls[name][c] += 1
names = sorted(agg, key=lambda n: -max(agg[n].values()))
for n in names:
    print(n[:60])
    for c, v in sorted(agg[n].items()):
        print('   %-28s total %.6g   per-launch %.6g   launches %d' % (c, v, v / calls[n][c], calls[n][c]))
