#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel, mean counter values and kernel duration."""
import csv, sys, collections
d = sys.argv[1]
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(d + '/pmc_counter_collection.csv')):
    ctr[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(d + '/pmc_kernel_trace.csv')):
    dur[r['Kernel_Name'][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, cs in ctr.items():
    if "conv" not in k and "bn_" not in k and "ssim" not in k and "wino" not in k:
        continue
    n = max(len(v) for v in cs.values())
    us = sum(dur[k]) / max(len(dur[k]), 1)
    print('%s  (%d dispatches, avg %.1f us)' % (k, n, us))
    for c, v in sorted(cs.items()):
        print('    %-28s %16.0f' % (c, sum(v) / len(v)))
