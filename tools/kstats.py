#!/usr/bin/env python3
"""Print per-kernel averages of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir>"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'upsp' in r['Name']:
        m = re.search(r'(\w+)(<[^>]*>)?\(', r['Name'].replace('(anonymous namespace)::', ''))
        print("%-36s calls %4s avg %9.1f us" % ((m.group(1) + (m.group(2) or '')) if m else r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3))
