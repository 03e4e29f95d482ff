#!/usr/bin/env python3
"""print name / calls / mean ms of the top kernels of a rocprofv3 --kernel-trace --stats directory: kstats.py <dir> [n]"""
import csv, glob, re, sys
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:n]:
        print('%-44s %5s calls  %8.3f ms mean' % (re.sub(r'\(.*', '', r['Name']).replace('void ', '')[:44], r['Calls'], float(r['AverageNs']) / 1e6))
