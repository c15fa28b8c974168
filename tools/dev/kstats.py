#!/usr/bin/env python3
"""Developer: print the per-kernel averages of a rocprofv3 --kernel-trace --stats --output-format csv run.
    python tools/dev/kstats.py <dir> [name-substring ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
keys = sys.argv[2:] or ["k_"]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in keys):
        print(f'{n[:90]:90s} {r["Calls"]:>7s} avg {float(r["AverageNs"])/1e3:8.2f} min {float(r["MinNs"])/1e3:8.2f} us')
