#!/usr/bin/env python3
"""Developer: from a rocprofv3 kernel trace of a sampler run, how the sweep time and the kernels' own durations change
with the number of sweeps since the last idle gap.   python tools/dev/trace_ramp.py <k_kernel_trace.csv>"""
import csv, sys
import numpy as np
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
st = np.array([r[0] for r in rows]); en = np.array([r[1] for r in rows])
names = [r[2] for r in rows]
# a sweep ends with k_record; runs are separated by idle gaps > 20 us
is_rec = np.array(["k_record" in n for n in names])
gap = np.concatenate([[0], st[1:] - en[:-1]])
run_id = np.cumsum(gap > 20000)
out = {}
for rid in np.unique(run_id):
    idx = np.where(run_id == rid)[0]
    recs = [i for i in idx if is_rec[i]]
    if len(recs) < 20:
        continue
    prev_end = st[idx[0]]
    lo = idx[0]
    for k, i in enumerate(recs):
        sl = slice(lo, i + 1)
        wall = en[i] - prev_end
        busy = int((en[sl] - st[sl]).sum())
        se = [en[j] - st[j] for j in range(lo, i + 1) if "k_se_chunk" in names[j]]
        mp = [en[j] - st[j] for j in range(lo, i + 1) if "k_move_pair" in names[j]]
        out.setdefault(k, []).append((wall, busy, np.mean(se) if se else 0, np.mean(mp) if mp else 0))
        prev_end = en[i]; lo = i + 1
print("sweep index since the idle gap: wall us, sum of kernel durations us, mean k_se_chunk us, mean k_move_pair us  (median over runs; n runs)")
for k in [0, 1, 2, 4, 8, 12, 16, 19, 30, 50, 99, 150, 199, 300, 399]:
    if k in out:
        a = np.array(out[k], dtype=float)
        m = np.median(a, axis=0) / 1e3
        print(f"  {k:4d}: {m[0]:8.1f} {m[1]:8.1f} {m[2]:7.2f} {m[3]:7.2f}   ({len(a)})")
