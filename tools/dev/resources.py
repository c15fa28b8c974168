#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS use from `hipcc -Rpass-analysis=kernel-resource-usage` output.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage ... 2> res.txt; python tools/dev/resources.py res.txt [filter]"""
import re
import shutil
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
filt = shutil.which("c++filt")
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip()

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    if filt:
        name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name).replace('void seir::', '')
    if flt in name:
        print(f"{name:64s} sgpr={g('SGPRs'):3d} vgpr={g('VGPRs'):3d} agpr={g('AGPRs'):3d} scratch={g('ScratchSize .bytes/lane.'):4d} "
              f"occ={g('Occupancy .waves/SIMD.')} lds={g('LDS Size .bytes/block.')}")
