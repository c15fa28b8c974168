#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS use from `hipcc -Rpass-analysis=kernel-resource-usage` remarks.

    python tools/dev/resources.py remarks.txt [filter]              # table
    python tools/dev/resources.py --json out.json remarks.txt       # {kernel: {...}} (what __graft_entry__.build() keeps
                                                                    #  next to the library as kernel_resources.json)
The parser lives in covid19uk_amd/kernel_resources.py."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from covid19uk_amd.kernel_resources import parse  # noqa: E402

if __name__ == "__main__":
    argv = sys.argv[1:]
    if argv and argv[0] == "--json":
        res = parse(open(argv[2]).read())
        json.dump(res, open(argv[1], "w"), indent=1, sort_keys=True)
        print(f"{len(res)} kernels -> {argv[1]}")
    else:
        res = parse(open(argv[0]).read())
        flt = argv[1] if len(argv) > 1 else ""
        for name, e in sorted(res.items()):
            if flt in name:
                print(f"{name:40s} sgpr={e['sgpr']:3d} vgpr={e['vgpr']:3d} agpr={e['agpr']:3d} scratch={e['scratch_bytes_per_lane']:4d} "
                      f"spill(s/v)={e['sgpr_spill']}/{e['vgpr_spill']} occ={e['occupancy_waves_per_simd']} lds={e['lds_bytes_per_block']}")
