#!/usr/bin/env python3
"""Developer experiments: tools/quick_sweep_bench.py against a variant library built by build_variant.sh.
    python tools/dev/sweep_variant.py <tag> [quick_sweep_bench args]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
tag = sys.argv.pop(1)
from covid19uk_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "dev", "variants", f"libseirhip_{tag}.so")
import __graft_entry__ as entry  # noqa: E402

entry.build = lambda: None
import quick_sweep_bench  # noqa: E402

quick_sweep_bench.main()
