#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/profile_round.sh into the tracked profiles/ files.

usage: summarize_profiles.py <tag> <raw_dir>

Writes profiles/<tag>_bench.json, _bench_under_rocprof.json, _bench_kernel_stats.csv,
_pmc_<COUNTER>.csv (mean per kernel) and _pmc_summary.json (HBM traffic per launch of the
dominant kernel: FETCH_SIZE x calibration + WRITE_SIZE, both reported in KB by rocprofv3).

The calibration factor follows MI355X_MICROARCH.md's HBM section: on gfx950 FETCH_SIZE counts
64 B for each 128-B request, so the load-only probe with exactly known traffic
(tools/probes/se_probe.hip, k_var<1,8>: 20 B per cell) is profiled in the same way and its
known_bytes / reported_bytes ratio (about 2) scales the product kernel's reading.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

# the sampler's gradient kernels: the persistent leapfrog launch (k_leap, 15 evaluations per launch), its per-step form
# (k_se_chunk) and the plain gradient kernel k_se<GRAD=true, SRC=1, TSM> (2 of the 17 evaluations of a sweep); of each
# family the instance with the most launches
FAMILIES = {"k_leap": "void seir::k_leap<", "k_se_chunk": "void seir::k_se_chunk<",
            "k_se<GRAD=true,SRC=planes>": "void seir::k_se<true, 1"}
PROBE = "void k_var<1, 8>"
PROBE_BYTES = 8 * 384 * 384 * 20


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def counter_means(path):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            a = acc[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])]   # drop the argument list
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (s / n, n) for k, (s, n) in acc.items()}


def write_means(path, means):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "mean", "count"])
        for (k, c), (m, n) in sorted(means.items()):
            w.writerow([k, c, m, n])


def main():
    tag, raw = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    for name in ("bench.json", "bench_under_rocprof.json"):
        src = os.path.join(raw, name)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(prof, f"{tag}_{name}"))
    stats = one(os.path.join(raw, "stats", "**", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats, os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
    summary = {"source": "tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate "
                         "passes over tools/quick_sweep_bench.py --groups 1 --sweeps 20 (UK-380 x 365, 8 chains)"}
    vals, names = {}, {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = one(os.path.join(raw, f"pmc_{c}", "**", "*counter_collection.csv"))
        if not p:
            continue
        means = counter_means(p)
        write_means(os.path.join(prof, f"{tag}_pmc_{c}.csv"), means)
        for fam, prefix in FAMILIES.items():
            cands = [(n, k) for (k, cc), (m, n) in means.items() if cc == c and k.startswith(prefix)]
            if cands:
                dominant = max(cands)[1]
                vals[(fam, c)] = means[(dominant, c)][0]
                names[fam] = dominant
    factor = None
    p = one(os.path.join(raw, "pmc_probe", "**", "*counter_collection.csv"))
    if p:
        means = counter_means(p)
        write_means(os.path.join(prof, f"{tag}_pmc_probe_FETCH_SIZE.csv"), means)
        if (PROBE, "FETCH_SIZE") in means:
            kb = means[(PROBE, "FETCH_SIZE")][0]
            factor = PROBE_BYTES / (kb * 1024.0)
            summary["calibration"] = {"kernel": "tools/probes/se_probe.hip k_var<1,8> (loads only, same access widths as k_se)",
                                      "known_bytes": PROBE_BYTES, "FETCH_SIZE_KB": kb, "factor": factor,
                                      "note": "gfx950 FETCH_SIZE counts 64 B per 128-B request: factor ~2 "
                                              "(MI355X_MICROARCH.md, HBM)"}
    for fam in FAMILIES:
        if (fam, "FETCH_SIZE") in vals:
            f = factor if factor else 2.0
            traffic = vals[(fam, "FETCH_SIZE")] * 1024.0 * f + vals.get((fam, "WRITE_SIZE"), 0.0) * 1024.0
            summary[fam] = {
                "kernel_name": names[fam].replace("void seir::", ""),
                "FETCH_SIZE_KB": vals[(fam, "FETCH_SIZE")], "WRITE_SIZE_KB": vals.get((fam, "WRITE_SIZE")),
                "fetch_factor": f, "traffic_bytes_per_launch": traffic}
    # matrix-core counters of the stateless evaluation (k_eval_tiles; k_gemm in the four-launch form): one SQ pass
    p = one(os.path.join(raw, "pmc_mfma", "**", "*counter_collection.csv"))
    if p:
        means = counter_means(p)
        write_means(os.path.join(prof, f"{tag}_pmc_mfma.csv"), means)
        mf = {}
        for (k, c), (m, n) in means.items():
            if "k_gemm" in k or "k_eval_tiles" in k:
                mf.setdefault(k.replace("void seir::", ""), {})[c] = m
        if mf:
            summary["mfma_counters"] = {
                "source": "rocprofv3 --pmc (SQ counters, own pass) over tools/quick_eval_bench.py (UK-380 x 365, 8 chains)",
                "per_launch_mean": mf}
    # BASELINE config 5 (SYN-2048 x 730): bench line, kernel stats, matrix-core counters of the contraction kernels
    src = os.path.join(raw, "bench_syn2048.json")
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(prof, f"{tag}_syn2048_bench.json"))
    stats = one(os.path.join(raw, "stats_syn2048", "**", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats, os.path.join(prof, f"{tag}_syn2048_kernel_stats.csv"))
    syn = {}
    for kind in ("f32", "f64"):
        p = one(os.path.join(raw, f"pmc_mfma_syn_{kind}", "**", "*counter_collection.csv"))
        if not p:
            continue
        means = counter_means(p)
        write_means(os.path.join(prof, f"{tag}_syn2048_pmc_mfma_{kind}.csv"), means)
        for (k, c), (m, n) in means.items():
            if "k_gemm" in k:
                syn.setdefault(kind, {}).setdefault(k.replace("void seir::", ""), {})[c] = m
    if syn:
        summary["mfma_counters_syn2048"] = {
            "source": "rocprofv3 --pmc (SQ counters, own passes) over tools/quick_eval_bench.py --workload syn2048 "
                      "--form four-launch [--gemm-f32 1] (SYN-2048 x 730, 8 chains)",
            "per_launch_mean": syn}
    with open(os.path.join(prof, f"{tag}_pmc_summary.json"), "w") as fo:
        json.dump(summary, fo, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
