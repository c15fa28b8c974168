#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/ (run HERE; commit the outputs).

The reference ships no golden vector for this path and cannot be imported
(tensorflow / tensorflow_probability / gemlib are not installed: plain
ModuleNotFoundError), so the expected values come from the build's own oracle:
oracle/seir_oracle.py for every case and, for the micro cases, 50-digit mpmath
evaluations of the same formulas.  The fixture pins the
ORACLE (tests/test_golden.py re-computes it on every CPU run) and is the common
reference the HIP path is compared with on the GPU box.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import mcmc_oracle as mo  # noqa: E402
from oracle import seir_oracle as so  # noqa: E402
from tests import helpers as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CASES = [("micro_1x1", 101), ("micro_2x3", 102), ("micro_3x5", 103), ("ni11", 104)]
CFG = dict(dmax=8, nmax=6, m=2, occult_nmax=5, num_event_time_updates=2)


def main():
    out = {}
    for name, seed in CASES:
        c = H.build_case(name, seed, alpha_t_sd=0.005)
        rng = np.random.default_rng(seed)
        u = c["u"] + 0.1 * rng.normal(size=c["u"].shape)
        T = c["k"].T
        u[6:6 + T - 1] = 0.005 * rng.normal(size=T - 1)
        lp, g = so.joint_log_prob_and_grad(u, c["events"], c["k"])
        lp_ref = so.joint_log_prob(u, c["events"], c["k"], "reference")
        cov = c["cov"]
        d = dict(C=cov.C, N=cov.N, W=cov.W, weekday=cov.weekday, area=cov.area, adjacency=cov.adjacency,
                 initial_state=c["init"], events=c["events"], u=u, logp=lp, logp_reference_form=lp_ref, grad=g)
        if name.startswith("micro"):
            d["logp_mpmath"] = float(so.joint_log_prob_mp(u, c["events"], c["k"], dps=50))
        for key, v in d.items():
            out[f"{name}/{key}"] = np.asarray(v)
    # a short sampler trace on the smallest spatial case (shared Philox stream)
    c = H.build_case("micro_3x5", 103, alpha_t_sd=0.005)
    ch = mo.OracleChain(c["k"], CFG, c["u"], c["events"], seed=2021, chain_id=3, t_range=(2, 5))
    ch.eps = 0.002
    sweeps = [ch.sweep_once() for _ in range(6)]
    out["trace/theta"] = np.stack([s["theta"] for s in sweeps])
    out["trace/events"] = np.stack([s["events"] for s in sweeps])
    out["trace/hmc_accept"] = np.array([s["hmc"]["is_accepted"] for s in sweeps])
    out["trace/hmc_logp"] = np.array([s["hmc"]["target_log_prob"] for s in sweeps])
    for key in ("move/S->E", "move/E->I", "occult/S->E", "occult/E->I"):
        out[f"trace/{key}/delta"] = np.stack([s[key]["proposed_delta"] for s in sweeps])
        out[f"trace/{key}/accept"] = np.array([s[key]["is_accepted"] for s in sweeps])
        out[f"trace/{key}/logp"] = np.array([s[key]["target_log_prob"] for s in sweeps])
    path = os.path.join(OUT, "seir_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
