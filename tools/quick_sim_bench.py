#!/usr/bin/env python3
"""Time seir_simulate (forward chain-binomial simulation) on the UK-380 workload."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="uk380")
    ap.add_argument("--draws", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=84)
    ap.add_argument("--initial-step", type=int, default=200)
    a = ap.parse_args()
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd import model_spec, synth
    from covid19uk_amd.posterior import predict as pp
    from covid19uk_amd.seir import SeirModel
    cov = synth.make_covariates(a.workload)
    events, init, truth = synth.simulate_epidemic(cov)
    theta = synth.pack_params(truth, cov.M, cov.T)
    n, S = a.draws, a.steps
    th = np.repeat(theta[None], n, 0)
    state = model_spec.compute_state(init, events)[:, a.initial_step, :]
    dc = model_spec.derive_constants(cov)
    par = th[:, :5].copy()
    a_path = pp.log_baseline_path(th[:, 5], th[:, 6:6 + cov.T - 1], a.initial_step, S)
    spatial = th[:, 6 + cov.T - 1:]
    W, wd = pp.clipped(dc.W, a.initial_step, S), pp.clipped(dc.weekday_c, a.initial_step, S)
    init_n = np.repeat(state[None], n, 0)
    with SeirModel(cov, init, max_chains=1) as model:
        model.simulate(par[:8], a_path[:8], spatial[:8], W, wd, init_n[:8], seed=1)
        t0 = time.perf_counter()
        ev = model.simulate(par, a_path, spatial, W, wd, init_n, seed=1)
        dt = time.perf_counter() - t0
    print(json.dumps({"workload": a.workload, "draws": n, "days": S, "seconds_incl_pcie": dt,
                      "draw_days_per_s": n * S / dt, "cell_draws_per_s": 3 * n * S * cov.M / dt,
                      "mean_daily_new_infections": float(ev[:, :, :, 0].sum() / n / S)}, indent=1))


if __name__ == "__main__":
    main()
