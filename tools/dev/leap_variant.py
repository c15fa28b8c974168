#!/usr/bin/env python3
"""Developer A/B (GPU box): ms per sweep and the leapfrog section (HIP events around k_leap) for one library.
    python tools/dev/leap_variant.py <variant|product> [chains] [sweeps] [workload]   -- one line of JSON; call it alternately per variant"""
import json, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from covid19uk_amd import _lib
if sys.argv[1] != "product":
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "dev", "variants", f"libseirhip_{sys.argv[1]}.so")
from covid19uk_amd import synth
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)
WL = sys.argv[4] if len(sys.argv) > 4 else "uk380"
cov = synth.make_covariates(WL)
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
u = synth.jitter_params(u0, B, scale=0.002, seed=7, T=cov.T)
ev = np.stack([events] * B)
with SeirModel(cov, init, max_chains=B) as model:
    with ChainSampler(model, cfg, B, seed=1, trace_capacity=n, log=None) as s:
        s.set_state(u, ev)
        s.set_kernel(step_size=1.2e-5 if WL == "uk380" else 2e-6)
        s.reset_trace(); s.run(50); model.sync()
        s.reset_trace()
        model.timer_start()
        s.run(n)
        ms = model.timer_stop()
        tr = s.read_trace(n, events=False)
        lm, ll, le = s.time_leapfrog(200)
        print(json.dumps({"lib": sys.argv[1], "workload": WL, "ms_per_sweep": round(ms / n, 5), "leap_us": round(1e3 * lm, 2), "launches": ll, "evals": le,
                          "hmc_acc": float(tr.hmc["is_accepted"].mean()), "lp": float(tr.hmc["target_log_prob"][-1, 0])}))
