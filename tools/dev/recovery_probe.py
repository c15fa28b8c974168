#!/usr/bin/env python3
"""Developer probe (GPU box): posterior z-scores of the generating parameters over several simulated NI-11 epidemics."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as entry
entry.build()
from covid19uk_amd import synth
from covid19uk_amd.inference import inference as inf
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel

CFG = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5, num_bursts=8, num_burst_samples=500)
names = ("psi", "sigma_space", "beta_area", "gamma0", "gamma1", "alpha_0")


class Collect:
    def __init__(self):
        self.theta = []
    def write_samples(self, d, first_dim_offset=0):
        self.theta.append(np.stack([d[n] for n in names], 1))
    def write_results(self, d, first_dim_offset=0):
        pass


cov = synth.make_covariates("ni11")
B = 4
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    events, init, truth = synth.simulate_epidemic(cov, seed=1000 + seed)
    M, T = cov.M, cov.T
    P = 6 + T - 1 + M
    posts = [Collect() for _ in range(B)]
    with SeirModel(cov, init, max_chains=B) as model:
        with ChainSampler(model, CFG, B, seed=4 + seed, trace_capacity=800) as s:
            s.set_state(np.zeros((B, P)), np.stack([events] * B))
            inf.run_mcmc(s, CFG, posts, log=open("/dev/null", "w"))
    draws = np.stack([np.concatenate(p.theta)[inf.warmup_size():] for p in posts])
    pooled = draws.reshape(-1, 6)
    z = [(truth[n] - pooled[:, i].mean()) / pooled[:, i].std() for i, n in enumerate(names)]
    lo, hi = np.percentile(pooled, [2.5, 97.5], axis=0)
    inside = [bool(lo[i] <= truth[n] <= hi[i]) for i, n in enumerate(names)]
    W = draws.var(axis=1, ddof=1).mean(0); Bv = draws.mean(axis=1).var(axis=0, ddof=1)
    print(seed, "z", np.round(z, 2), "inside", inside, "rhat", np.round(np.sqrt(1 + Bv / W), 3),
          "mean", np.round(pooled.mean(0), 3), "events", events.sum((0, 1)), flush=True)
