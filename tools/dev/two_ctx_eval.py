#!/usr/bin/env python3
"""Developer probe: stateless evaluations issued alternately on two contexts (two streams) against one context."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import __graft_entry__ as entry
entry.build()
from covid19uk_amd import synth
from covid19uk_amd.seir import SeirModel
cov = synth.make_covariates("uk380")
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
B = 8
u = synth.jitter_params(u0, B, T=cov.T)
dev = torch.device("cuda:0")
ut = torch.tensor(u, device=dev)
evt = torch.tensor(np.stack([events] * B), device=dev)
N = 200
for nctx in (1, 2, 3):
    models = [SeirModel(cov, init, max_chains=B) for _ in range(nctx)]
    lps = [torch.empty(B, dtype=torch.float64, device=dev) for _ in range(nctx)]
    gs = [torch.empty(B, u.shape[1], dtype=torch.float64, device=dev) for _ in range(nctx)]
    for grad in (False, True):
        for i in range(6):
            models[i % nctx].log_prob_dev(ut, evt, lps[i % nctx], gs[i % nctx] if grad else None)
        for m in models:
            m.sync()
        t0 = time.perf_counter()
        for i in range(N):
            models[i % nctx].log_prob_dev(ut, evt, lps[i % nctx], gs[i % nctx] if grad else None)
        for m in models:
            m.sync()
        dt = time.perf_counter() - t0
        print(f"{nctx} context(s), grad={grad}: {B * N / dt:.0f} evaluations/s ({1e6 * dt / N:.1f} us per batch of {B})")
    for m in models:
        m.close()
