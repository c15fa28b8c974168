#!/usr/bin/env python3
"""Developer probe: is the host or the device what a short burst after a synchronisation waits for?
For bursts of n sweeps after a device synchronisation: host time to enqueue the burst, device time by HIP events."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as entry
entry.build()
from covid19uk_amd import synth
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel
cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)
cov = synth.make_covariates("uk380")
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
B = 8
u = synth.jitter_params(u0, B, scale=0.002, seed=7, T=cov.T)
ev = np.stack([events] * B)
with SeirModel(cov, init, max_chains=B) as model:
    with ChainSampler(model, cfg, B, seed=1, trace_capacity=500, record_events="u16") as s:
        s.set_state(u, ev); s.set_kernel(step_size=1.2e-5)
        for _ in range(8):
            s.reset_trace(); s.run(100); model.sync()
        for n in (1, 4, 20, 20, 64, 200, 400, 20):
            s.reset_trace(); model.sync()
            model.timer_start()
            t0 = time.perf_counter()
            s.run(n)
            t1 = time.perf_counter()
            ms = model.timer_stop()
            t2 = time.perf_counter()
            print(f"n={n:4d}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/sweep ({1e6 * (t1 - t0) / n / 31:.1f} us per launch); "
                  f"device {ms / n:.3f} ms/sweep; host wall to completion {1e3 * (t2 - t0) / n:.3f} ms/sweep")
