#!/usr/bin/env python3
"""Developer probe: duration of each of the first sweeps after a device synchronisation (HIP events per sweep)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
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
    with ChainSampler(model, cfg, B, seed=1, trace_capacity=200, record_events="u16") as s:
        s.set_state(u, ev); s.set_kernel(step_size=1.2e-5)
        for _ in range(8):
            s.reset_trace(); s.run(100); model.sync()
        for gap_ms in (0.0, 1.0, 20.0):
            s.reset_trace()
            torch.cuda.synchronize(); model.sync()
            time.sleep(gap_ms * 1e-3)
            stream = torch.cuda.ExternalStream(model.stream_handle()) if hasattr(model, "stream_handle") else None
            evs = []
            t_all = []
            for i in range(24):
                model.timer_start(); s.run(1); t_all.append(model.timer_stop())
            print("gap %.0f ms, one sweep per timed call (sync after each):" % gap_ms, " ".join("%.3f" % x for x in t_all))
            s.reset_trace(); torch.cuda.synchronize(); model.sync(); time.sleep(gap_ms * 1e-3)
            out = []
            for n in (1, 2, 4, 8, 16, 32, 64):
                s.reset_trace(); torch.cuda.synchronize(); model.sync(); time.sleep(gap_ms * 1e-3)
                model.timer_start(); s.run(n); ms = model.timer_stop()
                out.append("%d: %.3f" % (n, ms))
            print("   burst of n sweeps after a sync, total ms:", "  ".join(out))
