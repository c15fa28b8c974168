#!/usr/bin/env python3
"""Developer soak (GPU box): the persistent launches' hand-offs must not depend on timing.  Two runs of N sweeps from the same
seed, the second with workgroup skew (debug_skew), must leave bit-identical chains -- a torn or stale hand-off word, a missed
wait or a race would show as a difference -- and no wait may time out.   python tools/dev/ll_soak.py [chains] [sweeps]"""
import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from covid19uk_amd import synth
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)
cov = synth.make_covariates("uk380")
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
u = synth.jitter_params(u0, B, scale=0.002, seed=7, T=cov.T)
ev = np.stack([events] * B)
out = []
for skew in (0, 3, 1):
    with SeirModel(cov, init, max_chains=B) as model:
        model.set_option(debug_skew=skew)
        with ChainSampler(model, cfg, B, seed=5, trace_capacity=100, record_events=False, log=None) as s:
            s.set_state(u, ev); s.set_kernel(step_size=1.2e-5)
            for it in range(N // 100):
                s.reset_trace(); s.run(100)
            model.sync()
            assert not s.pair_timeouts().any(), s.pair_timeouts()
            tr = s.read_trace(100, events=False)
            out.append((s.get_state(), tr))
            print("skew", skew, "hmc acc", float(tr.hmc["is_accepted"].mean()), "lp", float(tr.hmc["target_log_prob"][-1, 0]), flush=True)
(q0, e0, l0), t0 = out[0]
for (q, e, l), t in out[1:]:
    assert np.array_equal(q0, q) and np.array_equal(e0, e) and np.array_equal(l0, l)
    assert np.array_equal(t0.theta, t.theta)
print(f"soak ok: {B} chains x {N} sweeps, three runs bit-identical")
