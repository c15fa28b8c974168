import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as entry
entry.build()
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel
from covid19uk_amd import synth
from tests import helpers as H
name = sys.argv[1] if len(sys.argv) > 1 else "micro_17x70"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
case = H.build_case(name, 3, alpha_t_sd=0.005)
cfg = dict(dmax=8, nmax=6, m=2, occult_nmax=5, num_event_time_updates=0)
B = 2
u = synth.jitter_params(case["u"], B, scale=0.05, seed=3, T=case["k"].T)
ev = np.stack([case["events"]] * B)
out = {}
for hmc in ("chunk", "fused"):
    with SeirModel(case["cov"], case["init"], max_chains=B) as model:
        with ChainSampler(model, cfg, B, seed=77, trace_capacity=2, hmc=hmc, num_leapfrog_steps=L) as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=0.0004)
            tr = s.sample(1)
            uu, _, lp = s.get_state()
            out[hmc] = (tr.theta[0].copy(), tr.hmc["target_log_prob"][0].copy(), tr.hmc["is_accepted"][0].copy(), uu.copy())
a, b = out["chunk"], out["fused"]
T, M = case["k"].T, case["k"].M
print("lp", a[1], b[1], "acc", a[2], b[2])
d = np.abs(a[3] - b[3])
print("globals", d[:, :6].max(1), "alpha chunk0", d[:, 6:6 + 63].max(1), "alpha chunk1+", d[:, 6 + 63:6 + T - 1].max(1) if T > 64 else None,
      "spatial", d[:, 6 + T - 1:].max(1))
print("spatial per row chain0", np.round(d[0, 6 + T - 1:], 12))
