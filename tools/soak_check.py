import sys, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as e; e.build()
from covid19uk_amd import synth
from covid19uk_amd.seir import SeirModel
from covid19uk_amd.sampler import ChainSampler
from oracle import seir_oracle as so
from tests import helpers as H
case = H.build_case("uk380", 11)
B = 8
u = synth.jitter_params(case["u"], B, scale=0.002, seed=5, T=case["k"].T)
ev = np.stack([case["events"]] * B)
cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)
with SeirModel(case["cov"], case["init"], max_chains=B) as model:
    with ChainSampler(model, cfg, B, seed=123, trace_capacity=100, record_events=False) as s:
        s.set_state(u, ev); s.set_kernel(step_size=2e-5)
        s.set_adaptation(adapt_step_size=True, num_adaptation_steps=300)
        for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
            s.reset_trace(); s.run(100); model.sync()
            if it == 2: s.set_adaptation(adapt_step_size=False)
        tr = s.read_trace(100, events=False)
        u1, ev1, lp_run = s.get_state()
        late = s.pair_timeouts()
        print("k_move_pair hand-off time-outs per chain:", late)
        assert not late.any()
        print("hmc acc", tr.hmc["is_accepted"].mean(), "step", tr.hmc["step_size"][-1, 0], {k: float(v["is_accepted"].mean()) for k, v in tr.moves.items()})
        for b in range(B):
            want = H.c_oracle_eval(case["k"], u1[b], ev1[b], stable=1)
            st = so.compute_state(case["init"], ev1[b], closed=True)
            print(b, lp_run[b], want, abs(lp_run[b]-want)/abs(want), st.min() >= 0)
            assert abs(lp_run[b]-want) <= 1e-9*abs(want) and st.min() >= 0
print("soak ok")
