import ctypes, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from covid19uk_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libseirhip_stamps" + os.environ.get("STAMP_SLOT", "") + ("r2" if os.environ.get("STAMP_ROLE") == "2" else "") + ".so")
from covid19uk_amd import synth
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel
cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=int(os.environ.get("NSCAN", "0")))
cov = synth.make_covariates("uk380")
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
B = 8
u = synth.jitter_params(u0, B, scale=0.002, seed=7, T=cov.T)
ev = np.stack([events] * B)
lib = _lib.load()
lib.seir_sampler_debug_hs.argtypes = [ctypes.c_void_p, _lib.c_double_p]
with SeirModel(cov, init, max_chains=B) as model:
    with ChainSampler(model, cfg, B, seed=1, trace_capacity=10, record_events=False) as s:
        s.set_state(u, ev); s.set_kernel(step_size=1.2e-5)
        s.reset_trace(); s.run(3); model.sync()
        hs = np.empty((B, 32))
        lib.seir_sampler_debug_hs(s._s, hs.ctypes.data_as(_lib.c_double_p))
        st = hs[0, 16:32].view(np.uint64)
        if cfg["num_event_time_updates"] == 0:
            st = st.astype(np.int64)
            print("stamps (10ns ticks) deltas:", np.diff(st[:10]) * 10, "ns ; total", (st[9] - st[0]) * 10, "ns")
            print("phase 0 split: kernarg+scalars+L/Ppart", (st[10] - st[0]) * 10, "Kpart+T vectors", (st[11] - st[10]) * 10,
                  "Rpart+M vectors", (st[12] - st[11]) * 10, "barrier", (st[1] - st[12]) * 10, "ns")
        else:
            st = st.astype(np.int64)
            print("k_move_pair block 0 (se slot", int(os.environ.get("STAMP_SLOT", "1")) & 2, "): entry+finalize", (st[1]-st[0])*10, "se tables", (st[2]-st[1])*10,
                  "se propose", (st[3]-st[2])*10, "se delta", (st[4]-st[3])*10, "se accept/apply/trace", (st[5]-st[4])*10,
                  "nx tables", (st[6]-st[5])*10, "nx propose+store", (st[7]-st[6])*10, "ns; total", (st[7]-st[0])*10)
            print("   speculative role", os.environ.get("STAMP_ROLE", "1"), ": entry+pending", (st[9]-st[8])*10, "tables+propose", (st[10]-st[9])*10,
                  "store+own-rows", (st[11]-st[10])*10, "ns; total", (st[11]-st[8])*10)
            print("k_move_delta block", os.environ.get("STAMP_BLOCK", "0"), ": mv+ltab", (st[13]-st[12])*10, "band", (st[14]-st[13])*10,
                  "own rows", (st[15]-st[14])*10, "ns")
        tr = s.read_trace(3, events=False)
        print("accepts", {k: v["is_accepted"][-1].astype(int).tolist() for k, v in tr.moves.items()})
