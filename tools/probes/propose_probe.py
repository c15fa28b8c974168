"""Developer probe (GPU box): phases of a speculative role's event-time-move proposal inside k_move_pair.
Build: hipcc ... -DSEIR_STAMPS -DSEIR_STAMP_PROPOSE -DSEIR_STAMP_SLOT=<1|3> -DSEIR_STAMP_ROLE=<1|2> -> libseirhip_prop.so"""
import ctypes, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from covid19uk_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[1])
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
lib = _lib.load()
lib.seir_sampler_debug_hs.argtypes = [ctypes.c_void_p, _lib.c_double_p]
with SeirModel(cov, init, max_chains=B) as model:
    with ChainSampler(model, cfg, B, seed=1, trace_capacity=10, record_events=False) as s:
        s.set_state(u, ev); s.set_kernel(step_size=1.2e-5)
        s.reset_trace(); s.run(3); model.sync()
        hs = np.empty((B, 32))
        lib.seir_sampler_debug_hs(s._s, hs.ctypes.data_as(_lib.c_double_p))
        st = hs[0, 16:32].view(np.uint64).astype(np.int64)
        d = lambda a, b: (st[b] - st[a]) * 10
        print("role: entry+pending", d(12, 13), "| tables", d(13, 4), "| select rows", d(4, 5), "| stage rows", d(5, 6),
              "| select days", d(6, 7), "| bounds (mins)", d(7, 8), "| finish lanes", d(8, 9), "| compact+store", d(9, 14),
              "| own rows", d(14, 15), "ns; total", d(12, 15))
