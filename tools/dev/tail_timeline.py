#!/usr/bin/env python3
"""Developer probe (GPU box): timeline of one k_se_chunk launch -- tiles, the hand-off, the chunk roles.
Build: bash tools/dev/build_variant.sh tailst -DTAIL_STAMPS -mllvm -disable-machine-licm;  python tools/dev/tail_timeline.py tailst [uk380|syn2048]"""
import ctypes, os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from covid19uk_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dev", "variants", f"libseirhip_{sys.argv[1]}.so")
from covid19uk_amd import synth
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel
B = 8
cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)
wl = sys.argv[2] if len(sys.argv) > 2 else "uk380"
cov = synth.make_covariates(wl)
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
u = synth.jitter_params(u0, B, scale=0.002, seed=7, T=cov.T)
ev = np.stack([events] * B)
lib = _lib.load()
lib.seir_debug_tail_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
with SeirModel(cov, init, max_chains=B) as model:
    with ChainSampler(model, cfg, B, seed=1, trace_capacity=50, record_events=False, num_leapfrog_steps=3) as s:
        s.set_state(u, ev); s.set_kernel(step_size=1.2e-5 if wl == "uk380" else 2e-6)
        s.run(20); model.sync()
        out = np.zeros((B, 8), dtype=np.uint64)
        rows = []
        for rep in range(5):
            lib.seir_debug_tail_stamps(s._s, out.ctypes.data, 1)
            s.run(1); model.sync()
            lib.seir_debug_tail_stamps(s._s, out.ctypes.data, 0)
            st = out.astype(np.int64)
            t0 = st[:, 0]
            rel = lambda k: (st[:, k] - t0) * 10
            rows.append(np.stack([rel(5), rel(1), rel(2), rel(3), rel(4)]))
        r = np.median(np.stack(rows), axis=0)
        print("ns after the chain's first tile started (median of 5 launches; min / median / max over the 8 chains)")
        for name, v in zip(["first role workgroup starts", "last tile has counted in", "first role past its wait",
                            "last role past its wait", "last role done (stores acknowledged)"], r):
            print(f"  {name:40s} {v.min():7.0f} {np.median(v):7.0f} {v.max():7.0f}")
