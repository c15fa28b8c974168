#!/usr/bin/env python3
"""Developer probe (GPU box): when every tile workgroup of the gradient kernel k_se starts and ends within a launch.
Build: bash tools/dev/build_variant.sh sest -DSE_STAMPS;  python tools/dev/se_timeline.py sest [chains]"""
import ctypes, os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from covid19uk_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dev", "variants", f"libseirhip_{sys.argv[1]}.so")
from covid19uk_amd import synth
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)
cov = synth.make_covariates("uk380")
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
u = synth.jitter_params(u0, B, scale=0.002, seed=7, T=cov.T)
ev = np.stack([events] * B)
lib = _lib.load()
lib.seir_debug_read_ts.argtypes = [ctypes.c_void_p, _lib.c_double_p, ctypes.c_int64]
with SeirModel(cov, init, max_chains=B) as model:
    with ChainSampler(model, cfg, B, seed=1, trace_capacity=10, record_events=False) as s:
        s.set_state(u, ev); s.set_kernel(step_size=1.2e-5)
        s.run(2); model.sync()
        us = 1e3 * s.time_grad_kernel(50)
        ntile = 6 * 24
        ts = np.empty(B * ntile * 4)
        lib.seir_debug_read_ts(model._ctx, ts.ctypes.data_as(_lib.c_double_p), ts.size)
        raw = ts.view(np.uint64).reshape(B, ntile, 4)
        t0 = (raw[..., 2] & np.uint64(0xffffffffffff)).astype(np.int64)
        t1 = (raw[..., 3] & np.uint64(0xffffffffffff)).astype(np.int64)
        xcc = (raw[..., 3] >> np.uint64(60)).astype(np.int64)
        cu = ((raw[..., 3] >> np.uint64(48)) & np.uint64(0xfff)).astype(np.int64)
        base = t0.min()
        st, en = (t0 - base) * 10, (t1 - base) * 10
        print(f"kernel {us:.2f} us by events; {B} chains x {ntile} tiles")
        print("start ns: min/median/p90/max", st.min(), int(np.median(st)), int(np.percentile(st, 90)), st.max())
        print("end   ns: min/median/p90/max", en.min(), int(np.median(en)), int(np.percentile(en, 90)), en.max())
        dur = en - st
        print("tile duration ns: min/median/p90/max", dur.min(), int(np.median(dur)), int(np.percentile(dur, 90)), dur.max())
        for b in range(min(B, 2)):
            order = np.argsort(st[b])
            print(f"chain {b}: xcc {sorted(set(xcc[b].tolist()))}, distinct CU ids {len(set(cu[b].tolist()))}; starts by order:",
                  st[b][order][::12].tolist(), "ends:", en[b][order][::12].tolist())
        # workgroups per (xcc, cu)
        key = xcc * 4096 + cu
        uniq, cnt = np.unique(key, return_counts=True)
        print("workgroups per CU: ", dict(zip(*np.unique(cnt, return_counts=True))))
