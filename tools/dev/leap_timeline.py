#!/usr/bin/env python3
"""Developer probe (GPU box): timeline of the persistent leapfrog launch k_leap from the stamps of two tiles (0 and 77) and
two roles (T-chunk 0, M-chunk 0) of chain 0 -- plain stores of s_memrealtime, which do not disturb the hand-offs they time.
Build: bash tools/dev/build_variant.sh leapst -DLEAP_STAMPS=1 -mllvm -disable-machine-licm;  python tools/dev/leap_timeline.py leapst"""
import ctypes, os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from covid19uk_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dev", "variants", f"libseirhip_{sys.argv[1]}.so")
from covid19uk_amd import synth
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ROWS = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # tile shape of k_leap: 0 auto, 24, 32
cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)
cov = synth.make_covariates("uk380")
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
u = synth.jitter_params(u0, B, scale=0.002, seed=7, T=cov.T)
ev = np.stack([events] * B)
lib = _lib.load()
lib.seir_debug_leap_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
with SeirModel(cov, init, max_chains=B) as model:
    with ChainSampler(model, cfg, B, seed=1, trace_capacity=50, record_events=False, leap_rows=ROWS, log=None) as s:
        s.set_state(u, ev); s.set_kernel(step_size=1.2e-5)
        s.run(20); model.sync()
        out = np.zeros((B + 2) * 128 + 4096, dtype=np.uint64)
        reps = []
        for rep in range(9):
            lib.seir_debug_leap_stamps(s._s, out.ctypes.data, 0)
            s.run(1); model.sync()
            lib.seir_debug_leap_stamps(s._s, out.ctypes.data, 0)
            reps.append(out[:(B + 2) * 128].reshape(B + 2, 16, 8).astype(np.int64).copy())
            allt = out[(B + 2) * 128:].reshape(1024, 4).copy()
        st = np.stack(reps) * 10.0                             # ns (100 MHz clock)
        tiles = st[:, B:, :15, :7]                             # [rep, tile 0 / 77, step, stamp]
        roles = st[:, 1:3, :15, :5]
        inner = st[:, 1, 1:14, 5:8] - st[:, B, 1:14, :1]        # T-chunk 0: loads back, wave sums, scans (ns after tile 0 passed the wait)
        print("T-chunk 0 inside: loads back %.0f, three wave sums %.0f, two scans %.0f" % tuple(np.median(inner[..., k]) for k in range(3)))
        more = st[:, 3, 1:14, :5] - st[:, B, 1:14, :1]
        print("T-chunk 0 inside: column sums in %.0f; before the wait: entry loads issued %.0f, back + two wave sums %.0f, I->R part done %.0f" % (
            np.median(more[..., 0]), np.median(more[..., 2]), np.median(more[..., 3]), np.median(more[..., 4])))                            # [rep, T-chunk 0 / M-chunk 0, step, stamp]
        mi = st[:, 2, 1:14, 5:8] - st[:, B, 1:14, :1]           # M-chunk 0: entry loads issued, (Q s) formed, at the wait
        mm = st[:, 4, 1:14, :4] - st[:, B, 1:14, :1]            # ... partial sums in, -, three wave sums, softplus done
        print("M-chunk 0 inside: entry loads issued %.0f, (Q s) formed %.0f, at the wait %.0f | partial sums in %.0f, wave sums %.0f, softplus %.0f" % (
            np.median(mi[..., 0]), np.median(mi[..., 1]), np.median(mi[..., 2]), np.median(mm[..., 0]), np.median(mm[..., 2]), np.median(mm[..., 3])))
        period = np.median(tiles[:, 0, 2:15, 0] - tiles[:, 0, 1:14, 0])
        print(f"step period (tile 0 past its wait, step to step): median {period:.0f} ns; whole launch ~{period * 15 / 1e3:.1f} us")
        first = np.median(tiles[:, 0, 1, 0] - tiles[:, 0, 0, 0])
        d0 = tiles[:, 0, 0, :] - tiles[:, 0, 0, :1]
        print(f"step 0 (operands from memory, cold code): {first:.0f} ns until tile 0 passes the wait of step 1; inside step 0: " +
              ", ".join(f"{np.median(d0[:, k]):.0f}" for k in range(7)))
        r0 = roles[:, :, 0, :] - tiles[:, 0, 0, :1][:, None, :]
        print("roles in step 0 (T-chunk 0, M-chunk 0), ns after tile 0's first stamp:", np.median(r0, axis=0).round().tolist())
        pn = ["past the wait", "tables in LDS", "cells done", "past middle barrier", "reductions issued", "stores acknowledged", "counted in"]
        for pi, name in enumerate(("tile 0", "tile 77")):
            d = tiles[:, pi, 1:14, :] - tiles[:, pi, 1:14, :1]
            print(name, "ns after it passed the wait: " + ", ".join(f"{n} {np.median(d[..., k]):.0f}" for k, n in enumerate(pn)))
        rn = ["previous roles done", "tiles in", "stores issued", "stores acknowledged", "counted in"]
        for ri, name in enumerate(("T-chunk 0", "M-chunk 0")):
            d = roles[:, ri, 1:14, :] - tiles[:, 0, 1:14, :1]
            print(name, "ns after tile 0 passed the wait: " + ", ".join(f"{n} {np.median(d[..., k]):.0f}" for k, n in enumerate(rn)))
        nxt = tiles[:, 0, 2:15, 0] - roles[:, :, 1:14, 4].max(axis=1)
        print(f"tile 0 passes the next wait {np.median(nxt):.0f} ns after the later of the two roles counted in")
        # every tile workgroup of chain 0 in step 7 of the last sweep
        nwg = int((allt[:, 0] != 0).sum())
        a = allt[:nwg]
        t0 = a[:, 0].astype(np.int64).min()
        hw = a[:, 3] & 0xffffffff
        cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4)          # CU id | SE id  (HW_ID: wave 0-3, simd 4-5, cu 8-11, sh 12, se 13-15)
        rel = (a[:, :3].astype(np.int64) - t0) * 10
        order = np.argsort(rel[:, 2])
        print(f"{nwg} tile workgroups of chain 0, step 7: past the wait {np.percentile(rel[:,0],[0,50,100])}, cells done {np.percentile(rel[:,1],[0,50,100])}, counted in {np.percentile(rel[:,2],[0,50,90,100])}")
        import collections
        per = collections.Counter(cu.tolist())
        print("workgroups per CU:", sorted(collections.Counter(per.values()).items()))
        print("last eight to count in: (tix, cu, past wait, cells done, counted in, workgroups on that cu)")
        for i in order[-8:]:
            print("  ", i, int(cu[i]), rel[i].tolist(), per[int(cu[i])])
        print("first four:")
        for i in order[:4]:
            print("  ", i, int(cu[i]), rel[i].tolist(), per[int(cu[i])])
