#!/usr/bin/env python3
"""Developer probe (GPU box): per-step timeline of the event-update launches from the stamps of chain 0's workgroups
(roles 1, 2, 0 in slots 0, 1, 2; band workgroups from slot 3) -- plain stores of s_memrealtime.
Build: bash tools/dev/build_variant.sh pairst -DPAIR_STAMPS=1 -mllvm -disable-machine-licm
Run:   python tools/dev/pair_timeline.py pairst [paired|paired-launch]"""
import ctypes, os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from covid19uk_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dev", "variants", f"libseirhip_{sys.argv[1]}.so")
from covid19uk_amd import synth
from covid19uk_amd.sampler import ChainSampler
from covid19uk_amd.seir import SeirModel
moves = sys.argv[2] if len(sys.argv) > 2 else "paired"
B = 8
cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)
cov = synth.make_covariates("uk380")
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
u = synth.jitter_params(u0, B, scale=0.002, seed=7, T=cov.T)
ev = np.stack([events] * B)
lib = _lib.load()
lib.seir_debug_leap_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
NR = int(sys.argv[3]) if len(sys.argv) > 3 else 3          # role slots of the launch (the early-draw experiment of round 4 had 4)
NS = NR + 24
with SeirModel(cov, init, max_chains=B) as model:
    with ChainSampler(model, cfg, B, seed=1, trace_capacity=50, record_events=False, moves=moves) as s:
        s.set_state(u, ev); s.set_kernel(step_size=1.2e-5)
        s.run(20); model.sync()
        out = np.zeros((B + 2) * 128 + 4096, dtype=np.uint64)
        reps = []
        for rep in range(9):
            s.run(1); model.sync()
            lib.seir_debug_leap_stamps(s._s, out.ctypes.data, 0)
            reps.append(out[:NS * 12 * 16].reshape(NS, 12, 16).astype(np.int64).copy())
        st = np.stack(reps) * 10.0                             # [rep, slot, step, stamp] ns (100 MHz clock)
        R0 = NR - 1                                            # role 0 sits in the last role slot
        t0 = st[:, R0:R0 + 1, :, 0:1]                          # role 0's entry of the step
        rel = st - t0
        med = lambda a: np.median(a, axis=0)
        names = {0: "role 1", 1: "role 2", R0: "role 0", NR: "band 0", NR + 11: "band 11", NR + 23: "band 23"}
        if NR == 4:
            names[2] = "role 3"
        print(f"moves={moves}; ns after role 0 entered the step (median of 9 sweeps); stamps: 0 entry, roles: 6 entry loads issued, 12 back, 13 past the barrier, 14 uniforms drawn, 15 pending descriptors in LDS; 1..5 inside, 8 step done, 10 drained, 11 flag seen, 9 L1 dropped")
        for step in (0, 1, 4, 5, 8, 9):
            print(f"-- step {step}")
            for slot, name in names.items():
                row = med(rel[:, slot, step, :])
                print(f"   {name:8s} " + " ".join(f"{k}:{row[k]:.0f}" for k in (0, 4, 5, 7, 6, 12, 13, 14, 15, 1, 2, 3, 8, 10, 11, 9)))
        per = med(st[:, R0, 1:10, 0] - st[:, R0, 0:9, 0])
        print("step period (role 0 entry to entry):", per.round().tolist(), "mean %.0f ns" % per.mean())
        last = st[:, :, :10, 8].max(axis=1) - st[:, R0, :10, 0]
        who = st[:, :, :10, 8].argmax(axis=1)
        print("last workgroup done (ns after role 0's entry):", med(last).round().tolist())
        print("which slot is last (last sweep):", who[-1].tolist())
