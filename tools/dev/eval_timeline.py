#!/usr/bin/env python3
"""Developer probe (GPU box): where the one-launch stateless evaluation (k_eval_all) spends its time -- stamps of tile 0 of
chain 0 and of the chain's finish block.
Build: bash tools/dev/build_variant.sh evst -DEVAL_STAMPS=1 -mllvm -disable-machine-licm;  python tools/dev/eval_timeline.py evst [grad]"""
import ctypes, os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from covid19uk_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dev", "variants", f"libseirhip_{sys.argv[1]}.so")
import torch
from covid19uk_amd import synth
from covid19uk_amd.seir import SeirModel
with_grad = len(sys.argv) > 2
B = 8
cov = synth.make_covariates("uk380")
events, init, truth = synth.simulate_epidemic(cov)
u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
u = synth.jitter_params(u0, B, T=cov.T)
dev = torch.device("cuda:0")
ut = torch.tensor(u, device=dev); evt = torch.tensor(np.stack([events] * B), device=dev)
lp = torch.empty(B, dtype=torch.float64, device=dev); g = torch.empty(B, u.shape[1], dtype=torch.float64, device=dev)
lib = _lib.load()
lib.seir_debug_eval_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
with SeirModel(cov, init, max_chains=B) as model:
    for _ in range(5):
        model.log_prob_dev(ut, evt, lp, g if with_grad else None)
    model.sync()
    rows = []
    for rep in range(15):
        model.log_prob_dev(ut, evt, lp, g if with_grad else None)
        model.sync()
        out = np.zeros(8, dtype=np.uint64)
        lib.seir_debug_eval_stamps(model._ctx, out.ctypes.data)
        rows.append(out.astype(np.int64) * 10.0)
    st = np.stack(rows)
    rel = st - st[:, :1]
    med = np.median(rel, axis=0)
    names = ["tile 0 enters", "its share of the state scan done", "chain's state complete (wait over)", "tile done (contraction + S->E epilogue)",
             "finish block enters", "finish: all counted in", "finish done"]
    for k, n in enumerate(names):
        print(f"{n:45s} {med[k]:8.0f} ns")
    model.timer_start()
    for _ in range(50):
        model.log_prob_dev(ut, evt, lp, g if with_grad else None)
    print("per batch: %.1f us" % (model.timer_stop() / 50 * 1e3))
