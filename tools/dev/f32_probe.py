#!/usr/bin/env python3
"""Developer probe (GPU box): accuracy and speed of the fp32-MFMA contraction at SYN-2048 x 730."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as entry
entry.build()
from covid19uk_amd.seir import SeirModel
from oracle import c_binding
from tests import helpers as H

case = H.build_case("syn2048", 14)
c_binding.set_threads(16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng = np.random.default_rng(14)
u = np.tile(case["u"], (B, 1))
u[:, :6] += 0.01 * rng.normal(size=(B, 6))
ev = np.stack([case["events"]] * B)
want = [c_binding.evaluate(case["k"], u[b], ev[b], 1, want_grad=True) for b in range(min(B, 2))]
import torch
with SeirModel(case["cov"], case["init"], max_chains=B) as model:
    for f32 in (0, 1):
        model.set_option(gemm_f32=f32)
        lp, g = model.log_prob_grad(u, ev)
        for b in range(min(B, 2)):
            wl, wg = want[b]
            scale = np.maximum(np.abs(wg), 1e-6 * np.abs(wg).max())
            print("f32" if f32 else "f64", "chain", b, "logp rel err", abs(lp[b] - wl) / abs(wl), "grad max rel", np.max(np.abs(g[b] - wg) / scale),
                  "grad err / max|g|", np.max(np.abs(g[b] - wg)) / np.abs(wg).max())
        print("  k_gemm us", 1e3 * model.time_kernel("gemm", B, 10), "k_scan us", 1e3 * model.time_kernel("scan", B, 10))
