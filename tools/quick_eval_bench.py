#!/usr/bin/env python3
"""Developer timing of the full log-prob evaluation and its kernels (GPU box)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="uk380")
    ap.add_argument("--chains", type=int, default=8)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--form", default="fused", choices=["fused", "three-launch", "four-launch"])
    ap.add_argument("--gemm-f32", type=int, default=0, help="1: the fp32 MFMA mobility contraction (BASELINE config 5)")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as entry
    entry.build()
    from covid19uk_amd import synth
    from covid19uk_amd.seir import SeirModel

    cov = synth.make_covariates(args.workload)
    events, init, truth = synth.simulate_epidemic(cov)
    u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
    B = args.chains
    u = synth.jitter_params(u0, B, T=cov.T)
    dev = torch.device("cuda:0")
    ut = torch.tensor(u, device=dev)
    evt = torch.tensor(np.stack([events] * B), device=dev)
    lp = torch.empty(B, dtype=torch.float64, device=dev)
    g = torch.empty(B, u.shape[1], dtype=torch.float64, device=dev)
    out = {"workload": args.workload, "B": B, "M": cov.M, "T": cov.T}
    with SeirModel(cov, init, max_chains=B) as model:
        model.set_option(eval_form=args.form)
        if args.gemm_f32:
            model.set_option(gemm_f32=True)
        out["form"] = args.form
        out["gemm_f32"] = bool(args.gemm_f32)
        for grad in (None, g):
            for _ in range(3):
                model.log_prob_dev(ut, evt, lp, grad)
            model.sync()
            model.timer_start()
            for _ in range(args.iters):
                model.log_prob_dev(ut, evt, lp, grad)
            ms = model.timer_stop() / args.iters
            out["full_eval_%s_ms" % ("grad" if grad is not None else "value")] = ms
        model.prepare_events_dev(evt)
        for grad in (None, g):
            for _ in range(3):
                model.eval_prepared_dev(ut, lp, grad)
            model.sync()
            model.timer_start()
            for _ in range(args.iters):
                model.eval_prepared_dev(ut, lp, grad)
            ms = model.timer_stop() / args.iters
            out["prepared_eval_%s_ms" % ("grad" if grad is not None else "value")] = ms
        model.log_prob_dev(ut, evt, lp, g)
        model.sync()
        for name in ("scan", "gemm", "se_value", "se_grad", "finish", "state", "tiles_value", "tiles_grad", "finish_fused"):
            out["k_%s_us" % name] = 1e3 * model.time_kernel(name, B, args.iters)
    cells = B * cov.M * cov.T
    out["alg_bytes_per_eval"] = 24 * cells + 8 * cov.M * cov.M
    out["gemm_flop"] = 2.0 * B * cov.M * cov.M * cov.T
    out["gemm_tflops"] = out["gemm_flop"] / (out["k_gemm_us"] * 1e-6) / 1e12
    out["full_value_GBps_alg"] = out["alg_bytes_per_eval"] / (out["full_eval_value_ms"] * 1e-3) / 1e9
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
