#!/usr/bin/env python3
"""Developer timing of the sampler sweep (GPU box): ms/sweep for a few chain-group settings."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="uk380")
    ap.add_argument("--chains", type=int, default=8)
    ap.add_argument("--sweeps", type=int, default=100)
    ap.add_argument("--groups", default="1,2,4,8")
    ap.add_argument("--hmc", default="chunk")
    ap.add_argument("--moves", default="paired")
    ap.add_argument("--graph", action="store_true", help="replay the sweep as a hipGraph")
    ap.add_argument("--variant", default="", help="tools/dev/variants/libseirhip_<variant>.so instead of the product library")
    args = ap.parse_args()
    if args.variant:                                    # (BEFORE anything loads the product library: build() does)
        from covid19uk_amd import _lib
        _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev", "variants", f"libseirhip_{args.variant}.so")
    else:
        import __graft_entry__ as entry
        entry.build()
    from covid19uk_amd import synth
    from covid19uk_amd.sampler import ChainSampler
    from covid19uk_amd.seir import SeirModel
    cfg = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)
    cov = synth.make_covariates(args.workload)
    events, init, truth = synth.simulate_epidemic(cov)
    u0 = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
    B = args.chains
    u = synth.jitter_params(u0, B, scale=0.002, seed=7, T=cov.T)
    ev = np.stack([events] * B)
    out = {}
    for g in [int(x) for x in args.groups.split(",")]:
        if g > B:
            continue
        with SeirModel(cov, init, max_chains=B) as model:
            with ChainSampler(model, cfg, B, seed=1, trace_capacity=args.sweeps, chain_groups=g, hmc=args.hmc, moves=args.moves, use_graph=args.graph) as s:
                s.set_state(u, ev)
                s.set_kernel(step_size=1.2e-5)
                s.reset_trace(); s.run(10); model.sync()
                s.reset_trace()
                model.timer_start()
                s.run(args.sweeps)
                ms = model.timer_stop()
                tr = s.read_trace(args.sweeps, events=False)
                out[f"groups={g}"] = dict(ms_per_sweep=ms / args.sweeps, samples_per_s=B * args.sweeps / (ms * 1e-3),
                                          hmc_acc=float(tr.hmc["is_accepted"].mean()),
                                          grad_kernel_us=1e3 * s.time_grad_kernel(100))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
