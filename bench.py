#!/usr/bin/env python3
"""Headline benchmark: posterior samples/sec of the spatial SEIR Metropolis-within-Gibbs
sampler on the 380-LAD x 365-day UK workload (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

One "step" = one sweep = one posterior draw of every chain on this GPU: an HMC
update of the P=750 parameters (16 leapfrogs = 17 gradient evaluations) plus
num_event_time_updates x [S->E move, E->I move, S->E occult, E->I occult]
Metropolis-Hastings updates of the event tensor (example_config.yaml:26-30), with
every draw (parameters, event tensor [M,T,3], kernel results) recorded to the
device-side burst buffer as the reference's sample_chain does
(covid19uk/inference/inference.py:232-240,453-468).  Chains shard over GPUs with
no data-path collective (weak scaling: --chains-per-gpu chains on every GPU).

The timed region starts with the chain state resident in HBM; the burst
buffer's device->host copy and HDF5 write happen between bursts and are
reported separately (`pcie_inclusive_samples_per_sec`), never as `value`.

Rank 0 prints ONE JSON line (see README "Benchmark").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MCMC_CONFIG = dict(dmax=84, nmax=25, m=2, occult_nmax=15, num_event_time_updates=5)  # example_config.yaml:26-30
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md, chip-level parameters
F64_MFMA_PEAK_TFLOPS = 78.6     # dense fp64 matrix peak (v_mfma_f64_16x16x4_f64), same guide
F32_MFMA_PEAK_TFLOPS = 157.3    # dense fp32 matrix peak (v_mfma_f32_32x32x2_f32), same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="uk380", choices=["uk380", "ni11", "syn2048"],
                    help="uk380 = the configuration BASELINE.json's metric is quoted on (default, the headline); syn2048 = "
                         "BASELINE config 5 (2048 regions x 730 days, fp32 MFMA mobility contraction)")
    ap.add_argument("--gemm-f32", type=int, default=None, choices=[0, 1],
                    help="mobility contraction F = Cstar.X in fp32 MFMA (default: 1 for syn2048 -- BASELINE config 5 names "
                         "it -- and 0 otherwise: the fp64 MFMA form)")
    ap.add_argument("--chains-per-gpu", type=int, default=8)
    ap.add_argument("--adapt-sweeps", type=int, default=60, help="untimed dual-averaging sweeps during setup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-egress", action="store_true",
                    help="skip the overlapped-egress measurement (kernel profiles: concurrent copies stretch kernel durations)")
    ap.add_argument("--no-cli", action="store_true", help="skip the end-to-end (HDF5-writing) single-chain measurement")
    ap.add_argument("--no-chains-scaling", action="store_true",
                    help="skip the extra (untimed-by-the-contract) runs at 32 and 64 chains on this GPU")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (BASELINE.json configs 2, 3 and 5 beside the headline: NI-11 x 16 chains, "
                         "UK-380 x 1 chain, SYN-2048 x 8 chains with the fp32 MFMA contraction; ~30 s, most of it generating "
                         "the 2048 x 730 synthetic epidemic on the host)")
    ap.add_argument("--cpu-baseline-sweeps", type=int, default=0, help="0 = size for ~10 s per leg")
    ap.add_argument("--spinup-seconds", type=float, default=0.7,
                    help="untimed sweeps run back to back before the timed region so that a short --steps run is "
                         "measured at steady clocks (a fresh box ramps for a few hundred ms)")
    ap.add_argument("--seed", type=int, default=20210101)
    return ap.parse_args()


def cpu_baseline(cov, init, events, u0, n_sweeps, seed, cores=None):
    """The oracle sampler (full re-evaluation of the joint log-prob for every proposal, as the reference does) on the host
    cores: oracle/mcmc_oracle.c on top of oracle/seir_oracle.c -- the sweep in plain C, the density's loops under OpenMP,
    nothing but C in the timed loop."""
    from oracle import c_binding, seir_oracle as so
    k = so.make_constants(cov.C, cov.N, cov.W, cov.weekday, cov.area, cov.adjacency, init)
    # the box's CPU share for one GPU is 16 cores; more OpenMP threads than that only add overhead
    if cores is None:
        cores = min(len(os.sched_getaffinity(0)), 16)
    c_binding.set_threads(cores)
    ch = c_binding.COracleChain(k, MCMC_CONFIG, u0, events, seed=seed, chain_id=0)
    ch.eps = 2e-5
    t0 = time.perf_counter()
    ch.run(1)
    one = time.perf_counter() - t0
    n = n_sweeps if n_sweeps > 0 else max(2, min(40, int(10.0 / max(one, 1e-3))))
    e0 = ch.n_evals
    t0 = time.perf_counter()
    ch.run(n)
    dt = time.perf_counter() - t0
    n_evals = ch.n_evals - e0
    # the second half of BASELINE's metric on the same cores: full log-prob evaluations by the C restatement alone
    evals = {}
    for key, grad in (("value", False), ("value_and_grad", True)):
        c_binding.evaluate(k, u0, events, 1, want_grad=grad)
        t0 = time.perf_counter()
        m = 0
        while time.perf_counter() - t0 < 1.5:
            c_binding.evaluate(k, u0, events, 1, want_grad=grad)
            m += 1
        evals[key] = m / (time.perf_counter() - t0)
    return {"value": n / dt, "unit": "posterior samples/sec", "cores": cores, "kind": "port",
            "log_prob_evals_per_sec": evals,
            "sample": f"{n} sweeps of 1 chain, oracle/mcmc_oracle.c + oracle/seir_oracle.c (plain C, OpenMP over the density's "
                      f"loops, {cores} threads; no Python in the timed loop), {n_evals} full log-prob evaluations, same workload"}


def alg_bytes_per_eval(M, T, P, B):
    """algorithmic bytes of one gradient evaluation of B chains (fp64 events + vectors per chain, Cstar once)."""
    return B * (24 * M * T + 8 * (4 * M + 3 * T + P) + 8 + 8 * P) + 8 * M * M


def leap_section_name(launches, evals):
    """What ran between the events of seir_sampler_time_leapfrog, from its launch and evaluation counts (include/seir_hip.h)."""
    if launches == 1:
        return "k_leap (persistent: the whole HMC trajectory)"
    if launches == 2 and evals > 2:
        return "k_leap x 2 (persistent: the whole HMC trajectory, two launches of 8 chains one after the other)"
    if launches == evals + 1:
        return "k_se_chunk x (L+1) + k_hmc_final (one launch per gradient evaluation: tiles + chunk roles; the accept test by the roles' own launch)"
    if launches == evals:
        return "k_se_chunk (one launch per inner leapfrog step: gradient tiles + chunk roles)"
    return "k_se + k_hmc_chunk (two launches per inner leapfrog step)"


def aux_config(label, workload, B, local, seed, steps, warm, eps, f32=False):
    """One of BASELINE.json's other configurations on this GPU, beside the headline: `steps` timed sweeps of B chains after
    `warm` untimed ones (HIP events on the context stream, inputs resident, draws recorded to the burst buffer), the sweep's
    dominant kernel -- the leapfrog section, timed in place -- priced like the headline's, and for the fp32 configuration the
    mobility contraction against the fp32 MFMA peak.  Fixed step size, no adaptation: a throughput figure."""
    import torch
    from covid19uk_amd import synth
    from covid19uk_amd.sampler import ChainSampler
    from covid19uk_amd.seir import SeirModel
    t_start = time.perf_counter()
    cov = synth.make_covariates(workload, seed)
    events, init, truth = synth.simulate_epidemic(cov, seed)
    u = synth.jitter_params(synth.unconstrain(synth.pack_params(truth, cov.M, cov.T)), B, scale=0.002, seed=7, T=cov.T)
    ev = np.stack([events] * B)
    M, T = cov.M, cov.T
    out = {"config": label, "workload": f"{workload}: M={M} x T={T}", "chains_per_gpu": B, "steps": steps, "warmup": warm}
    with SeirModel(cov, init, max_chains=B, device=local) as model:
        P = model.P
        if f32:
            model.set_option(gemm_f32=True)
        with ChainSampler(model, MCMC_CONFIG, B, seed=seed, trace_capacity=steps, record_events="u16") as s:
            s.set_state(u, ev)
            s.set_kernel(step_size=eps)
            s.run(warm)
            model.sync()
            s.reset_trace()
            model.timer_start()
            s.run(steps)
            ms = model.timer_stop()
            tr = s.read_trace(steps, events=False)
            lm, ll, le = s.time_leapfrog(min(20, steps))
            alg = alg_bytes_per_eval(M, T, P, B)
            out.update({
                "value": B * steps / (ms * 1e-3), "unit": "posterior samples/sec", "ms_per_step": ms / steps,
                "dtype": "f64 (mobility contraction: fp32 MFMA)" if f32 else "f64",
                "hmc_acceptance": float(tr.hmc["is_accepted"].mean()),
                "all_log_probs_finite": bool(np.isfinite(tr.hmc["target_log_prob"]).all()),
                "launch_form": list(s.launch_form()), "recoveries": len(s.recoveries),
                "dominant_kernel": {
                    "kernel": leap_section_name(ll, le),
                    "launches_per_sweep": ll, "gradient_evaluations": le, "section_us": 1e3 * lm,
                    "share_of_sweep": lm / (ms / steps), "bound": "hbm",
                    "achieved": le * alg / (lm * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": le * alg / (lm * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "algorithmic_bytes_per_evaluation": alg}})
        if f32:
            # BASELINE config 5 names the contraction F = Cstar . X (model_spec.py:258-263) on the fp32 matrix cores: once
            # per burst in the sampler (the event updates keep F current by rank-1 bands), every call in the stateless path
            dev = torch.device("cuda", local)
            ut, evt = torch.tensor(u, device=dev), torch.tensor(ev, device=dev)
            lp = torch.empty(B, dtype=torch.float64, device=dev)
            gr = torch.empty(B, P, dtype=torch.float64, device=dev)
            model.log_prob_dev(ut, evt, lp, gr)
            model.sync()
            ms_g = model.time_kernel("gemm", B, 10)
            flops = 2.0 * M * M * T * B
            out["contraction"] = {"kernel": "k_gemm_f32 (v_mfma_f32_32x32x2_f32, 128 x 128 tiles)", "bound": "mfma",
                                  "mean_launch_us": 1e3 * ms_g, "achieved": flops / (ms_g * 1e-3) / 1e12,
                                  "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": flops / (ms_g * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS}
    out["wall_s_including_setup"] = time.perf_counter() - t_start
    return out


def self_launch(n):
    """Run this script as n ranks on this node; returns the job's exit code."""
    import socket
    import subprocess
    with socket.socket() as so_:
        so_.bind(("127.0.0.1", 0))
        port = so_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as a fresh child job (one process per GPU under
        # torch.distributed.run, RCCL over xGMI) BEFORE anything here touches the GPU, relay its output (rank 0
        # prints the JSON line) and leave with its exit code.  Never an exec of a process that has used the GPU.
        sys.exit(self_launch(a.gpus))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    # one process per GPU; RCCL ("nccl") over xGMI.  BENCH_DIST_BACKEND=gloo + BENCH_SHARE_GPU=1
    # rehearses the N>1 code path with several ranks on ONE GPU (collectives on CPU tensors).
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU") == "1":
        local = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        dist.barrier()
    from covid19uk_amd import distributed as D
    from covid19uk_amd import synth
    from covid19uk_amd.sampler import ChainSampler
    from covid19uk_amd.seir import SeirModel

    B = a.chains_per_gpu
    first_chain, _ = D.shard_chains(world * B, world, rank)
    cov = synth.make_covariates(a.workload, a.seed)
    events, init, truth = synth.simulate_epidemic(cov, a.seed)
    u_true = synth.unconstrain(synth.pack_params(truth, cov.M, cov.T))
    # every chain of the job gets its own start point (global chain id = rank*B + b)
    u_all = synth.jitter_params(u_true, world * B, scale=0.002, seed=7, T=cov.T)
    u0 = u_all[first_chain:first_chain + B].copy()
    ev0 = np.stack([events] * B)
    K, W = a.steps, a.warmup

    syn = a.workload == "syn2048"
    f32 = bool(a.gemm_f32) if a.gemm_f32 is not None else syn
    if syn:
        # the extras below are UK-380 context for the headline number; at SYN-2048 they would only cost time
        a.no_egress = a.no_chains_scaling = a.no_cli = True
    model = SeirModel(cov, init, max_chains=B, device=local)
    if f32:
        model.set_option(gemm_f32=True)
    burst = max(1, min(K, 50))                    # sweeps per burst of the overlapped-egress measurement
    # (several ranks on ONE GPU -- the CPU-collective rehearsal of the N > 1 path, BENCH_SHARE_GPU=1 -- cannot each have a
    # whole-chip persistent launch resident: there the leapfrog steps and the pairs of event updates are one launch each)
    sampler = ChainSampler(model, MCMC_CONFIG, B, seed=a.seed, first_chain_id=first_chain,
                           trace_capacity=max(K, 2 * burst), record_events="u16",
                           hmc="chunk-launch" if os.environ.get("BENCH_SHARE_GPU") == "1" and world > 1 else "chunk",
                           moves="paired-launch" if os.environ.get("BENCH_SHARE_GPU") == "1" and world > 1 else "paired")
    sampler.set_state(u0, ev0)
    sampler.set_kernel(step_size=2e-6 if syn else 2e-5)
    # setup (untimed): a short dual-averaging window, then pool the step size over ALL chains
    # of the job (the optional cross-chain gather of BASELINE.json config 4: RCCL all_gather).
    if a.adapt_sweeps > 0:
        sampler.set_adaptation(adapt_step_size=True, num_adaptation_steps=a.adapt_sweeps)
        sampler.reset_trace()
        sampler.run(a.adapt_sweeps)
        model.sync()
    eps, _ = sampler.get_kernel()
    pooled = D.pool_step_sizes(eps, device=local)       # all_gather over RCCL when world > 1
    sampler.set_adaptation(adapt_step_size=False)
    sampler.set_kernel(step_size=pooled)

    # warm-up steps, then the timed region
    sampler.reset_trace()
    sampler.run(W)
    model.sync()
    # spin-up (untimed): keep the GPU busy until it holds its steady clock; with the default
    # --steps the timed region is only ~0.1 s and would otherwise sit on the clock ramp
    spin_sweeps, t_spin = 0, time.perf_counter()
    while time.perf_counter() - t_spin < a.spinup_seconds:
        # bursts of 100 sweeps whatever K is (sweeps beyond the trace capacity are simply not recorded): with K-sweep
        # bursts a short --steps run would spin up in 10 ms pieces with a host round trip between them, and the
        # timed region measured 7 % slower than the steady state
        sampler.reset_trace()
        sampler.run(100)
        model.sync()
        spin_sweeps += 100
    sampler.reset_trace()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    model.sync()
    t0 = time.perf_counter()
    model.timer_start()
    sampler.run(K)
    ev_ms = model.timer_stop()              # blocks until the K sweeps are done (HIP events, ctx stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, device=local)

    # draws leave the device between bursts: D2H of the whole burst buffer (PCIe-inclusive rate)
    t1 = time.perf_counter()
    tr = sampler.read_trace(min(K, sampler.cap))
    d2h = time.perf_counter() - t1
    # context for `value` when --steps is small: the same sweeps as one uninterrupted run of 400.  Per-sweep time
    # falls with the length of the uninterrupted run (after a device synchronisation: 0.62 ms for one sweep, 0.59 at 8,
    # 0.57 at 16-32, 0.55 at 64, 0.54 from ~200 on -- tools/dev/burst_shape.py): the clock climbs over tens of
    # milliseconds of queued work and any idle gap, however short, restarts the climb.  Not part of `value`.
    steady = None
    if K < 200:
        sampler.reset_trace()
        torch.cuda.synchronize()
        model.sync()
        model.timer_start()
        sampler.run(400)
        ms400 = model.timer_stop()
        steady = {"sweeps": 400, "ms_per_step": ms400 / 400, "samples_per_sec": world * B * 400 / (ms400 * 1e-3),
                  "note": "one uninterrupted run of 400 sweeps after the timed region (this rank's HIP events); "
                          "`value` is the --steps sweeps the contract asks for"}
    acc = {"hmc": float(tr.hmc["is_accepted"].mean())}
    for key, mv in tr.moves.items():
        acc[key] = float(mv["is_accepted"].mean())
    finite = bool(np.isfinite(tr.hmc["target_log_prob"]).all())

    # the same sweeps with every draw leaving the device: bursts of `burst` sweeps, each crossing PCIe into
    # page-locked memory on a copy stream while the next one runs (ChainSampler.sample_bursts; what the CLI does)
    n_bursts = max(16, K // burst)
    touched = []
    overlapped = None
    if not a.no_egress:
        sampler.sample_bursts(2, burst, lambda tr_, i: None)          # untimed: page-locks the two host buffers
        t2 = time.perf_counter()
        sampler.sample_bursts(n_bursts, burst, lambda tr_, i: touched.append(int(tr_.events[-1, 0, 0, 0, 0])))
        overlapped = time.perf_counter() - t2

    # dominant kernel of the sweep: the gradient evaluation, 17 per sweep.  Where it fits the chip ONE persistent launch
    # (k_leap) performs all 17 (the whole trajectory, its end included); otherwise the 15 inner leapfrog steps are one
    # k_se_chunk launch each and the end points plain k_se launches.  Timed in place: HIP events around that section of
    # ordinary sweeps, on the stream the kernels run on; k_se also stand-alone.
    leap_ms, leap_launches, leap_evals = sampler.time_leapfrog(min(200, max(20, K)))
    grad_ms = sampler.time_grad_kernel(200)
    # The persistent launch also closes the trajectory (last half kick, accept test, adaptation, trace: k_hmc_step<2>'s
    # work in the other forms), which is not gradient work: the same section with that part as a launch of its own
    # (hmc="chunk-stage") is timed beside it, so that the gradient evaluations' own figure stays comparable over rounds.
    leap_noend_ms = None
    if leap_launches == 1 and world == 1:
        with SeirModel(cov, init, max_chains=B, device=local) as mx_:      # (a context of its own: one sampler per context)
            with ChainSampler(mx_, MCMC_CONFIG, B, seed=a.seed, first_chain_id=first_chain, trace_capacity=8,
                              record_events=False, hmc="chunk-stage") as sx_:
                sx_.set_state(*sampler.get_state()[:2])
                sx_.set_kernel(*sampler.get_kernel())
                sx_.run(10)
                mx_.sync()
                lm_, ll_, le_ = sx_.time_leapfrog(100)
                if ll_ == 1 and le_ == leap_evals:
                    leap_noend_ms = lm_
    M, T, P = cov.M, cov.T, model.P
    alg_bytes = alg_bytes_per_eval(M, T, P, B)
    achieved = leap_evals * alg_bytes / (leap_ms * 1e-3) / 1e9
    achieved_k_se = alg_bytes / (grad_ms * 1e-3) / 1e9
    # what this kernel itself moves: int32 k_se, S, I + fp64 F per padded cell (20 B), the per-row /
    # per-day tables, and its partial sums out (row sums per day chunk, column sums per row tile, 4 tile scalars)
    Mp, Tp = -(-M // 64) * 64, -(-T // 64) * 64
    nmt, ntc = Mp // 16, Tp // 64
    kernel_bytes = B * (20 * Mp * Tp + 8 * (Mp + 2 * Tp) + 8 * (ntc * Mp + nmt * Tp) + 8 * 6 * nmt * ntc)
    # HBM-side traffic per launch from the committed PMC passes (rocprofv3 cannot run inside this
    # process): FETCH_SIZE x the gfx950 calibration factor + WRITE_SIZE, same workload and batch.
    traffic, traffic_src, traffic_k_se = None, None, None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if a.workload == "uk380" and B == 8 and pmcs:
        doc = json.load(open(pmcs[-1]))
        ent = doc.get("k_se<GRAD=true,SRC=planes>")
        if ent:
            traffic_k_se = ent["traffic_bytes_per_launch"]
        ent = doc.get("k_leap" if leap_launches == 1 else "k_se_chunk")
        if ent:
            traffic = ent["traffic_bytes_per_launch"]
            traffic_src = "profiles/" + os.path.basename(pmcs[-1])

    # secondary metric: full log_prob evaluations/sec through the stateless C-ABI
    dev = torch.device("cuda", local)
    ut = torch.tensor(u0, device=dev)
    evt = torch.tensor(ev0, device=dev)
    lp = torch.empty(B, dtype=torch.float64, device=dev)
    gr = torch.empty(B, P, dtype=torch.float64, device=dev)
    evals = {}
    for name, g in (("value", None), ("value_and_grad", gr)):
        for _ in range(3):
            model.log_prob_dev(ut, evt, lp, g)
        model.sync()
        model.timer_start()
        for _ in range(50):
            model.log_prob_dev(ut, evt, lp, g)
        evals[name] = B * 50 / (model.timer_stop() * 1e-3)
    # the same with the event-dependent part prepared once (what the 17 evaluations of an HMC
    # draw share: state scan, binomial coefficients, E->I term and the mobility contraction)
    model.prepare_events_dev(evt)
    for name, g in (("value_events_prepared", None), ("value_and_grad_events_prepared", gr)):
        for _ in range(3):
            model.eval_prepared_dev(ut, lp, g)
        model.sync()
        model.timer_start()
        for _ in range(50):
            model.eval_prepared_dev(ut, lp, g)
        evals[name] = B * 50 / (model.timer_stop() * 1e-3)

    # independent evaluations issued on several contexts (one stream each) overlap: the HBM-bound state pass of one
    # runs under the matrix-core tiles of another.  Same call, same batch, three contexts used in turn.
    try:
        if syn:
            raise RuntimeError("skipped at SYN-2048")
        extra = [SeirModel(cov, init, max_chains=B, device=local) for _ in range(2)]
        ctxs = [model] + extra
        lps = [lp] + [torch.empty_like(lp) for _ in extra]
        grs = [gr] + [torch.empty_like(gr) for _ in extra]
        for name, use_g in (("value_3_contexts", False), ("value_and_grad_3_contexts", True)):
            for i in range(9):
                ctxs[i % 3].log_prob_dev(ut, evt, lps[i % 3], grs[i % 3] if use_g else None)
            for m_ in ctxs:
                m_.sync()
            t0c = time.perf_counter()
            for i in range(150):
                ctxs[i % 3].log_prob_dev(ut, evt, lps[i % 3], grs[i % 3] if use_g else None)
            for m_ in ctxs:
                m_.sync()
            evals[name] = B * 150 / (time.perf_counter() - t0c)
        for m_ in extra:
            m_.close()
    except Exception as e:                                  # an extra, never the reason a bench line is missing
        evals["three_contexts_error"] = str(e)

    # BASELINE's "log_prob evals/sec" names no batch size: the same call with more chains per batch on ONE context (its own
    # context per size; the one-launch form serves 8 and 16 chains, the three-launch form the larger batches, where the
    # wide kernels fill the chip by themselves)
    try:
        if syn:
            raise RuntimeError("skipped at SYN-2048")
        by_batch = {}
        for Bx in (16, 32, 64):
            with SeirModel(cov, init, max_chains=Bx, device=local) as mb_:
                ub_ = torch.tensor(synth.jitter_params(u_true, Bx, scale=0.002, seed=7, T=cov.T), device=dev)
                eb_ = torch.tensor(np.stack([events] * Bx), device=dev)
                lb_ = torch.empty(Bx, dtype=torch.float64, device=dev)
                gb_ = torch.empty(Bx, P, dtype=torch.float64, device=dev)
                ent_ = {}
                for name, g in (("value", None), ("value_and_grad", gb_)):
                    for _ in range(3):
                        mb_.log_prob_dev(ub_, eb_, lb_, g)
                    mb_.sync()
                    mb_.timer_start()
                    for _ in range(30):
                        mb_.log_prob_dev(ub_, eb_, lb_, g)
                    ms_b = mb_.timer_stop()
                    ent_[name] = Bx * 30 / (ms_b * 1e-3)
                    ent_[name + "_us_per_batch"] = 1e3 * ms_b / 30
                by_batch[str(Bx)] = ent_
                del ub_, eb_, lb_, gb_
        evals["by_chains_per_batch_one_context"] = by_batch
    except Exception as e:                                  # an extra, never the reason a bench line is missing
        evals["by_batch_error"] = str(e)

    # the stateless evaluation (what the reference calls 37x per draw) against ITS bound: the larger of the
    # HBM time of the algorithmic bytes and the fp64 matrix time of the mobility contraction 2 M^2 T per chain
    t_hbm = alg_bytes / (HBM_PEAK_GBPS * 1e9)
    t_mfma = 2.0 * M * M * T * B / ((F32_MFMA_PEAK_TFLOPS if f32 else F64_MFMA_PEAK_TFLOPS) * 1e12)
    t_eval = B / evals["value_and_grad"]
    stateless = {"path": "seir_log_prob_dev, value + gradient, 8 chains: k_eval_all [parameter tables | tiles: their share of the state "
                         "scan, then the fp64 MFMA contraction with the S->E term as epilogue | row constants | I->R fold] with XCD-local "
                         "hand-offs inside the launch, then k_finish (inside the same launch for value-only calls); other batch "
                         "sizes: k_state_params, k_eval_tiles, k_finish",
                 "bound": "mfma" if t_mfma > t_hbm else "hbm",
                 "t_hbm_us": 1e6 * t_hbm, "t_mfma_us": 1e6 * t_mfma, "measured_us_per_batch": 1e6 * t_eval,
                 "frac": max(t_hbm, t_mfma) / t_eval,
                 "frac_3_contexts": (max(t_hbm, t_mfma) * evals["value_and_grad_3_contexts"] / B)
                                    if "value_and_grad_3_contexts" in evals else None,
                 "achieved": (2.0 * M * M * T * B / t_eval / 1e12) if t_mfma > t_hbm else alg_bytes / t_eval / 1e9,
                 "peak": (F32_MFMA_PEAK_TFLOPS if f32 else F64_MFMA_PEAK_TFLOPS) if t_mfma > t_hbm else HBM_PEAK_GBPS,
                 "unit": "TFLOP/s" if t_mfma > t_hbm else "GB/s"}
    try:
        stateless["three_launch_form_kernels_us"] = {n: 1e3 * model.time_kernel(n, B, 50) for n in ("state", "tiles_grad", "finish_fused")}
        stateless["four_launch_form_kernels_us"] = {n: 1e3 * model.time_kernel(n, B, 50) for n in ("scan", "gemm", "se_grad", "finish")}
    except Exception as e:                                  # timing hook only
        stateless["kernels_us"] = str(e)

    # BASELINE config 5: the mobility contraction F = Cstar . X (model_spec.py:258-263) on the matrix cores, both
    # operand types, timed stand-alone on the context stream (HIP events) against the dense MFMA peak of the type
    contraction = None
    if syn:
        contraction = {}
        flops = 2.0 * M * M * T * B
        for name, on, peak in (("f32", True, F32_MFMA_PEAK_TFLOPS), ("f64", False, F64_MFMA_PEAK_TFLOPS)):
            model.set_option(gemm_f32=on)
            model.log_prob_dev(ut, evt, lp, gr)            # the scan writes the operand of this type
            model.sync()
            ms_g = model.time_kernel("gemm", B, 20)
            contraction[name] = {"kernel": "k_gemm_f32 (v_mfma_f32_32x32x2_f32, 128 x 128 tiles)" if on
                                           else "k_gemm<64> (v_mfma_f64_16x16x4_f64, 64 x 64 tiles)",
                                 "bound": "mfma", "mean_launch_us": 1e3 * ms_g, "algorithmic_flops_per_launch": flops,
                                 "achieved": flops / (ms_g * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                                 "frac": flops / (ms_g * 1e-3) / 1e12 / peak, "traffic": None}
            model.log_prob_dev(ut, evt, lp, gr)
            model.sync()
            model.timer_start()
            for _ in range(20):
                model.log_prob_dev(ut, evt, lp, gr)
            contraction[name]["log_prob_grad_evals_per_sec"] = B * 20 / (model.timer_stop() * 1e-3)
        model.set_option(gemm_f32=f32)

    # context for the headline number (N=1 only, not part of `value`): the same sweep with more
    # chains resident on this GPU -- at 8 chains the sweep is bound by its dependent-launch chain,
    # so throughput keeps growing until the gradient kernel saturates HBM
    scaling = None
    if world == 1 and not a.no_chains_scaling and a.workload == "uk380":
        sampler.close()
        model.close()
        scaling = {}
        for Bx in (16, 32, 64):                           # (a single chain on one GPU -- BASELINE config 3 -- is in `configs`)
            ux = synth.jitter_params(u_true, Bx, scale=0.002, seed=7, T=cov.T)
            mx = SeirModel(cov, init, max_chains=Bx, device=local)
            sx = ChainSampler(mx, MCMC_CONFIG, Bx, seed=a.seed, first_chain_id=0, trace_capacity=40,
                              record_events="u16")
            sx.set_state(ux, np.stack([events] * Bx))
            sx.set_kernel(step_size=pooled)
            sx.run(10)
            mx.sync()
            sx.reset_trace()
            mx.timer_start()
            sx.run(40)
            msx = mx.timer_stop()
            gx = sx.time_grad_kernel(100)
            lmx, llx, lex = sx.time_leapfrog(20)
            bytes_x = Bx * (24 * M * T + 8 * (4 * M + 3 * T + P) + 8 + 8 * P) + 8 * M * M
            scaling[str(Bx)] = {"samples_per_sec": Bx * 40 / (msx * 1e-3), "ms_per_step": msx / 40,
                                "grad_kernel_us": 1e3 * gx,
                                "grad_kernel_frac_of_hbm_peak": bytes_x / (gx * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                "leapfrog_section_us": 1e3 * lmx, "leapfrog_launches": llx,
                                "leapfrog_frac_of_hbm_peak": lex * bytes_x / (lmx * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            sx.close()
            mx.close()

    # the drop-in surface end to end (N=1 only): the sampling phase of run_mcmc for ONE chain, as the reference runs
    # it, every draw written to posterior.hd5 in the reference's schema (samples/seir float64 [n,M,T,3]: 3.3 MB per
    # draw at UK-380) -- warm-up excluded, device->host->HDF5 included
    cli = None
    if world == 1 and not a.no_cli and a.workload == "uk380":
        import tempfile
        from covid19uk_amd.inference import inference as inf
        m1 = SeirModel(cov, init, max_chains=1, device=local)
        s1 = ChainSampler(m1, MCMC_CONFIG, 1, seed=a.seed, trace_capacity=100, record_events="u16")
        s1.set_state(u_all[:1], ev0[:1])
        s1.set_kernel(step_size=pooled)
        nbc, nsc = 6, 50
        with tempfile.TemporaryDirectory() as tmp:
            post = inf.Posterior(os.path.join(tmp, "posterior.hd5"), M, T, MCMC_CONFIG["m"], nbc * nsc, burst=nsc)
            off = [0]

            def flush(tr_, i):
                post.write_samples(inf.draws_to_dict(tr_.theta, tr_.events, 0), first_dim_offset=off[0])
                post.write_results(inf.trace_to_dict(tr_, 0), first_dim_offset=off[0])
                off[0] += tr_.theta.shape[0]
            s1.sample_bursts(2, nsc, lambda tr_, i: None)          # untimed: page-locks the host buffers
            t3 = time.perf_counter()
            s1.sample_bursts(nbc, nsc, flush)
            dt_cli = time.perf_counter() - t3
            post.close()
            cli = {"value": nbc * nsc / dt_cli, "unit": "posterior samples/sec", "chains": 1,
                   "format": "hdf5" if post.use_h5 else "npz (no libhdf5 on this host)",
                   "bytes_written": nbc * nsc * (8 * (3 * M * T + P) + 200),
                   "note": f"{nbc} bursts x {nsc} draws through ChainSampler.sample_bursts + Posterior (inference.py:453-468)"}
        s1.close()
        m1.close()

    # BASELINE.json's other configurations on the driver's line (N=1, default workload only): each a short run of its own
    # sampler after the headline's has been closed.  Stated budget: ~30 s in all, two thirds of it the host-side
    # simulation of the 2048 x 730 synthetic epidemic.
    configs = None
    if world == 1 and rank == 0 and a.workload == "uk380" and not a.no_configs:
        if scaling is None:
            sampler.close()
            model.close()
            scaling = {}                                   # (closed: nothing below touches them)
        configs, t_cfg = [], time.perf_counter()
        for label, wl, Bc, st_, wm_, eps_, f32_ in (
                ("BASELINE config 2: 11-LAD NI, 16 chains on one GPU", "ni11", 16, 200, 50, 0.002, False),
                ("BASELINE config 3: 380-LAD UK x 365 days, single chain", "uk380", 1, 100, 30, float(np.mean(pooled)), False),
                ("BASELINE config 5: synthetic 2048 x 730, 8 chains per GPU, fp32 MFMA mobility contraction", "syn2048", 8, 12, 3,
                 2e-6, True)):
            try:
                configs.append(aux_config(label, wl, Bc, local, a.seed, st_, wm_, eps_, f32_))
            except Exception as e:                          # an extra, never the reason a bench line is missing
                configs.append({"config": label, "error": repr(e)})
        configs.append({"budget_note": "stated budget: 30 s for the three runs together (most of it the host-side simulation of the 2048 x 730 "
                                       "epidemic; 2 - 15 s observed)", "wall_s": time.perf_counter() - t_cfg})

    if rank == 0:
        out = {
            "metric": {"uk380": "posterior samples/sec, 380-LAD UK SEIR", "ni11": "posterior samples/sec, 11-LAD NI SEIR",
                       "syn2048": "posterior samples/sec, synthetic 2048-region x 730-day SEIR"}[a.workload],
            "value": world * B * K / elapsed,
            "unit": "posterior samples/sec",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (mobility contraction: fp32 MFMA)" if f32 else "f64", "data": "synthetic",
            "config": {"workload": f"{a.workload}: M={M} LADs x T={T} days, P={P} parameters",
                       "chains_per_gpu": B, "chains_total": world * B,
                       "sweep": "HMC(16 leapfrogs) + 5 x [S->E move, E->I move, S->E occult, E->I occult]",
                       "mcmc": MCMC_CONFIG, "draws_recorded": "theta + events[M,T,3] (uint16 counts) + kernel results per sweep",
                       "parallelism": f"chains sharded over {world} GPU(s), no data-path collective"},
            "roofline": {"kernel": (f"k_leap<TSM,NTC,NST,RW> (persistent: {leap_evals} of the sweep's 17 gradient evaluations in one launch -- the "
                                    "whole HMC trajectory: per evaluation the S->E term's gradient sums of all chains "
                                    "from register-resident cells (and its value at the two end points), then the chunk roles' leapfrog "
                                    "update; at the end the roles' last half kick, accept test, adaptation and trace)")
                                   if leap_launches == 1 else
                                   f"{leap_section_name(leap_launches, leap_evals)}; {leap_evals} of the sweep's 17 gradient evaluations",
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         # `frac` above is the contract's figure: ALGORITHMIC bytes / duration / peak.  It is not a utilisation
                         # for the persistent launch (cells are read once per trajectory and kept in registers), so the two
                         # readings are also given under names that say what they are:
                         "frac_algorithmic_bytes": achieved / HBM_PEAK_GBPS,
                         "frac_measured_traffic": (traffic / (leap_ms * 1e-3 / leap_launches) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "limited_by": ("the chunk roles' serial part of a leapfrog step (one wave per role: gather, sums, exponentials) and the tiles' fp64 vector "
                                        "issue, joined by two hand-offs through the XCD's L2 (self-validating words): latency, not bandwidth"
                                        if leap_launches == 1 else "HBM / fabric streaming of the planes, then role latency"),
                         "frac_uses": "algorithmic bytes of SURVEY.md 8d (fp64 events + vectors per chain, Cstar once) x the gradient "
                                      "evaluations the timed section performs / its duration (HIP events around it in ordinary sweeps)",
                         "algorithmic_bytes_per_evaluation": alg_bytes, "evaluations_per_section": leap_evals,
                         "launches_per_section": leap_launches, "section_us": 1e3 * leap_ms,
                         "us_per_evaluation": 1e3 * leap_ms / leap_evals,
                         "mean_launch_us": 1e3 * leap_ms / leap_launches,
                         "without_the_trajectory_end": None if leap_noend_ms is None else {
                             "what": "the same launch with the trajectory's end as k_hmc_step<2>'s own launch (hmc='chunk-stage'): "
                                     "the 17 gradient evaluations and 16 leapfrog steps alone",
                             "section_us": 1e3 * leap_noend_ms, "us_per_evaluation": 1e3 * leap_noend_ms / leap_evals,
                             "frac": leap_evals * alg_bytes / (leap_noend_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                         "note": "the persistent kernel reads its cells from HBM once per trajectory and keeps them in registers, so its "
                                 "memory traffic is far below the algorithmic bytes: what bounds it is the fp64 vector issue rate of the "
                                 "tile phase (~70 instructions per cell) and the two in-L2 hand-offs per step -- `frac` prices it "
                                 "against the HBM time of the algorithmic bytes as SURVEY.md 8d defines",
                         "k_se_stand_alone": {"kernel": "k_se<GRAD=true,SRC=planes> (the streaming form of one evaluation: what the multi-launch "
                                                        "forms run at a trajectory's end points)",
                                              "mean_launch_us": 1e3 * grad_ms, "achieved": achieved_k_se,
                                              "frac": achieved_k_se / HBM_PEAK_GBPS,
                                              "kernel_bytes_per_launch": kernel_bytes,
                                              "frac_kernel_bytes": kernel_bytes / (grad_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                              "frac_traffic": (traffic_k_se / (grad_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic_k_se else None,
                                              "traffic": traffic_k_se},
                         "frac_time_weighted_17": (17 * alg_bytes / ((leap_ms + max(0, 17 - leap_evals) * grad_ms) * 1e-3) / 1e9) / HBM_PEAK_GBPS},
            "roofline_stateless": stateless,
            "spinup_sweeps": spin_sweeps,
            "steady_state": steady,
            "log_prob_evals_per_sec": evals,
            "hip_event_ms_per_step": ev_ms / K,
            "pcie_inclusive_samples_per_sec": (world * B * n_bursts * burst / overlapped) if overlapped else None,
            "pcie_inclusive_note": f"{n_bursts} bursts of {burst} sweeps, draws (theta, events uint16, kernel results) copied to "
                                   "page-locked host memory on a copy stream while the next burst runs",
            "pcie_serial_samples_per_sec": world * B * K / (elapsed + d2h),
            "acceptance": acc, "step_size": pooled, "all_log_probs_finite": finite,
        }
        if cli:
            out["cli_samples_per_sec"] = cli
        if scaling:
            out["chains_per_gpu_scaling"] = scaling
        if configs:
            out["configs"] = configs
        if contraction:
            # at SYN-2048 the line's `roofline` is the matrix-core contraction BASELINE config 5 names; the sweep's own
            # dominant kernel (HBM-bound, as at UK-380) stays alongside
            out["roofline_sweep_kernel"] = out["roofline"]
            out["roofline"] = contraction["f32" if f32 else "f64"]
            out["contraction"] = contraction
        if not a.no_cpu_baseline and world == 1:              # (the contract: rank 0, N = 1 only)
            out["cpu_baseline"] = cpu_baseline(cov, init, events, u0[0], a.cpu_baseline_sweeps or (2 if syn else 0), a.seed)
            if not syn:       # one core would need minutes per sweep at this size
                out["cpu_baseline_1core"] = cpu_baseline(cov, init, events, u0[0], a.cpu_baseline_sweeps, a.seed, cores=1)
        print(json.dumps(out), flush=True)
    if scaling is None:
        sampler.close()
        model.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
