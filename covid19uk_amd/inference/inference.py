"""MCMC driver for the COVID-19 UK spatial SEIR model on MI355X.

Host-side mirror of the reference's `covid19uk/inference/inference.py`: the same
CLI (`python -m covid19uk_amd.inference.inference -c config.yaml -o posterior.hd5
data.nc`; `python -m covid19uk.inference.inference` forwards here), the same
`config["Mcmc"]` keys, the same windowed warm-up schedule and the same
`posterior.hd5` layout.  Everything numerical -- the joint log-probability, HMC,
the event-time and occult Metropolis-Hastings kernels -- runs in libseirhip's HIP
kernels through `covid19uk_amd.sampler.ChainSampler`; this file only sequences
windows and moves draws from the device burst buffer to disk.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

from .. import hdf5io
from .. import model_spec
from ..sampler import MOVE_KEYS, ChainSampler
from ..seir import SeirModel
from .mcmc_kernel_factory import event_kernel_config, hmc_kernel_kwargs_default

DTYPE = model_spec.DTYPE


# ---------------------------------------------------------------------------
# inference data (the reference's NetCDF4 `constant_data` / `observations` groups)
# ---------------------------------------------------------------------------
def read_inference_data(path):
    """(Covariates, cases [M,T], dates [T] of str) from the file `assemble_data` writes
    (covid19uk/data/assemble.py:8-16; variables of model_spec.py:88-105).  `.npz` files with
    the same variable names are accepted too."""
    if str(path).endswith(".npz"):
        d = np.load(path, allow_pickle=False)
        cov = model_spec.Covariates(C=d["C"], W=d["W"], N=d["N"], adjacency=d["adjacency"],
                                    weekday=d["weekday"], area=d["area"])
        dates = [str(x) for x in d["time"]] if "time" in d.files else [str(i) for i in range(cov.T)]
        return cov, np.asarray(d["cases"], DTYPE), dates
    with hdf5io.File(path, "r") as f:
        g = {k: f.read(f"/constant_data/{k}") for k in ("C", "W", "N", "adjacency", "weekday", "area")}
        cases = np.asarray(f.read("/observations/cases"), DTYPE)
        dates = None
        if f.exists("/observations/time"):
            t = f.read("/observations/time")
            if t.dtype.kind == "S":
                dates = [x.decode() for x in t]
            else:
                units = f.read_str_attr("/observations/time", "units") or ""
                if units.startswith("days since "):
                    t0 = np.datetime64(units[len("days since "):].split()[0])
                    dates = [str(t0 + np.timedelta64(int(x), "D")) for x in t]
                else:
                    dates = [str(int(x)) for x in t]
    cov = model_spec.Covariates(**{k: np.asarray(v, DTYPE) for k, v in g.items()})
    if cases.shape == (cov.T, cov.M) and cov.T != cov.M:
        cases = cases.T
    if dates is None:
        dates = [str(i) for i in range(cases.shape[1])]
    return cov, cases, dates


def write_inference_data(path, cov: model_spec.Covariates, cases, dates=None):
    """Write an input file with the layout `read_inference_data` expects (tests, synthetic runs)."""
    cases = np.asarray(cases, DTYPE)
    dates = [str(i) for i in range(cases.shape[1])] if dates is None else list(dates)
    if str(path).endswith(".npz"):
        np.savez(path, C=cov.C, W=cov.W, N=cov.N, adjacency=cov.adjacency, weekday=cov.weekday, area=cov.area,
                 cases=cases, time=np.array(dates))
        return
    with hdf5io.File(path, "w") as f:
        for k in ("C", "W", "N", "adjacency", "weekday", "area"):
            a = np.asarray(getattr(cov, k), DTYPE)
            f.create_dataset(f"/constant_data/{k}", a.shape, np.float64)
            f.write(f"/constant_data/{k}", a)
        f.create_dataset("/observations/cases", cases.shape, np.float64)
        f.write("/observations/cases", cases)
        n = max(len(s) for s in dates)
        f.create_dataset("/observations/time", (len(dates),), f"S{n}")
        f.write("/observations/time", np.array(dates, dtype=f"S{n}"))


def read_location_names(path):
    """The `location` coordinate of the input file (LAD codes; covid19uk/data/assemble.py writes it with the
    `constant_data` group), or None when the file carries none."""
    if str(path).endswith(".npz"):
        return None
    with hdf5io.File(path, "r") as f:
        for name in ("/constant_data/location", "/observations/location"):
            if f.exists(name):
                v = f.read(name)
                if v.dtype.kind in "SO":
                    return [x.decode() if isinstance(x, bytes) else str(x) for x in v.reshape(-1)]
                return [str(int(x)) for x in v.reshape(-1)]
    return None


# ---------------------------------------------------------------------------
# posterior.hd5 (gemlib `Posterior`: samples/<key>, results/<nested/keys>)
# ---------------------------------------------------------------------------
class Posterior:
    """HDF5 sink with the dataset layout of inference.py:285-300 / :245-282 / :588-592.
    `is_accepted` is a bool dataset the way h5py stores one (gemlib's Posterior writes numpy bools through h5py):
    the int8 enum {FALSE = 0, TRUE = 1}."""

    def __init__(self, filename, M, T, mmax, num_samples, burst=100):
        self.filename = filename
        self.use_h5 = not str(filename).endswith(".npz") and hdf5io.available()
        self.shapes = {
            "samples/psi": (), "samples/sigma_space": (), "samples/beta_area": (), "samples/gamma0": (),
            "samples/gamma1": (), "samples/alpha_0": (), "samples/alpha_t": (T - 1,),
            "samples/spatial_effect": (M,), "samples/seir": (M, T, 3),
            "results/hmc/is_accepted": (), "results/hmc/target_log_prob": (), "results/hmc/step_size": (),
        }
        self.dtypes = {"results/hmc/is_accepted": np.bool_}
        for key in MOVE_KEYS:
            self.shapes[f"results/{key}/is_accepted"] = ()
            self.shapes[f"results/{key}/target_log_prob"] = ()
            self.shapes[f"results/{key}/proposed_delta"] = (4, mmax)
            self.dtypes[f"results/{key}/is_accepted"] = np.bool_
        self.num_samples = int(num_samples)
        self._scratch = {}
        if self.use_h5:
            self._file = hdf5io.File(filename, "w")
            for name, shp in self.shapes.items():
                self._file.create_dataset("/" + name, (self.num_samples,) + shp, self.dtypes.get(name, np.float64),
                                          raw=name == "samples/seir")
        else:
            self._file = None
            self._mem = {name: np.zeros((self.num_samples,) + shp, self.dtypes.get(name, np.float64))
                         for name, shp in self.shapes.items()}
            self._extra = {}

    def write(self, name, value, first_dim_offset):
        if self.use_h5:
            v = np.asarray(value)
            if name == "samples/seir":
                # the event tensor arrives as the device's integer counts (a strided view of the burst): converted to
                # the file's float64 and written at its file address by a few threads (hdf5io.write_rows_parallel)
                self._file.write_rows_parallel("/" + name, v, offset=first_dim_offset)
                return
            if v.dtype != np.float64 and v.dtype.kind in "iu" and v.nbytes > (1 << 20):
                # the event tensor arrives as the device's integer counts (a strided view of the burst); it is
                # converted to the file's float64 in ONE pass into a buffer kept between bursts -- a fresh
                # 165 MB array per burst costs more in page faults than the conversion itself
                buf = self._scratch.get(v.shape)
                if buf is None:
                    buf = self._scratch[v.shape] = np.empty(v.shape, np.float64)
                np.copyto(buf, v, casting="unsafe")
                v = buf
            self._file.write("/" + name, v, offset=first_dim_offset)
        else:
            v = np.asarray(value)
            self._mem[name][first_dim_offset:first_dim_offset + v.shape[0]] = v

    def write_samples(self, samples: dict, first_dim_offset):
        for k, v in samples.items():
            self.write(f"samples/{k}", v, first_dim_offset)

    def write_results(self, results: dict, first_dim_offset):
        for k, v in results.items():
            for kk, vv in v.items():
                self.write(f"results/{k}/{kk}", vv, first_dim_offset)

    def create_dataset(self, name, data):
        data = np.asarray(data)
        if self.use_h5:
            self._file.create_dataset("/" + name, data.shape, data.dtype if data.dtype.kind == "S" else np.float64)
            self._file.write("/" + name, data)
        else:
            self._extra[name] = data

    def __getitem__(self, name):
        if self.use_h5:
            self._file.flush()
            return self._file.read("/" + name)
        return self._mem[name]

    def close(self):
        if self.use_h5:
            self._file.close()
        else:
            out = {k.replace("/", "__"): v for k, v in {**self._mem, **self._extra}.items()}
            np.savez(self.filename, **out)


def get_weighted_running_variance(u_draws):
    """inference.py:36-47: mean/variance of the second half of a window's (unconstrained)
    draws, with pseudo-count n/2.  u_draws [n,B,P] -> (count [B], mean [B,P], var [B,P])."""
    n = u_draws.shape[0]
    half = u_draws[(-n) // 2:]      # as the reference's `draws[-draws.shape[0] // 2:]`: floor of the NEGATIVE -> 13 of 25
    mean, var = half.mean(axis=0), half.var(axis=0)
    return np.full(u_draws.shape[1], n / 2.0), mean, np.maximum(var, 1e-300)


def unconstrain_theta(theta):
    u = np.array(theta, dtype=DTYPE, copy=True)
    y = u[..., :2] - np.finfo(DTYPE).eps
    u[..., :2] = y + np.log(-np.expm1(-y))
    return u


def draws_to_dict(theta, events, chain):
    """inference.py:285-300 for one chain: theta [n,B,P] constrained, events [n,B,M,T,3]."""
    M, T = events.shape[2], events.shape[3]
    th = theta[:, chain]
    return {
        "psi": th[:, 0], "sigma_space": th[:, 1], "beta_area": th[:, 2], "gamma0": th[:, 3],
        "gamma1": th[:, 4], "alpha_0": th[:, 5], "alpha_t": th[:, 6:6 + T - 1],
        "spatial_effect": th[:, 6 + T - 1:6 + T - 1 + M],
        "seir": events[:, chain] if events.dtype.kind in "iu" else events[:, chain].astype(DTYPE),   # Posterior.write converts
    }


def trace_to_dict(tr, chain):
    """trace_results_fn (inference.py:245-282) for one chain."""
    out = {"hmc": {k: v[:, chain] for k, v in tr.hmc.items()}}
    for key in MOVE_KEYS:
        out[key] = {k: v[:, chain] for k, v in tr.moves[key].items()}
    return out


def run_mcmc(sampler: ChainSampler, config, posteriors, log=sys.stderr, pool_step_size=False):
    """The windowed schedule of inference.py:303-470: fast 200, slow 25*2^k (k<6), fast 50,
    then num_bursts x num_burst_samples with the kernel fixed.  Every draw -- warm-up
    included -- is written, as in the reference."""
    first_window_size, last_window_size, slow_window_size, num_slow_windows = 200, 50, 25, 6
    dual_averaging_kwargs = {"target_accept_prob": 0.75}
    offset = 0

    def flush(tr):
        nonlocal offset
        n = tr.theta.shape[0]
        for c, post in enumerate(posteriors):
            post.write_samples(draws_to_dict(tr.theta, tr.events, c), first_dim_offset=offset)
            post.write_results(trace_to_dict(tr, c), first_dim_offset=offset)
        offset += n

    def window(n, adapt_mass, running_variance=None):
        sampler.set_adaptation(adapt_step_size=True, adapt_mass=adapt_mass, num_adaptation_steps=n,
                               running_variance=running_variance, **dual_averaging_kwargs)
        tr = sampler.sample(n)
        flush(tr)
        return tr, get_weighted_running_variance(unconstrain_theta(tr.theta))

    print(f"Fast window {first_window_size}", file=log, flush=True)
    sampler.set_kernel(step_size=hmc_kernel_kwargs_default()["step_size"])
    tr, running_variance = window(first_window_size, False)
    for k in range(num_slow_windows):
        n = slow_window_size * 2 ** k
        print(f"Slow window {n}", file=log, flush=True)
        tr, running_variance = window(n, True, running_variance)
    print(f"Fast window {last_window_size}", file=log, flush=True)
    tr, _ = window(last_window_size, False)

    print("Sampling...", file=log, flush=True)
    step_size = tr.hmc["step_size"][(-last_window_size) // 2:].mean(axis=0)      # inference.py:439-441
    if pool_step_size:
        # build extension (the reference is single-chain): every chain of the job -- all ranks -- samples with
        # the geometric mean of the adapted step sizes; one float64 per chain over RCCL / gloo
        from .. import distributed as D
        step_size = np.full(sampler.B, D.pool_step_sizes(step_size, device=sampler.model.device))
        print(f"Pooled step size over all chains: {step_size[0]:.4g}", file=log, flush=True)
    sampler.set_adaptation(adapt_step_size=False)
    sampler.set_kernel(step_size=step_size, variance=sampler.get_kernel()[1])
    nb, ns = int(config["num_bursts"]), int(config["num_burst_samples"])
    t0 = time.perf_counter()
    if nb and ns and sampler.cap >= 2 * ns:
        # bursts overlap: while burst k+1 runs, burst k crosses PCIe into page-locked memory and is written
        # to the HDF5 file on a worker thread (ChainSampler.sample_bursts)
        def on_burst(tr, i):
            flush(tr)
            print(f"  burst {i + 1}/{nb}", file=log, flush=True)
        sampler.sample_bursts(nb, ns, on_burst)
    else:
        for i in range(nb):
            flush(sampler.sample(ns))
            print(f"  burst {i + 1}/{nb}", file=log, flush=True)
    dt = time.perf_counter() - t0
    if nb * ns:
        print(f"Sampling: {nb * ns * sampler.B / dt:.1f} posterior samples/s "
              f"({sampler.B} chain(s), device->host->disk included)", file=log, flush=True)
    return offset


def warmup_size():
    return 200 + 25 * (2 ** 6 - 1) + 50          # inference.py:312-322


def job_layout(num_chains, device=None, env=os.environ):
    """Where this process sits in a multi-GPU job: one process per GPU (torchrun / torch.distributed.run
    sets RANK, WORLD_SIZE, LOCAL_RANK), `num_chains` chains on each.  Read from the environment only --
    nothing here touches the GPU.  Returns dict(rank, world, device, first_chain_id)."""
    rank, world = int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1"))
    if not (0 <= rank < world):
        raise ValueError(f"RANK={rank} outside WORLD_SIZE={world}")
    dev = int(env.get("LOCAL_RANK", "0")) if device is None else int(device)
    return dict(rank=rank, world=world, device=dev, first_chain_id=rank * int(num_chains))


def chain_file_name(output_file, chain, total_chains):
    """posterior.hd5 for a single-chain job (the reference); posterior_chain{c}.hd5 with the GLOBAL chain id otherwise."""
    if total_chains == 1:
        return output_file
    root, ext = os.path.splitext(output_file)
    return f"{root}_chain{chain}{ext}"


def dispersed_start(P, chain_ids, scale, seed):
    """Starting points u0[B,P]: the reference's zeros (inference.py:563-573) for global chain 0 and, when
    scale > 0, N(0, scale^2) perturbations for the others, keyed by the GLOBAL chain id so that a chain's
    start does not depend on how the job is sharded (over-dispersed starts for R-hat)."""
    u0 = np.zeros((len(chain_ids), P))
    if scale > 0:
        for b, c in enumerate(chain_ids):
            if c > 0:
                u0[b] = np.random.default_rng([int(seed), 0x5EED, int(c)]).normal(0.0, scale, size=P)
    return u0


def trace_events_dtype(cases, choice="auto"):
    """Width of the event counts in the device-side burst buffer: "u16" halves the buffer and the bytes that cross PCIe
    per draw, but a count above 65535 cannot be held (the read then fails loudly, after the burst has run).  "auto" decides
    from the data before anything runs: the latent S->E / E->I counts of a LAD-day are of the order of its observed
    removals, so 16-bit counts are used while 16 x the largest observed count stays below the limit, int32 otherwise."""
    if choice in ("u16", "int32"):
        return "u16" if choice == "u16" else True
    if choice != "auto":
        raise ValueError(f"events dtype {choice!r}: choose auto, u16 or int32")
    return "u16" if 16.0 * float(np.max(cases, initial=0.0)) < 65535.0 else True


def launch_forms(lay, device_arg, hmc="auto", moves="auto", env=os.environ):
    """The launch forms of the sampler for this process (`ChainSampler(hmc=..., moves=...)`).  "auto": the persistent
    whole-chip launches ("chunk", "paired") when this rank has its GPU to itself, the per-step forms ("chunk-launch",
    "paired-launch") when several ranks of the job were given the SAME device -- an explicit `--device` in a job with more
    than one rank on the node: torchrun hands every rank the same command line -- because two persistent launches cannot
    both be resident on one GPU (include/seir_hip.h, seir_sampler_desc::moves_mode).  Whatever is chosen here, a hand-off
    time-out at run time makes `ChainSampler` fall back by itself."""
    local_world = int(env.get("LOCAL_WORLD_SIZE", lay["world"]))
    shared = lay["world"] > 1 and device_arg is not None and local_world > 1
    return ("chunk-launch" if shared else "chunk") if hmc == "auto" else hmc, \
           ("paired-launch" if shared else "paired") if moves == "auto" else moves


def mcmc(data_file, output_file, config, seed=0, num_chains=1, device=None, pool_step_size=False, init_jitter=0.0,
         events_dtype="auto", hmc="auto", moves="auto"):
    """Constructs and runs the MCMC (covid19uk/inference/inference.py:473-608).

    Multi-GPU: launched as one process per GPU, every rank runs `num_chains` chains with
    global ids rank*num_chains ... (the Philox streams are keyed by the global id, so the draws of chain c
    do not depend on how the job is sharded) and writes its own posterior_chain{c}.hd5; there is no
    data-path collective.  `pool_step_size` adds the one optional exchange: an all_gather of one float64
    per chain after warm-up."""
    lay = job_layout(num_chains, device)                    # before any GPU call
    cov, cases, dates = read_inference_data(data_file)
    rng = np.random.default_rng(seed)                       # same imputation on every rank: one initial state per job
    B = int(num_chains)
    initial_state, events = model_spec.initial_conditions(cases, cov.N, rng)
    M, T = events.shape[0], events.shape[1]
    P = model_spec.num_params(M, T)
    cfg = event_kernel_config(config)
    num_samples = warmup_size() + int(config["num_burst_samples"]) * int(config["num_bursts"])
    cap = max(800, 2 * int(config["num_burst_samples"]))      # two halves: a burst runs while the previous one is written

    if lay["world"] > 1 and pool_step_size:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = os.environ.get("SEIR_DIST_BACKEND", "nccl")
            if backend == "nccl":
                torch.cuda.set_device(lay["device"])
                dist.init_process_group("nccl", device_id=torch.device("cuda", lay["device"]))
            else:
                dist.init_process_group(backend)
    model = SeirModel(cov, initial_state, max_chains=B, device=lay["device"])
    hmc_form, moves_form = launch_forms(lay, device, hmc, moves)
    if (hmc_form, moves_form) != ("chunk", "paired") and hmc == "auto" and moves == "auto":
        print(f"[rank {lay['rank']}] ranks of this job share device {lay['device']}: launch forms {hmc_form!r}, {moves_form!r} "
              "(one launch per leapfrog step / per pair of event updates)", file=sys.stderr, flush=True)
    sampler = ChainSampler(model, cfg, B, seed=seed, t_range=(max(T - 21, 0), T),
                           num_leapfrog_steps=hmc_kernel_kwargs_default()["num_leapfrog_steps"],
                           trace_capacity=cap, record_events=trace_events_dtype(cases, events_dtype),
                           first_chain_id=lay["first_chain_id"], hmc=hmc_form, moves=moves_form)
    u0 = dispersed_start(P, [lay["first_chain_id"] + c for c in range(B)], float(init_jitter), seed)
    sampler.set_state(u0, np.stack([events] * B))
    print("Initial logpi:", sampler.log_prob(), flush=True)

    total = lay["world"] * B
    names = [chain_file_name(output_file, lay["first_chain_id"] + c, total) for c in range(B)]
    posteriors = [Posterior(name, M, T, cfg["m"], num_samples, burst=int(config["num_burst_samples"]))
                  for name in names]
    run_mcmc(sampler, config, posteriors, pool_step_size=pool_step_size and total > 1)
    if sampler.recoveries:
        print(f"{len(sampler.recoveries)} burst(s) were run again after a hand-off time-out (shared GPU?)", flush=True)
    for post in posteriors:
        post.create_dataset("initial_state", initial_state)
        n = max(len(s) for s in dates)
        post.create_dataset("time", np.array(dates, dtype=f"S{n}"))
        print(f"Acceptance theta: {post['results/hmc/is_accepted'].mean()}")
        for key, label in zip(MOVE_KEYS, ("move S->E", "move E->I", "occult S->E", "occult E->I")):
            print(f"Acceptance {label}: {post[f'results/{key}/is_accepted'].mean()}")
        post.close()
    sampler.close()
    model.close()
    if lay["world"] > 1 and pool_step_size:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    return names


def main(argv=None):
    from argparse import ArgumentParser

    import yaml
    parser = ArgumentParser(description="Run MCMC inference algorithm")
    parser.add_argument("-c", "--config", type=str, help="Config file", required=True)
    parser.add_argument("-o", "--output", type=str, help="Output file", required=True)
    parser.add_argument("data_file", type=str, help="Data NetCDF file")
    parser.add_argument("--seed", type=int, default=0, help="RNG seed (the reference is unseeded)")
    parser.add_argument("--chains", type=int, default=1, help="independent chains on this GPU (per rank under torchrun)")
    parser.add_argument("--device", type=int, default=None, help="HIP device (default: LOCAL_RANK, 0 outside torchrun)")
    parser.add_argument("--pool-step-size", action="store_true",
                        help="sample with the geometric mean of all chains' adapted HMC step sizes (one all_gather)")
    parser.add_argument("--init-jitter", type=float, default=0.0,
                        help="sd of the N(0, sd^2) start of chains 1.. in the unconstrained space (chain 0 starts at 0 as the reference)")
    parser.add_argument("--events-dtype", choices=["auto", "u16", "int32"], default="auto",
                        help="width of the event counts in the device-side burst buffer (auto: 16 bit while 16 x the largest "
                             "observed count fits, else 32)")
    from ..sampler import HMC_MODES, MOVES_MODES
    parser.add_argument("--hmc", choices=["auto"] + sorted(HMC_MODES), default="auto",
                        help="launch form of the HMC update (auto: the persistent whole-trajectory launch, or one launch per "
                             "leapfrog step when ranks share a GPU); what is sampled does not depend on it")
    parser.add_argument("--moves", choices=["auto"] + sorted(MOVES_MODES), default="auto",
                        help="launch form of the event updates (auto: one persistent launch per sweep, or one launch per pair "
                             "of updates when ranks share a GPU)")
    args = parser.parse_args(argv)
    with open(args.config, "r") as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    mcmc(args.data_file, args.output, config["Mcmc"], seed=args.seed, num_chains=args.chains, device=args.device,
         pool_step_size=args.pool_step_size, init_jitter=args.init_jitter, events_dtype=args.events_dtype,
         hmc=args.hmc, moves=args.moves)


if __name__ == "__main__":
    main()
