"""In-sample / out-of-sample predictions from thinned posterior draws (mirror of
covid19uk/posterior/predict.py:13-160).  The chain-binomial simulation itself runs in
libseirhip's `seir_simulate` (csrc/sim_kernels.h); this file resolves the model's time indexing
and moves arrays.

Output: HDF5 group `predictions` with `events` [iteration, location, time, event] and
`initial_state` [iteration, location, state] (predict.py:124-146; the reference writes the same
two variables through xarray/NetCDF4).

The reference is unseeded; `seed` selects this build's Philox stream (and the NumPy generator
that draws the out-of-sample alpha_t random walk, which TFP would draw from its prior,
model_spec.py:158-165).
"""
import pickle as pkl

import numpy as np

from .. import hdf5io, model_spec
from ..inference.inference import read_inference_data, read_location_names
from ..seir import SeirModel

ALPHA_T_SCALE = 0.005          # model_spec.py:158-165


def log_baseline_path(alpha_0, alpha_t, initial_step, num_steps):
    """a_t [n, num_steps] for absolute days t = initial_step + s with the indexing of
    model_spec.py:245-256: alpha_0 at t == 0, else (alpha_0 + cumsum(alpha_t))[clip(t-1, 0, len-1)]."""
    alpha_0 = np.asarray(alpha_0, dtype=np.float64).reshape(-1)
    alpha_t = np.asarray(alpha_t, dtype=np.float64).reshape(alpha_0.shape[0], -1)
    t = initial_step + np.arange(num_steps)
    if alpha_t.shape[1] == 0:
        return np.repeat(alpha_0[:, None], num_steps, axis=1)
    b = alpha_0[:, None] + np.cumsum(alpha_t, axis=1)
    idx = np.clip(t - 1, 0, alpha_t.shape[1] - 1)
    return np.where((t == 0)[None, :], alpha_0[:, None], b[:, idx])


def clipped(values, initial_step, num_steps):
    """values[clip(t, 0, len-1)] for t = initial_step .. (model_spec.py:234-241)."""
    values = np.asarray(values, dtype=np.float64).reshape(-1)
    return values[np.clip(initial_step + np.arange(num_steps), 0, values.shape[0] - 1)]


def predicted_incidence(posterior_samples, init_state, covar_data: model_spec.Covariates, init_step, num_steps,
                        out_of_sample=False, seed=0, device=0):
    """Simulate forward from the posterior state at `init_step` for `num_steps` days.
    Returns (estimated initial state [n,M,4], events [n,M,num_steps,3]) like predict.py:13-70."""
    samples = dict(posterior_samples)
    seir = np.asarray(samples.pop("seir"), dtype=np.float64)
    n = seir.shape[0]
    posterior_state = model_spec.compute_state(np.asarray(init_state, dtype=np.float64), seir)   # [n,M,T,4]
    new_init = np.ascontiguousarray(posterior_state[:, :, init_step, :])
    alpha_0 = np.asarray(samples["alpha_0"], dtype=np.float64).reshape(n)
    alpha_t = np.asarray(samples["alpha_t"], dtype=np.float64).reshape(n, -1)
    if out_of_sample:
        # predict.py:38-48: restart the random walk at its value at init_step and let the prior
        # re-simulate the increments
        b = alpha_0[:, None] + np.cumsum(alpha_t, axis=1)
        if init_step > 0:
            alpha_0 = b[:, init_step - 1]
        rng = np.random.default_rng(seed)
        alpha_t = rng.normal(0.0, ALPHA_T_SCALE, size=(n, max(num_steps - 1, 0)))
    a_path = log_baseline_path(alpha_0, alpha_t, init_step, num_steps)
    weekday = np.asarray(covar_data.weekday, dtype=np.float64).reshape(-1)
    wd = clipped(weekday - weekday.mean(), init_step, num_steps)         # model_spec.py:224-225,239-241
    W = clipped(covar_data.W, init_step, num_steps)
    par = np.stack([np.asarray(samples[k], dtype=np.float64).reshape(n)
                    for k in ("psi", "sigma_space", "beta_area", "gamma0", "gamma1")], axis=1)
    spatial = np.asarray(samples["spatial_effect"], dtype=np.float64).reshape(n, -1)
    # the context supplies Cstar, N and log-area; its own W/weekday tables (length T) are not used
    # by the simulator, so the prediction calendar may be longer than the observation window
    Wc = np.asarray(covar_data.W, dtype=np.float64).reshape(-1)
    ctx_cov = model_spec.Covariates(C=covar_data.C, W=Wc, N=covar_data.N, adjacency=covar_data.adjacency,
                                    weekday=np.zeros_like(Wc), area=covar_data.area)
    with SeirModel(ctx_cov, np.asarray(init_state, dtype=np.float64), max_chains=1, device=device) as model:
        events = model.simulate(par, a_path, spatial, W, wd, new_init, seed=seed)
    return new_init, events


def prediction_weekday(dates, total_days, fallback):
    """(weekday < 5) for `total_days` days from the first observation date (predict.py:101-112);
    falls back to the covariate when the file carries no calendar dates."""
    try:
        if "-" not in str(dates[0]):
            raise ValueError("not a calendar date")
        origin = np.datetime64(dates[0], "D")
    except (ValueError, TypeError):
        return np.asarray(fallback, dtype=np.float64).reshape(-1), None
    days = origin + np.arange(total_days).astype("timedelta64[D]")
    dow = (days.astype("datetime64[D]").view("int64") - 4) % 7            # 1970-01-01 was a Thursday; Monday = 0
    return (dow < 5).astype(np.float64), days


def predict(data, posterior_samples, output_file, initial_step, num_steps, out_of_sample=False, seed=0, device=0):
    cov, _, dates = read_inference_data(data)
    with open(posterior_samples, "rb") as f:
        samples = pkl.load(f)
    initial_state = samples.pop("initial_state")
    if initial_step < 0:
        initial_step = np.asarray(samples["seir"]).shape[-2] + initial_step
    weekday, days = prediction_weekday(dates, initial_step + num_steps, cov.weekday)
    cov = model_spec.Covariates(C=cov.C, W=cov.W, N=cov.N, adjacency=cov.adjacency, weekday=weekday, area=cov.area)
    init, events = predicted_incidence(samples, initial_state, cov, initial_step, num_steps, out_of_sample,
                                       seed=seed, device=device)
    # the `predictions` group as the reference's xarray.Dataset.to_netcdf writes it (predict.py:124-147): events on
    # (iteration, location, time, event), initial_state on (iteration, location, state), every dimension a coordinate
    locations = read_location_names(data) or [str(i) for i in range(events.shape[1])]
    times = days[initial_step:] if days is not None else np.arange(initial_step, initial_step + events.shape[2])
    with hdf5io.File(output_file, "a") as f:
        f.write_netcdf_group("predictions",
                             {"iteration": np.arange(events.shape[0]), "location": locations, "time": times,
                              "event": np.arange(events.shape[3]), "state": np.arange(init.shape[-1])},
                             {"events": (("iteration", "location", "time", "event"), events),
                              "initial_state": (("iteration", "location", "state"), init)})
    return init, events


def main(argv=None):
    from argparse import ArgumentParser
    parser = ArgumentParser()
    parser.add_argument("-i", "--initial-step", type=int, default=0, help="Initial step")
    parser.add_argument("-n", "--num-steps", type=int, default=1, help="Number of steps")
    parser.add_argument("-o", "--out-of-sample", action="store_true",
                        help="Out of sample prediction (sample alpha_t)")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("data_pkl", type=str, help="Inference-data file")
    parser.add_argument("posterior_samples_pkl", type=str, help="Posterior samples pickle")
    parser.add_argument("output_file", type=str, help="Output file")
    args = parser.parse_args(argv)
    predict(args.data_pkl, args.posterior_samples_pkl, args.output_file, args.initial_step, args.num_steps,
            args.out_of_sample, seed=args.seed)


if __name__ == "__main__":
    main()
