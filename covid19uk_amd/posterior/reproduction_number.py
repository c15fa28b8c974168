"""Time-varying reproduction number R_it / R_t from thinned posterior draws (mirror of
covid19uk/posterior/reproduction_number.py:50-88).  The next-generation-matrix arithmetic
(model_spec.py:302-368) runs in libseirhip's k_rt kernel; this file moves arrays.

Output: HDF5 group `posterior_predictive` with `R_it` [iteration, time, location] and the
population-weighted `R_t` [iteration, time] (reproduction_number.py:76-79).
"""
import pickle as pkl

import numpy as np

from .. import hdf5io
from ..inference.inference import read_inference_data, read_location_names
from ..seir import SeirModel

CHUNKSIZE = 50            # reproduction_number.py:47


def pack_theta(samples):
    """Constrained parameter matrix [n,P] in the order of inference.py:541-552."""
    cols = [np.asarray(samples[k], dtype=np.float64).reshape(len(samples["psi"]), -1)
            for k in ("psi", "sigma_space", "beta_area", "gamma0", "gamma1", "alpha_0", "alpha_t", "spatial_effect")]
    return np.concatenate(cols, axis=1)


def calc_posterior_rit(samples, initial_state, covariates, device=0):
    theta = pack_theta(samples)
    events = np.asarray(samples["seir"], dtype=np.float64)
    with SeirModel(covariates, initial_state, max_chains=CHUNKSIZE, device=device) as model:
        return model.reproduction_number(theta, events)


def reproduction_number(input_files, output_file, device=0):
    cov, _, dates = read_inference_data(input_files[0])
    with open(input_files[1], "rb") as f:
        samples = pkl.load(f)
    initial_state = samples.pop("initial_state")
    r_it = calc_posterior_rit(samples, initial_state, cov, device)
    N = np.asarray(cov.N, dtype=np.float64).reshape(-1)
    r_t = (r_it * (N / N.sum())[None, None, :]).sum(-1)
    # the `posterior_predictive` group as the reference's xarray.Dataset.to_netcdf writes it
    # (reproduction_number.py:73-88): R_it on (iteration, time, location), R_t on (iteration, time)
    locations = read_location_names(input_files[0]) or [str(i) for i in range(r_it.shape[2])]
    try:
        times = np.array(dates, dtype="datetime64[D]") if "-" in str(dates[0]) else np.arange(r_it.shape[1])
    except (ValueError, TypeError):
        times = np.arange(r_it.shape[1])
    with hdf5io.File(output_file, "a") as f:
        f.write_netcdf_group("posterior_predictive",
                             {"iteration": np.arange(r_it.shape[0]), "time": times, "location": locations},
                             {"R_it": (("iteration", "time", "location"), r_it), "R_t": (("iteration", "time"), r_t)})
    return r_it, r_t


def main(argv=None):
    from argparse import ArgumentParser
    parser = ArgumentParser()
    parser.add_argument("samples", type=str, help="A pickle file with MCMC samples")
    parser.add_argument("-d", "--data", type=str, help="The inference-data file", required=True)
    parser.add_argument("-o", "--output", type=str, help="The output file", required=True)
    args = parser.parse_args(argv)
    reproduction_number([args.data, args.samples], args.output)


if __name__ == "__main__":
    main()
