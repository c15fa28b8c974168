"""Thin a posterior.hd5 into a pickle (mirror of covid19uk/posterior/thin.py:7-21).

Host-side only: slices `samples/*` with range(start, end, by) of config["ThinPosterior"] and
carries `initial_state` along, as the reference does.
"""
import pickle as pkl

import numpy as np

from .. import hdf5io

SAMPLE_KEYS = ("psi", "sigma_space", "beta_area", "gamma0", "gamma1", "alpha_0", "alpha_t", "spatial_effect", "seir")


def thin_posterior(input_file, output_file, config):
    idx = np.arange(int(config["start"]), int(config["end"]), int(config["by"]))
    with hdf5io.File(input_file, "r") as f:
        n = f.shape("/samples/psi")[0]
        idx = idx[idx < n]
        out = {k: f.read(f"/samples/{k}")[idx] for k in SAMPLE_KEYS}
        out["initial_state"] = f.read("/initial_state")
    with open(output_file, "wb") as fh:
        pkl.dump(out, fh)
    return out


if __name__ == "__main__":
    import argparse

    import yaml
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, required=True, help="Configuration file")
    parser.add_argument("-o", "--output", type=str, required=True, help="Output pkl file")
    parser.add_argument("samples", type=str, help="MCMC samples file (posterior.hd5)")
    args = parser.parse_args()
    with open(args.config, "r") as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader)
    thin_posterior(args.samples, args.output, cfg["ThinPosterior"])
