"""Thin a posterior.hd5 into a pickle (mirror of covid19uk/posterior/thin.py:7-21).

Host-side only: slices `samples/*` with range(start, end, by) of config["ThinPosterior"] and
carries `initial_state` along, as the reference does.
"""
import pickle as pkl

import numpy as np

from .. import hdf5io

SAMPLE_KEYS = ("psi", "sigma_space", "beta_area", "gamma0", "gamma1", "alpha_0", "alpha_t", "spatial_effect", "seir")


BURST_BYTES = 256 << 20      # rows of samples/seir are read in bursts of about this size


def _read_thinned(f, name, rows):
    """ds[rows] for rows = range(start, stop, step): strided hyperslabs, so only the kept draws are read
    (samples/seir of a full-UK run is ~23 GB; the reference slices it on disk too, thin.py:13)."""
    shape = f.shape(name)
    if len(rows) == 0:
        return np.empty((0,) + shape[1:])
    if rows.step < 0:                                   # h5py-style reversed slice: read forward, flip
        return _read_thinned(f, name, rows[::-1])[::-1]
    row_bytes = 8 * int(np.prod(shape[1:], dtype=np.int64)) if len(shape) > 1 else 8
    per = max(1, BURST_BYTES // row_bytes)
    parts = [f.read_rows(name, rows[i], min(per, len(rows) - i), rows.step) for i in range(0, len(rows), per)]
    return parts[0] if len(parts) == 1 else np.concatenate(parts, axis=0)


def thin_posterior(input_file, output_file, config):
    """covid19uk/posterior/thin.py:7-21: idx = slice(start, end, by) applied to every samples/* dataset."""
    with hdf5io.File(input_file, "r") as f:
        n = f.shape("/samples/psi")[0]
        sl = slice(config.get("start"), config.get("end"), config.get("by"))
        rows = range(*sl.indices(n))                    # slice semantics: negative / None / past-the-end as h5py
        out = {k: _read_thinned(f, f"/samples/{k}", rows) for k in SAMPLE_KEYS}
        out["initial_state"] = f.read("/initial_state")
    with open(output_file, "wb") as fh:
        pkl.dump(out, fh)
    return out


def main(argv=None):
    import argparse

    import yaml
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, required=True, help="Configuration file")
    parser.add_argument("-o", "--output", type=str, required=True, help="Output pkl file")
    parser.add_argument("samples", type=str, help="MCMC samples file (posterior.hd5)")
    args = parser.parse_args(argv)
    with open(args.config, "r") as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader)
    thin_posterior(args.samples, args.output, cfg["ThinPosterior"])


if __name__ == "__main__":
    main()
