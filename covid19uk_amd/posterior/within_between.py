"""Within-/between-location attributable infection pressure (mirror of
covid19uk/posterior/within_between.py:60-92).  The pressure components run in libseirhip's
k_within_between kernel; this file reduces the sampled events to the last state and
summarises.  Output: a csv with within_mean, between_mean, p_within_gt_between per location.
"""
import pickle as pkl

import numpy as np

from .. import model_spec
from ..inference.inference import read_inference_data
from ..seir import SeirModel


def calc_pressure_components(covariates, psi, state_last, device=0, initial_state=None):
    """(within, between) [n,M]; `state_last` [n,M,4] is the state at the last time index."""
    state_last = np.asarray(state_last, dtype=np.float64)
    init = state_last[0] if initial_state is None else initial_state
    W = np.asarray(covariates.W, dtype=np.float64).reshape(-1)
    with SeirModel(covariates, init, max_chains=1, device=device) as model:
        return model.within_between(psi, state_last[..., 2], W[-1])      # t = len(W) clips to len(W)-1


def within_between(input_files, output_file, device=0):
    cov, _, _ = read_inference_data(input_files[0])
    with open(input_files[1], "rb") as f:
        samples = pkl.load(f)
    init_state = np.asarray(samples["initial_state"], dtype=np.float64)
    events = np.asarray(samples["seir"], dtype=np.float64)
    # state at the last time index = init + all increments before it (gemlib compute_state, exclusive cumsum)
    inc = np.einsum("nmtx,xs->nms", events[:, :, :-1, :], model_spec.STOICHIOMETRY)
    state_last = init_state[None] + inc
    within, between = calc_pressure_components(cov, samples["psi"], state_last, device, init_state)
    rows = np.stack([within.mean(0), between.mean(0), (within > between).mean(0)], axis=1)
    with open(output_file, "w") as f:
        f.write("location,within_mean,between_mean,p_within_gt_between\n")
        for i, r in enumerate(rows):
            f.write(f"{i},{r[0]!r},{r[1]!r},{r[2]!r}\n")
    return within, between


def main(argv=None):
    from argparse import ArgumentParser
    parser = ArgumentParser()
    parser.add_argument("-d", "--datafile", type=str, help="Inference-data file", required=True)
    parser.add_argument("-s", "--samples", type=str, help="Posterior samples pickle", required=True)
    parser.add_argument("-o", "--output", type=str, help="Output csv")
    args = parser.parse_args(argv)
    within_between([args.datafile, args.samples], args.output)


if __name__ == "__main__":
    main()
