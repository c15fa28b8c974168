"""Shared builders for the parity tests (seeded inputs, oracle handles)."""
import os

import numpy as np

from covid19uk_amd import model_spec as ms
from covid19uk_amd import synth
from oracle import seir_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def small_covariates(M, T, seed):
    """Random dense covariates for micro cases (asymmetric C, ring adjacency)."""
    rng = np.random.default_rng(seed)
    C = rng.integers(0, 400, size=(M, M)).astype(np.float64)
    N = rng.integers(5_000, 50_000, size=M).astype(np.float64)
    W = rng.uniform(0.5, 1.5, size=T)
    weekday = (np.arange(T) % 7 < 5).astype(np.float64)
    area = rng.uniform(1e8, 9e8, size=M)
    if M == 1:
        A = np.ones((1, 1))        # a lone node needs d>0 for a proper CAR precision
    elif M == 2:
        A = np.array([[0.0, 1.0], [1.0, 0.0]])
    else:
        A = synth._ring_adjacency(M, 2)
    return ms.Covariates(C=C, W=W, N=N, adjacency=A, weekday=weekday, area=area)


def oracle_constants(cov, init):
    return so.make_constants(cov.C, cov.N, cov.W, cov.weekday, cov.area, cov.adjacency, init)


def build_case(name, seed=20210101, alpha_t_sd=0.0):
    """name in synth.WORKLOADS, 'micro_MxT', 'slow_MxT' or 'slower_MxT' -> dict(cov, events, init, u, k).

    'slow_MxT': populations 40 times larger and a reproduction number just above one, so that the epidemic is
    still running on the last day of a series of 400+ days (a 'micro' epidemic burns out after ~130 days and
    would leave the day chunks beyond it -- the ones the long-series kernel forms exist for -- without events)."""
    if name in synth.WORKLOADS:
        cov = synth.make_covariates(name, seed)
        params = None
    else:
        kind = name.split("_")[0]
        M, T = (int(x) for x in name.split("_")[1].split("x"))
        cov = small_covariates(M, T, seed)
        params = dict(alpha_0=-0.5)
        if kind in ("slow", "slower"):
            import dataclasses
            cov = dataclasses.replace(cov, N=cov.N * (40.0 if kind == "slow" else 400.0))
            params = dict(alpha_0=-1.45 if kind == "slow" else -1.5)      # "slower": still running after 800 days
    events, init, truth = synth.simulate_epidemic(cov, seed, alpha_t_sd=alpha_t_sd, params=params)
    theta = synth.pack_params(truth, cov.M, cov.T)
    u = synth.unconstrain(theta)
    return dict(cov=cov, events=events, init=init, u=u, theta=theta,
                k=oracle_constants(cov, init))


from oracle import c_binding


def c_oracle():
    return c_binding.lib()


def c_oracle_eval(k, u, events, stable=1, want_grad=False):
    return c_binding.evaluate(k, u, events, stable, want_grad)
