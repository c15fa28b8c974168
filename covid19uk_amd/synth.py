"""Seeded synthetic inputs for tests and benchmarks.

The reference's real inputs need network downloads and a geopackage that is a
missing blob, so the workloads are built from the two data files it does ship
(LAD populations and commuting flows, compacted by tools/make_uk_covariates.py
into data/uk_lad2019.npz) plus seeded synthetic adjacency/area/weekday, and
epidemics simulated forward with the model's own chain-binomial process
(doc/lancs_space_model_concept.tex:256-275; the reference's equivalent is
DiscreteTimeStateTransitionModel.sample, used by posterior/predict.py:57-64).

Pure NumPy, host side, data preparation only.
"""
from __future__ import annotations

import os

import numpy as np

from . import model_spec as ms

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "uk_lad2019.npz")

WORKLOADS = {
    # name: (M, T)
    "ni11": (11, 32),
    "uk380": (380, 365),
    "syn2048": (2048, 730),
}

GENERATING_PARAMS = dict(alpha_0=-1.3, psi=0.5, gamma0=float(np.log(0.25)), gamma1=0.0,
                         beta_area=0.0, sigma_space=0.1)


def _ring_adjacency(M, k):
    """Each node linked to its k nearest neighbours by index on a ring (k even)."""
    A = np.zeros((M, M))
    idx = np.arange(M)
    for d in range(1, k // 2 + 1):
        A[idx, (idx + d) % M] = 1.0
        A[(idx + d) % M, idx] = 1.0
    np.fill_diagonal(A, 0.0)
    return A


def make_covariates(name: str, seed: int = 20210101) -> ms.Covariates:
    if name not in WORKLOADS:
        raise KeyError(f"unknown workload {name!r}; choose from {sorted(WORKLOADS)}")
    M, T = WORKLOADS[name]
    rng = np.random.default_rng(seed)
    if name in ("ni11", "uk380"):
        d = np.load(_DATA)
        codes = d["lad19cd"]
        sel = np.array([c.startswith("N") for c in codes]) if name == "ni11" \
            else np.ones(len(codes), bool)
        C = d["C"][np.ix_(sel, sel)].astype(np.float64)
        N = d["N"][sel].astype(np.float64)
        assert C.shape[0] == M
        if name == "ni11":
            A = _ring_adjacency(M, 2)
            A[0, 5] = A[5, 0] = 1.0
            A[2, 8] = A[8, 2] = 1.0
        else:
            A = _ring_adjacency(M, 6)
    else:
        N = np.floor(np.exp(rng.uniform(np.log(2.0e4), np.log(1.2e6), size=M)))
        C = rng.poisson(np.exp(rng.normal(1.4, 2.0, size=(M, M)))).astype(np.float64)
        np.fill_diagonal(C, 0.0)
        A = _ring_adjacency(M, 6)
    area = np.exp(rng.normal(np.log(5.0e8), 0.5, size=M))
    days = np.arange(T)
    weekday = (((days + 4) % 7) < 5).astype(np.float64)      # 2021-01-01 was a Friday
    W = np.ones(T)
    return ms.Covariates(C=C, W=W, N=N, adjacency=A, weekday=weekday, area=area)


def simulate_epidemic(cov: ms.Covariates, seed: int = 20210101, params=None,
                      alpha_t_sd: float = 0.0):
    """Forward chain-binomial simulation.  Returns (events [M,T,3] float64,
    initial_state [M,4] float64, true parameter dict)."""
    par = dict(GENERATING_PARAMS)
    if params:
        par.update(params)
    k = ms.derive_constants(cov)
    M, T = cov.M, cov.T
    rng = np.random.default_rng(seed)
    spatial = rng.normal(0.0, 1.0, size=M)
    alpha_t = rng.normal(0.0, alpha_t_sd, size=T - 1) if alpha_t_sd > 0 else np.zeros(T - 1)

    order = np.argsort(-k.N)
    I0 = rng.poisson(3.0, size=M).astype(np.float64)
    E0 = rng.poisson(3.0, size=M).astype(np.float64)
    I0[order[:5]] += 50.0
    E0[order[:5]] += 50.0
    state = np.stack([k.N - I0 - E0, E0, I0, np.zeros(M)], axis=-1)
    init = state.copy()

    a = par["alpha_0"] + np.concatenate([[0.0], np.cumsum(alpha_t)])
    b = par["beta_area"] * k.log_area_c + par["sigma_space"] * spatial
    events = np.zeros((M, T, 3))
    for t in range(T):
        S, E, I = state[:, 0], state[:, 1], state[:, 2]
        h = I + par["psi"] * k.W[t] * (k.Cstar @ (I / k.N))
        lam = np.exp(a[t] + b) * np.maximum(h, 0.0) / k.N + ms.RATE_FLOOR
        r_ir = np.exp(par["gamma0"] + par["gamma1"] * k.weekday_c[t])
        se = rng.binomial(S.astype(np.int64), -np.expm1(-lam))
        ei = rng.binomial(E.astype(np.int64), -np.expm1(-ms.NU))
        ir = rng.binomial(I.astype(np.int64), -np.expm1(-r_ir))
        events[:, t, 0], events[:, t, 1], events[:, t, 2] = se, ei, ir
        state = state + np.stack([-se, se - ei, ei - ir, ir], axis=-1)
    truth = dict(par, alpha_t=alpha_t, spatial_effect=spatial)
    return events, init, truth


def pack_params(truth, M, T):
    """Constrained parameter vector theta[P] in the order of inference.py:541-552."""
    th = np.zeros(ms.num_params(M, T))
    for i, n in enumerate(ms.PARAM_NAMES):
        th[i] = truth[n]
    th[6:6 + T - 1] = truth["alpha_t"]
    th[6 + T - 1:] = truth["spatial_effect"]
    return th


def unconstrain(theta):
    """theta -> u for the Softplus(low=eps) blocks of inference.py:525-535."""
    u = np.array(theta, dtype=np.float64, copy=True)
    y = u[..., :2] - np.finfo(np.float64).eps
    u[..., :2] = y + np.log(-np.expm1(-y))
    return u


def jitter_params(u, B, scale=0.1, seed=7, T=None):
    """B parameter vectors u ~ N(u, scale^2) (alpha_t jittered at its prior scale)."""
    rng = np.random.default_rng(seed)
    u = np.asarray(u, dtype=np.float64)
    z = rng.normal(size=(B, u.shape[0]))
    sd = np.full(u.shape[0], scale)
    if T is not None:
        sd[6:6 + T - 1] = 0.005
    return u[None, :] + sd[None, :] * z
