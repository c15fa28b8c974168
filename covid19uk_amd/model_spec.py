"""Host-side model specification: constants and covariate preparation.

Mirrors the parameter-free part of the reference's `covid19uk/model_spec.py`
so that the device context can be created from the same `constant_data`
arrays the reference feeds `CovidUK(covariates, initial_state, initial_step,
num_steps)` (model_spec.py:139).  The parameter-dependent arithmetic (rates,
chain-binomial log-probability, priors, gradient) lives in the HIP library
(`covid19uk_amd/csrc`) behind the C-ABI of `include/seir_hip.h`; there is no
CPU implementation of it in this package.
"""
from __future__ import annotations

import dataclasses

import numpy as np

DTYPE = np.float64                       # model_spec.py:22
STOICHIOMETRY = np.array(                # model_spec.py:24
    [[-1, 1, 0, 0], [0, -1, 1, 0], [0, 0, -1, 1]], dtype=DTYPE)
TIME_DELTA = 1.0                         # model_spec.py:25
NU = 0.28                                # model_spec.py:26, E->I rate assumed known
RATE_FLOOR = 1.0e-9                      # model_spec.py:264-266
CAR_RHO = 0.25                           # model_spec.py:174
NUM_GLOBAL = 6                           # psi, sigma_space, beta_area, gamma0, gamma1, alpha_0

PARAM_NAMES = ("psi", "sigma_space", "beta_area", "gamma0", "gamma1", "alpha_0")


@dataclasses.dataclass
class Covariates:
    """The `constant_data` group of the reference's inference-data file
    (model_spec.py:88-105): C [M,M] (dest, src), W [T], N [M],
    adjacency [M,M], weekday [T], area [M]."""
    C: np.ndarray
    W: np.ndarray
    N: np.ndarray
    adjacency: np.ndarray
    weekday: np.ndarray
    area: np.ndarray

    @property
    def M(self):
        return int(np.asarray(self.N).reshape(-1).shape[0])

    @property
    def T(self):
        return int(np.asarray(self.W).reshape(-1).shape[0])


@dataclasses.dataclass
class DerivedConstants:
    """What `seir()` (model_spec.py:216-230) and `spatial_effect()` (:171-181)
    build from the covariates before any parameter is seen."""
    Cstar: np.ndarray          # [M,M]
    N: np.ndarray              # [M]
    W: np.ndarray              # [T]
    weekday_c: np.ndarray      # [T]
    log_area_c: np.ndarray     # [M]
    car_Q: np.ndarray          # [M,M]
    car_half_logdet: float


def derive_constants(cov: Covariates) -> DerivedConstants:
    C = np.array(cov.C, dtype=DTYPE)
    if C.ndim != 2 or C.shape[0] != C.shape[1]:
        raise ValueError("C must be square [M,M]")
    M = C.shape[0]
    C[np.arange(M), np.arange(M)] = 0.0
    Cstar = C + C.T
    Cstar[np.arange(M), np.arange(M)] = -C.sum(axis=0)
    N = np.asarray(cov.N, dtype=DTYPE).reshape(-1)
    W = np.asarray(cov.W, dtype=DTYPE).reshape(-1)
    wd = np.asarray(cov.weekday, dtype=DTYPE).reshape(-1)
    area = np.asarray(cov.area, dtype=DTYPE).reshape(-1)
    adj = np.asarray(cov.adjacency, dtype=DTYPE)
    if N.shape[0] != M or area.shape[0] != M or adj.shape != (M, M):
        raise ValueError("covariate shapes disagree on M")
    if wd.shape[0] != W.shape[0]:
        raise ValueError("W and weekday disagree on T")
    if np.any(N <= 0):
        raise ValueError("population sizes must be positive")
    la = np.log(area / 1.0e8)
    Q = np.diag(adj.sum(axis=1)) - CAR_RHO * adj
    # Cholesky doubles as the positive-definiteness check the reference's
    # tf.linalg.cholesky(inv(Q)) performs implicitly.
    L = np.linalg.cholesky(Q)
    return DerivedConstants(
        Cstar=np.ascontiguousarray(Cstar), N=N, W=W,
        weekday_c=wd - wd.mean(), log_area_c=la - la.mean(),
        car_Q=np.ascontiguousarray(Q),
        car_half_logdet=float(np.sum(np.log(np.diag(L)))))


def num_params(M: int, T: int) -> int:
    """Length of the flat parameter vector (inference.py:533,563-573)."""
    return NUM_GLOBAL + (T - 1) + M


def compute_state(initial_state, events):
    """Host helper used only for data preparation (inference.py:500-513):
    state at the start of each day, [M,T,4]."""
    inc = np.einsum("...tx,xs->...ts", np.asarray(events, DTYPE), STOICHIOMETRY)
    cs = np.cumsum(inc, axis=-2)
    cs = np.concatenate([np.zeros_like(cs[..., :1, :]), cs[..., :-1, :]], axis=-2)
    return np.asarray(initial_state, DTYPE)[..., None, :] + cs


# ---------------------------------------------------------------------------
# Initialisation of the censored events (host side, once per run)
# ---------------------------------------------------------------------------
def distribute_geom(events, rate, rng, delta_t=1.0):
    """Spread `events` [M,T] back in time with geometric waiting times of
    per-step probability 1-exp(-rate*delta_t) (covid19uk/util.py:120-145).
    Returns [M, L, T]; slice l holds the events whose lag is l (lag 0 is empty,
    as in the reference, whose loop counter starts at 1)."""
    ev = np.asarray(events, dtype=np.int64)
    prob = 1.0 - np.exp(-rate * delta_t)
    out = [np.zeros_like(ev)]
    remaining = ev.copy()
    while remaining.sum() > 0:
        hit = rng.binomial(remaining, prob)
        out.append(hit)
        remaining = remaining - hit
    return np.stack(out, axis=1).astype(DTYPE)


def reduce_diagonals(m):
    """[M, L, T] -> [M, L+T-1]: element (l, t) lands on day t - l + L - 1 (util.py:148-159)."""
    M, L, T = m.shape
    out = np.zeros((M, L + T - 1), dtype=m.dtype)
    for lag in range(L):
        out[:, L - 1 - lag:L - 1 - lag + T] += m[:, lag, :]
    return out


def impute_previous_cases(events, rate, rng, delta_t=1.0):
    """util.py:162-182: back-distributed events with the leading all-zero days trimmed,
    and the number of days the series grew by."""
    distn = distribute_geom(events, rate, rng, delta_t)
    prev = reduce_diagonals(distn)
    total = prev.sum(axis=-2)
    num_zero_days = total.shape[-1] - int(np.count_nonzero(np.cumsum(total)))
    return prev[..., num_zero_days:], distn.shape[-2] - num_zero_days


def impute_censored_events(cases, rng):
    """model_spec.py:108-126 of the reference: S->E and E->I events imputed from the observed
    I->R series with the reference's rates 0.25 and 0.5.  cases [M,T] -> events [M,T',3]."""
    cases = np.asarray(cases, dtype=DTYPE)
    ei_events, lag_ei = impute_previous_cases(cases, 0.25, rng)
    se_events, lag_se = impute_previous_cases(ei_events, 0.5, rng)
    ir_events = np.pad(cases, ((0, 0), (lag_ei + lag_se - 2, 0)))
    ei_events = np.pad(ei_events, ((0, 0), (lag_se - 1, 0)))
    return np.stack([se_events, ei_events, ir_events], axis=-1)


def initial_conditions(cases, N, rng, extra_weeks=3):
    """inference.py:487-513: repeat the last week `extra_weeks` times, impute the censored
    events, and return (initial_state [M,4], events [M,T,3]) for the observed window."""
    cases = np.asarray(cases, dtype=DTYPE)
    extra = np.tile(cases[:, -7:], (1, extra_weeks))
    padded = np.concatenate([cases, extra], axis=-1)
    events = impute_censored_events(padded, rng)
    init = np.concatenate([np.asarray(N, DTYPE).reshape(-1, 1), np.zeros_like(events[:, 0, :])], axis=-1)
    state = compute_state(init, events)
    start = state.shape[1] - padded.shape[1]
    return state[:, start, :].copy(), events[:, start:-extra.shape[1], :].copy()
