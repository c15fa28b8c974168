"""CPU oracle (NumPy) for the reproduction-number post-processing.

TEST INFRASTRUCTURE ONLY.  Restates, literally,
  next_generation_matrix_fn   covid19uk/model_spec.py:302-368
  calc_posterior_rit          covid19uk/posterior/reproduction_number.py:13-44
  R_t weighting               covid19uk/posterior/reproduction_number.py:76-77
These two reference functions are pure TensorFlow arithmetic with no gemlib code inside
(compute_state aside), so this restatement follows the reference text line by line; it is
still not executed against TensorFlow here (not installed): parity unpinned, pinned by the
hand case and the independent loop form in tests/test_rt.py.

Quirks kept on purpose: the NGM indexes b_t with t, not t-1 as the
model does (model_spec.py:336-343 vs :245-256); log-area enters by ROW and the spatial effect
by COLUMN of the matrix (:345-349 broadcasting); 1-exp(-rate) is evaluated naively.
"""
import numpy as np

from . import seir_oracle as so


def next_generation_matrix(t, state_t, par, k: so.ModelConstants, stable=False):
    """[M,M] matrix of model_spec.py:316-365 for day t and state [M,4].  stable=True evaluates
    1-exp(-rate) as -expm1(-rate): the literal form loses ~1e-16/rate (rates are ~1e-8..1e-5)."""
    T1 = len(par["alpha_t"])
    b_t = par["alpha_0"] + np.cumsum(par["alpha_t"])                      # :331
    alpha_t_ = par["alpha_0"] if t == 0 else b_t[min(max(t, 0), T1 - 1)]  # :332-343
    w_t = k.W[min(max(t, 0), len(k.W) - 1)]                               # :329-330
    eta = alpha_t_ + par["beta_area"] * k.log_area_c[:, None] + par["sigma_space"] * par["spatial_effect"]  # :345-349
    M = k.M
    infec_rate = np.exp(eta) * (np.eye(M) + par["psi"] * w_t * k.Cstar / k.N[None, :]) / k.N[:, None]      # :350-357
    infec_prob = -np.expm1(-infec_rate) if stable else 1.0 - np.exp(-infec_rate)   # :358
    expected_new_infec = infec_prob * state_t[:, 0][:, None]              # :360
    expected_infec_period = 1.0 / (1.0 - np.exp(-np.exp(par["gamma0"])))  # :361-363
    return expected_new_infec * expected_infec_period


def posterior_rit(theta, events, k: so.ModelConstants, stable=False):
    """theta [n,P] constrained draws, events [n,M,T,3] -> R_it [n,T,M] (sum over destinations,
    reproduction_number.py:41) and R_t [n,T] (population-weighted, :76-77)."""
    n = theta.shape[0]
    M, T = k.M, k.T
    R = np.empty((n, T, M))
    for s in range(n):
        par = so.unpack(theta[s], M, T)
        state = so.compute_state(k.initial_state, events[s])
        for t in range(T):
            R[s, t] = next_generation_matrix(t, state[:, t, :], par, k, stable).sum(axis=-2)
    weight = k.N / k.N.sum()
    return R, (R * weight[None, None, :]).sum(-1)


def pressure_components(psi, state_last, C, N, W_last):
    """within/between fractions [n,M] -- literal restatement of
    covid19uk/posterior/within_between.py:13-57 (`C` raw [dest,src], diagonal zeroed there)."""
    C = np.array(C, dtype=np.float64)
    np.fill_diagonal(C, 0.0)
    N = np.asarray(N, dtype=np.float64).reshape(-1)
    psi = np.asarray(psi, dtype=np.float64).reshape(-1)
    I = np.asarray(state_last, dtype=np.float64)[..., 2]
    within = I - psi[:, None] * I / N * W_last * C.sum(axis=-2)
    between = psi[:, None] * W_last * (I / N) @ (C + C.T).T
    total = within + between
    return within / total, between / total
