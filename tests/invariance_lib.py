"""Shared pieces of the invariant-distribution tests (tests/test_invariance.py on the CPU oracle,
tests/test_invariance_gpu.py on the HIP sampler).

Why these tests exist.  The Metropolis-Hastings event kernels re-state gemlib code that is not under
/root/reference (mcmc_kernel_factory.py:63-113 only shows how they are configured), so nothing the
reference ships can pin them, and a bookkeeping error mirrored in the oracle and the HIP kernels would
pass every draw-by-draw comparison.  What does NOT depend on gemlib is the defining property of the
kernels: each one must leave the model's own conditional distribution invariant.  Two constructions
test exactly that with independent replicates (no autocorrelation to argue about):

  * enumerated toy -- a 2-LAD model small enough that every feasible event tensor can be listed, so
    the exact posterior p(z | theta, y) is a table.  Starts are drawn from the table, a kernel is
    applied k times, and the end states are chi-square tested against the table.  An invariant
    kernel maps exact draws to exact draws whatever its mixing or reducibility.
  * exact-sample (Geweke-style joint) test -- (z, y) ~ p(. | theta) by forward simulation (for the
    HIP path: the device simulator), then k sweeps of the event kernels given y;
    moments of (z, y) before and after must agree (paired z-tests over replicates).

z = the latent S->E / E->I events, y = the observed I->R events, theta fixed.  The density is the
pinned one (oracle/seir_oracle.py: scipy / mpmath / literal-Multinomial triangulated).
"""
import math

import numpy as np
from scipy import stats

from covid19uk_amd import model_spec as ms
from oracle import seir_oracle as so

KERNELS = ("move/S->E", "move/E->I", "occult/S->E", "occult/E->I")


def only(kernel):
    """disable list that leaves exactly `kernel` (or all four event kernels for "all") running."""
    keep = KERNELS if kernel == "all" else (kernel,)
    return ("hmc",) + tuple(k for k in KERNELS if k not in keep)


# ----------------------------------------------------------------------------------------------
# enumerated toy
# ----------------------------------------------------------------------------------------------
def _row_paths(S0, E0, I0, kir_row):
    """Every (k_se[t], k_ei[t]) path of one LAD that keeps all compartments feasible given k_ir."""
    T = len(kir_row)
    out = []

    def rec(t, S, E, I, acc):
        if t == T:
            out.append(tuple(acc))
            return
        if I < kir_row[t]:
            return
        for a in range(S + 1):
            for b in range(E + 1):
                rec(t + 1, S - a, E + a - b, I + b - kir_row[t], acc + [(a, b)])
    rec(0, S0, E0, I0, [])
    return out


def toy_problem(T=4):
    """2 LADs x T days with single-digit populations: dict(cov, init, k, u, states [n,2,T,3],
    logp [n], prob [n], index {bytes: i}).  Rates are of order 0.2-0.4 per day so that the posterior
    mass is spread over hundreds of tensors."""
    init = np.array([[2.0, 1.0, 1.0, 0.0], [1.0, 0.0, 1.0, 0.0]])
    kir = np.array([[0, 1, 0, 1], [0, 0, 1, 0]])[:, :T]
    C = np.array([[0.0, 1.0], [2.0, 0.0]])
    cov = ms.Covariates(C=C, W=np.ones(T), N=init.sum(axis=1), adjacency=np.array([[0.0, 1.0], [1.0, 0.0]]),
                        weekday=np.array([1.0, 1.0, 0.0, 1.0])[:T], area=np.array([2.0e8, 4.0e8]))
    k = so.make_constants(cov.C, cov.N, cov.W, cov.weekday, cov.area, cov.adjacency, init)
    theta = np.zeros(k.P)
    theta[:6] = (0.5, 0.2, 0.3, math.log(0.3), 0.2, math.log(1.2))      # psi sigma beta gamma0 gamma1 alpha_0
    theta[6:6 + T - 1] = np.linspace(-0.2, 0.2, T - 1)
    theta[6 + T - 1:] = (0.5, -0.7)
    y = theta[:2] - np.finfo(float).eps
    u = theta.copy()
    u[:2] = y + np.log(-np.expm1(-y))
    paths = [_row_paths(int(init[m, 0]), int(init[m, 1]), int(init[m, 2]), kir[m]) for m in range(2)]
    states = np.zeros((len(paths[0]) * len(paths[1]), 2, T, 3))
    i = 0
    for p0 in paths[0]:
        for p1 in paths[1]:
            states[i, 0, :, :2] = p0
            states[i, 1, :, :2] = p1
            states[i, :, :, 2] = kir
            i += 1
    logp = np.array([so.joint_log_prob(u, s, k, "stable") for s in states])
    assert np.isfinite(logp).all()
    w = np.exp(logp - logp.max())
    prob = w / w.sum()
    index = {s.astype(np.int8).tobytes(): j for j, s in enumerate(states)}
    return dict(cov=cov, init=init, k=k, u=u, theta=theta, states=states, logp=logp, prob=prob, index=index)


def state_indices(toy, events):
    """events [R,2,T,3] -> index into toy["states"] (KeyError for a tensor outside the feasible set)."""
    return np.array([toy["index"][np.asarray(e).astype(np.int8).tobytes()] for e in events])


def chi_square(idx, prob, min_expected=8.0):
    """Pearson chi-square of observed state indices against the exact table; cells with small
    expectation are pooled (sorted by probability, greedily, so that every bin expects >= min_expected).
    Returns (statistic, dof, p_value)."""
    R = len(idx)
    counts = np.bincount(idx, minlength=len(prob)).astype(float)
    order = np.argsort(-prob)
    obs, exp = [], []
    o = e = 0.0
    for j in order:
        o += counts[j]
        e += R * prob[j]
        if e >= min_expected:
            obs.append(o)
            exp.append(e)
            o = e = 0.0
    if e > 0:                      # the tail joins the last bin
        obs[-1] += o
        exp[-1] += e
    obs, exp = np.array(obs), np.array(exp)
    stat = float(((obs - exp) ** 2 / exp).sum())
    dof = len(obs) - 1
    return stat, dof, float(stats.chi2.sf(stat, dof))


# ----------------------------------------------------------------------------------------------
# exact-sample (joint) test
# ----------------------------------------------------------------------------------------------
def small_population_case(M=3, T=6, seed=5, gamma0=math.log(0.3), alpha_0=math.log(0.9)):
    """A few LADs with populations of a few hundred and a handful of infectives: event counts are
    small integers, where the feasibility bounds and the occult add/delete bookkeeping matter most.
    Returns dict(cov, init, k, u, theta).  With a removal rate of ~0.01/day and an infection rate just above it
    (gamma0, alpha_0) the epidemic is a slow trickle of single events that is still going after 400 days: the
    long-series forms of the kernels then see events in every day chunk."""
    rng = np.random.default_rng(seed)
    N = rng.integers(150, 400, size=M).astype(float)
    C = rng.integers(0, 12, size=(M, M)).astype(float)      # psi W colsum(C)/N < 1: the hazard stays positive
    np.fill_diagonal(C, 0.0)
    A = np.zeros((M, M))
    for m in range(M):
        A[m, (m + 1) % M] = A[(m + 1) % M, m] = 1.0
    np.fill_diagonal(A, 0.0)
    cov = ms.Covariates(C=C, W=rng.uniform(0.7, 1.3, size=T), N=N, adjacency=A,
                        weekday=(np.arange(T) % 7 < 5).astype(float), area=rng.uniform(1e8, 9e8, size=M))
    E0, I0 = rng.integers(2, 7, size=M).astype(float), rng.integers(2, 7, size=M).astype(float)
    init = np.stack([N - E0 - I0, E0, I0, np.zeros(M)], axis=-1)
    k = so.make_constants(cov.C, cov.N, cov.W, cov.weekday, cov.area, cov.adjacency, init)
    theta = np.zeros(k.P)
    theta[:6] = (0.5, 0.2, 0.1, gamma0, 0.1, alpha_0)
    theta[6:6 + T - 1] = rng.normal(0.0, 0.05, size=T - 1)
    theta[6 + T - 1:] = rng.normal(0.0, 0.5, size=M)
    y = theta[:2] - np.finfo(float).eps
    u = theta.copy()
    u[:2] = y + np.log(-np.expm1(-y))
    return dict(cov=cov, init=init, k=k, u=u, theta=theta)


def simulate_numpy(case, R, rng):
    """R exact draws of the event tensor from the model given theta (the chain-binomial of
    doc/lancs_space_model_concept.tex:256-275 with the rates of model_spec.py:232-276): [R,M,T,3]."""
    k, th = case["k"], case["theta"]
    M, T = k.M, k.T
    psi, sig, beta, g0, g1, a0 = th[:6]
    a = a0 + np.concatenate([[0.0], np.cumsum(th[6:6 + T - 1])])
    b = beta * k.log_area_c + sig * th[6 + T - 1:]
    out = np.zeros((R, M, T, 3))
    for r in range(R):
        st = case["init"].copy()
        for t in range(T):
            S, E, I = st[:, 0], st[:, 1], st[:, 2]
            h = I + psi * k.W[t] * (k.Cstar @ (I / k.N))
            assert (h >= 0).all()
            lam = np.exp(a[t] + b) * h / k.N + ms.RATE_FLOOR
            r_ir = math.exp(g0 + g1 * k.weekday_c[t])
            se = rng.binomial(S.astype(np.int64), -np.expm1(-lam))
            ei = rng.binomial(E.astype(np.int64), -np.expm1(-ms.NU))
            ir = rng.binomial(I.astype(np.int64), -np.expm1(-r_ir))
            out[r, :, t, 0], out[r, :, t, 1], out[r, :, t, 2] = se, ei, ir
            st = st + np.stack([-se, se - ei, ei - ir, ir], axis=-1)
    return out


STAT_NAMES = ("sum k_se", "sum k_ei", "sum t*k_se", "sum t*k_ei", "sum k_se^2", "sum k_ei^2",
              "k_se last 2 days", "k_ei last 2 days", "k_se row 0", "k_ei row 0", "sum k_se*k_ei", "min E")


def event_statistics(events, init):
    """[R,M,T,3] -> [R, len(STAT_NAMES)] functions of the latent events the kernels move."""
    ev = np.asarray(events, dtype=float)
    T = ev.shape[2]
    t = np.arange(T)[None, None, :]
    kse, kei = ev[..., 0], ev[..., 1]
    E = init[None, :, None, 1] + np.cumsum(kse - kei, axis=2)
    return np.stack([
        kse.sum((1, 2)), kei.sum((1, 2)), (t * kse).sum((1, 2)), (t * kei).sum((1, 2)),
        (kse ** 2).sum((1, 2)), (kei ** 2).sum((1, 2)), kse[:, :, -2:].sum((1, 2)), kei[:, :, -2:].sum((1, 2)),
        kse[:, 0].sum(1), kei[:, 0].sum(1), (kse * kei).sum((1, 2)), E.min((1, 2))], axis=1)


def paired_z(before, after):
    """z-score of mean(after - before) per statistic over independent replicates."""
    d = after - before
    sd = d.std(axis=0, ddof=1)
    z = np.where(sd > 0, d.mean(axis=0) / np.where(sd > 0, sd, 1.0) * math.sqrt(len(d)), 0.0)
    return z
