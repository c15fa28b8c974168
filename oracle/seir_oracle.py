"""CPU oracle for the covid19uk SEIR log-probability (NumPy / SciPy, fp64).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (covid19uk_amd/) may
import this module; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do, and there only as the checker.

PARITY UNPINNED.  The reference's arithmetic for this path lives in the
un-vendored dependency gemlib (pyproject.toml:15, rev 9fa5e0ff) and in
tensorflow-probability (unpinned); neither is under /root/reference nor
installed here, and the reference ships no test, golden vector or fixture for
this path.  This file therefore *restates* the published algorithm from the
reference's own call sites and is pinned by independent implementations only
(scipy.stats, mpmath, hand-derived micro cases, finite differences) -- see
tests/test_oracle.py.

What is restated, with the reference lines each function follows:

* constants                      covid19uk/model_spec.py:22-26
* derived covariates (Cstar, ..) covid19uk/model_spec.py:216-230
* compute_state                  call site covid19uk/inference/inference.py:500-510
                                 (gemlib.util.compute_state: exclusive cumsum of
                                 events @ stoichiometry + initial state)
* transition rates               covid19uk/model_spec.py:232-276
* chain-binomial log-prob        call site covid19uk/model_spec.py:278-285
                                 (gemlib DiscreteTimeStateTransitionModel.log_prob:
                                 per-(m,t) 4x4 Markov matrix, TFP Multinomial log_prob)
                                 and doc/lancs_space_model_concept.tex:256-275
* priors                         covid19uk/model_spec.py:140-198
* bijector / joint_log_prob      covid19uk/inference/inference.py:525-557
"""
from __future__ import annotations

import dataclasses

import numpy as np
from scipy.special import gammaln

# covid19uk/model_spec.py:22-26
STOICHIOMETRY = np.array([[-1, 1, 0, 0], [0, -1, 1, 0], [0, 0, -1, 1]], dtype=np.float64)
TIME_DELTA = 1.0
NU = 0.28
RATE_FLOOR = 1e-9          # covid19uk/model_spec.py:264-266 ("+ 0.000000001")
CAR_RHO = 0.25             # covid19uk/model_spec.py:174
EPS64 = float(np.finfo(np.float64).eps)   # Softplus(low=eps), inference.py:529
LOG_2PI = float(np.log(2.0 * np.pi))

NUM_GLOBAL = 6             # psi, sigma_space, beta_area, gamma0, gamma1, alpha_0


@dataclasses.dataclass
class ModelConstants:
    """Everything in `seir()`'s closure that does not depend on parameters.

    Follows covid19uk/model_spec.py:216-230 (Cstar, W, N, centred weekday,
    centred log-area) and :171-175 (CAR precision).
    """
    Cstar: np.ndarray       # [M,M]
    N: np.ndarray           # [M]
    W: np.ndarray           # [T]
    weekday_c: np.ndarray   # [T]
    log_area_c: np.ndarray  # [M]
    Q: np.ndarray           # [M,M] CAR precision  D_w - rho * W_adj
    half_logdet_Q: float
    initial_state: np.ndarray  # [M,4]

    @property
    def M(self):
        return self.N.shape[0]

    @property
    def T(self):
        return self.W.shape[0]

    @property
    def P(self):
        return NUM_GLOBAL + (self.T - 1) + self.M


def make_constants(C, N, W, weekday, area, adjacency, initial_state) -> ModelConstants:
    C = np.array(C, dtype=np.float64)
    np.fill_diagonal(C, 0.0)                       # model_spec.py:217
    Cstar = C + C.T                                # :218
    np.fill_diagonal(Cstar, -C.sum(axis=-2))       # :219  (minus column sums)
    weekday = np.asarray(weekday, dtype=np.float64)
    weekday_c = weekday - weekday.mean()           # :224-225
    area = np.asarray(area, dtype=np.float64)
    log_area = np.log(area / 100000000.0)          # :229
    log_area_c = log_area - log_area.mean()        # :230
    A = np.asarray(adjacency, dtype=np.float64)
    Q = np.diag(A.sum(axis=-1)) - CAR_RHO * A      # :172-175
    sign, logdet = np.linalg.slogdet(Q)
    if sign <= 0:
        raise ValueError("CAR precision is not positive definite")
    return ModelConstants(
        Cstar=Cstar, N=np.asarray(N, dtype=np.float64).reshape(-1),
        W=np.asarray(W, dtype=np.float64).reshape(-1),
        weekday_c=weekday_c, log_area_c=log_area_c, Q=Q,
        half_logdet_Q=0.5 * float(logdet),
        initial_state=np.asarray(initial_state, dtype=np.float64))


# ---------------------------------------------------------------------------
# state
# ---------------------------------------------------------------------------
def compute_state(initial_state, events, closed=False):
    """State at the START of each day: init + exclusive cumsum(events @ stoich).

    events [M,T,3] -> state [M,T,4]  (closed=True appends the state after the
    last day, giving [M,T+1,4]).
    """
    inc = np.einsum("...tx,xs->...ts", np.asarray(events, dtype=np.float64), STOICHIOMETRY)
    cs = np.cumsum(inc, axis=-2)
    if closed:
        cs = np.concatenate([np.zeros_like(cs[..., :1, :]), cs], axis=-2)
    else:
        cs = np.concatenate([np.zeros_like(cs[..., :1, :]), cs[..., :-1, :]], axis=-2)
    return np.asarray(initial_state, dtype=np.float64)[..., None, :] + cs


# ---------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------
def softplus(x):
    return np.logaddexp(0.0, x)


def log_sigmoid(x):
    return -np.logaddexp(0.0, -x)


def sigmoid(x):
    return np.exp(log_sigmoid(x))


def constrain(u):
    """unconstrained u[P] -> theta[P]  (inference.py:525-538): first two softplus + eps."""
    theta = np.array(u, dtype=np.float64, copy=True)
    theta[..., :2] = softplus(theta[..., :2]) + EPS64
    return theta


def unconstrain(theta):
    u = np.array(theta, dtype=np.float64, copy=True)
    y = u[..., :2] - EPS64
    u[..., :2] = y + np.log(-np.expm1(-y))
    return u


def unpack(theta, M, T):
    """Parameter order of inference.py:541-552."""
    return dict(
        psi=theta[0], sigma_space=theta[1], beta_area=theta[2],
        gamma0=theta[3], gamma1=theta[4], alpha_0=theta[5],
        alpha_t=theta[6:6 + T - 1],
        spatial_effect=theta[6 + T - 1:6 + T - 1 + M])


# ---------------------------------------------------------------------------
# rates  (model_spec.py:232-276, all t at once)
# ---------------------------------------------------------------------------
def transition_rates(par, k: ModelConstants, state):
    """Returns lam[M,T], r_ei (scalar), r_ir[T]."""
    T = k.T
    a = np.empty(T)
    a[0] = par["alpha_0"]
    if T > 1:
        a[1:] = par["alpha_0"] + np.cumsum(par["alpha_t"])       # :242-256
    eta = a[None, :] + (par["beta_area"] * k.log_area_c
                        + par["sigma_space"] * par["spatial_effect"])[:, None]   # :257
    I = state[..., 2]                                             # [M,T]
    F = k.Cstar @ (I / k.N[:, None])                              # :262 matvec for every t
    lam = np.exp(eta) * (I + par["psi"] * k.W[None, :] * F)       # :258-263
    lam = lam / k.N[:, None] + RATE_FLOOR                         # :264-266
    r_ir = np.exp(par["gamma0"] + par["gamma1"] * k.weekday_c)    # :271-274
    return lam, NU, r_ir


# ---------------------------------------------------------------------------
# chain-binomial log-probability
# ---------------------------------------------------------------------------
def _mnn(x, y):
    """tf.math.multiply_no_nan(x, y): 0 where y == 0 even if x is nan/inf."""
    with np.errstate(invalid="ignore"):
        return np.where(y == 0, 0.0, x * y)


def lbinom(n, k):
    """log C(n,k) the way TFP's log_combinations evaluates it (poles -> -inf)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        out = gammaln(n + 1.0) - gammaln(k + 1.0) - gammaln(n - k + 1.0)
    bad = (k < 0) | (k > n)
    return np.where(bad, -np.inf, out)


def _ll_reference(n, k, r):
    """One transition's Multinomial row log-prob as TFP/gemlib evaluate it:
    p = 1-exp(-r*dt); probs = [1-p, p]; counts = [n-k, k]."""
    with np.errstate(invalid="ignore", divide="ignore"):
        p = 1.0 - np.exp(-r * TIME_DELTA)
        q = 1.0 - p
        return _mnn(np.log(q), n - k) + _mnn(np.log(p), k) + lbinom(n, k)


def _ll_stable(n, k, r):
    """Same quantity with log(1-p) = -r and log p = log1mexp(r)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        rr = r * TIME_DELTA
        return _mnn(-rr, n - k) + _mnn(np.log(-np.expm1(-rr)), k) + lbinom(n, k)


def seir_log_prob(par, k: ModelConstants, events, formulation="reference"):
    """sum_{m,t,x} chain-binomial log-pmf (closed form of the Multinomial rows)."""
    events = np.asarray(events, dtype=np.float64)
    state = compute_state(k.initial_state, events)
    lam, r_ei, r_ir = transition_rates(par, k, state)
    ll = _ll_reference if formulation == "reference" else _ll_stable
    tot = ll(state[..., 0], events[..., 0], lam)
    tot = tot + ll(state[..., 1], events[..., 1], np.full_like(lam, r_ei))
    tot = tot + ll(state[..., 2], events[..., 2], np.broadcast_to(r_ir[None, :], lam.shape))
    return float(np.sum(tot))


def seir_log_prob_literal(par, k: ModelConstants, events):
    """Literal restatement of gemlib's discrete-Markov log_prob: build the
    [M,T,4,4] rate matrix, approx_expm, the event matrix with diagonal
    state - sum(events), and TFP Multinomial(total=state, probs=row).log_prob.
    O(16 M T) memory: small cases only."""
    events = np.asarray(events, dtype=np.float64)
    M, T, _ = events.shape
    state = compute_state(k.initial_state, events)           # [M,T,4]
    lam, r_ei, r_ir = transition_rates(par, k, state)
    rates = np.zeros((M, T, 4, 4))
    rates[:, :, 0, 1] = lam
    rates[:, :, 1, 2] = r_ei
    rates[:, :, 2, 3] = r_ir[None, :]
    with np.errstate(invalid="ignore", divide="ignore"):
        total = rates.sum(-1, keepdims=True)
        prob = 1.0 - np.exp(-total * TIME_DELTA)
        mt1 = _mnn(rates / total, prob)
        # multiply_no_nan(x=rates/total, y=prob): zero where prob == 0
        markov = mt1.copy()
        idx = np.arange(4)
        markov[..., idx, idx] = 1.0 - mt1.sum(-1)
        ev = np.zeros((M, T, 4, 4))
        ev[:, :, 0, 1] = events[..., 0]
        ev[:, :, 1, 2] = events[..., 1]
        ev[:, :, 2, 3] = events[..., 2]
        ev[..., idx, idx] = state - ev.sum(-1)
        logp = _mnn(np.log(markov), ev).sum(-1)
        norm = gammaln(state + 1.0) - gammaln(ev + 1.0).sum(-1)
        norm = np.where((ev < 0).any(-1), -np.inf, norm)
        return float(np.sum(logp + norm))


# ---------------------------------------------------------------------------
# priors  (model_spec.py:140-198), TFP log-density formulas
# ---------------------------------------------------------------------------
def _normal_lp(x, scale):
    return -0.5 * (x / scale) ** 2 - np.log(scale) - 0.5 * LOG_2PI


def prior_log_prob(par, k: ModelConstants, literal_car=False):
    lp = _normal_lp(par["alpha_0"], 10.0)                              # :140-144
    lp += _normal_lp(par["beta_area"], 1.0)                            # :146-150
    psi = par["psi"]
    with np.errstate(invalid="ignore", divide="ignore"):
        lp += 3.0 * np.log(10.0) - gammaln(3.0) + 2.0 * np.log(psi) - 10.0 * psi   # :152-156
    lp += np.sum(_normal_lp(par["alpha_t"], 0.005))                    # :158-165
    s = par["sigma_space"]
    lp += (0.5 * np.log(2.0 / np.pi) - np.log(0.1) - s * s / (2 * 0.1 ** 2)
           if s >= 0 else -np.inf)                                     # :167-169
    x = par["spatial_effect"]
    if literal_car:                                                    # :171-181 as written
        cov = np.linalg.inv(k.Q)
        L = np.linalg.cholesky(cov)
        y = np.linalg.solve(L, x)
        lp += -0.5 * y @ y - np.sum(np.log(np.diag(L))) - 0.5 * k.M * LOG_2PI
    else:
        lp += -0.5 * x @ (k.Q @ x) + k.half_logdet_Q - 0.5 * k.M * LOG_2PI
    lp += _normal_lp(par["gamma0"], 100.0) + _normal_lp(par["gamma1"], 100.0)   # :188-198
    return float(lp)


# ---------------------------------------------------------------------------
# the target density  (inference.py:537-557)
# ---------------------------------------------------------------------------
def joint_log_prob(u, events, k: ModelConstants, formulation="reference"):
    u = np.asarray(u, dtype=np.float64)
    theta = constrain(u)
    par = unpack(theta, k.M, k.T)
    lp = prior_log_prob(par, k) + seir_log_prob(par, k, events, formulation)
    return lp + float(np.sum(log_sigmoid(u[:2])))       # inverse_log_det_jacobian


def joint_log_prob_and_grad(u, events, k: ModelConstants):
    """Value and analytic gradient w.r.t. the unconstrained vector u (the
    reference gets the gradient by TF autodiff of joint_log_prob).  Stable
    formulation; derivation in SURVEY.md appendix C."""
    u = np.asarray(u, dtype=np.float64)
    events = np.asarray(events, dtype=np.float64)
    M, T = k.M, k.T
    theta = constrain(u)
    par = unpack(theta, M, T)
    state = compute_state(k.initial_state, events)
    S, E, I = state[..., 0], state[..., 1], state[..., 2]
    kse, kei, kir = events[..., 0], events[..., 1], events[..., 2]

    a = np.empty(T)
    a[0] = par["alpha_0"]
    a[1:] = par["alpha_0"] + np.cumsum(par["alpha_t"])
    b = par["beta_area"] * k.log_area_c + par["sigma_space"] * par["spatial_effect"]
    F = k.Cstar @ (I / k.N[:, None])
    expeta = np.exp(a)[None, :] * np.exp(b)[:, None]
    h = I + par["psi"] * k.W[None, :] * F
    lam0 = expeta * h / k.N[:, None]
    lam = lam0 + RATE_FLOOR
    r_ir = np.exp(par["gamma0"] + par["gamma1"] * k.weekday_c)

    logL = (np.sum(_ll_stable(S, kse, lam)) + np.sum(_ll_stable(E, kei, np.full_like(lam, NU)))
            + np.sum(_ll_stable(I, kir, np.broadcast_to(r_ir[None, :], lam.shape))))

    with np.errstate(invalid="ignore", divide="ignore"):
        g_lam = _mnn(1.0 / np.expm1(lam), kse) - (S - kse)
        g_eta = g_lam * lam0
        g_r = (_mnn(1.0 / np.expm1(r_ir)[None, :], kir) - (I - kir)).sum(0)   # [T]
    row = g_eta.sum(1)           # [M]
    col = g_eta.sum(0)           # [T]

    g = np.zeros(k.P)
    psi, sig = par["psi"], par["sigma_space"]
    g_psi = np.sum(g_lam * expeta * k.W[None, :] * F / k.N[:, None]) + 2.0 / psi - 10.0
    g_sig = np.dot(par["spatial_effect"], row) - sig / 0.01
    g[0] = g_psi * sigmoid(u[0]) + (1.0 - sigmoid(u[0]))
    g[1] = g_sig * sigmoid(u[1]) + (1.0 - sigmoid(u[1]))
    g[2] = np.dot(k.log_area_c, row) - par["beta_area"]
    g[3] = np.sum(g_r * r_ir) - par["gamma0"] / 1.0e4
    g[4] = np.sum(g_r * r_ir * k.weekday_c) - par["gamma1"] / 1.0e4
    g[5] = np.sum(col) - par["alpha_0"] / 100.0
    rc = np.cumsum(col[::-1])[::-1]       # rc[t] = sum_{t' >= t} col[t']
    g[6:6 + T - 1] = rc[1:] - par["alpha_t"] / 0.005 ** 2
    g[6 + T - 1:] = sig * row - k.Q @ par["spatial_effect"]

    lp = prior_log_prob(par, k) + float(logL) + float(np.sum(log_sigmoid(u[:2])))
    return lp, g


# ---------------------------------------------------------------------------
# high-precision reference for micro cases (mpmath), used by tests only
# ---------------------------------------------------------------------------
def joint_log_prob_mp(u, events, k: ModelConstants, dps=50):
    import mpmath as mp
    mp.mp.dps = dps
    M, T = k.M, k.T
    mpf = mp.mpf
    u = [mpf(float(x)) for x in u]
    theta = list(u)
    for i in range(2):
        theta[i] = mp.log1p(mp.e ** u[i]) + mpf(EPS64)
    psi, sig, beta, g0, g1, a0 = theta[:6]
    alpha_t = theta[6:6 + T - 1]
    sp = theta[6 + T - 1:]
    ev = np.asarray(events, dtype=np.float64)
    st = compute_state(k.initial_state, ev)
    a = [a0]
    for j in range(T - 1):
        a.append(a[-1] + alpha_t[j])

    def lgam(x):
        return mp.loggamma(mpf(x))

    def ll(n, kk, r):
        n, kk = int(n), int(kk)
        if kk < 0 or kk > n:
            return mpf("-inf")
        out = lgam(n + 1) - lgam(kk + 1) - lgam(n - kk + 1)
        if n - kk:
            out += (n - kk) * (-r)
        if kk:
            out += kk * mp.log(1 - mp.e ** (-r))
        return out

    tot = mpf(0)
    for t in range(T):
        x = [mpf(float(st[j, t, 2])) / mpf(float(k.N[j])) for j in range(M)]
        r_ir = mp.e ** (g0 + g1 * mpf(float(k.weekday_c[t])))
        for m in range(M):
            Fm = sum(mpf(float(k.Cstar[m, j])) * x[j] for j in range(M))
            eta = a[t] + beta * mpf(float(k.log_area_c[m])) + sig * sp[m]
            lam = mp.e ** eta * (mpf(float(st[m, t, 2])) + psi * mpf(float(k.W[t])) * Fm) \
                / mpf(float(k.N[m])) + mpf("1e-9")
            tot += ll(st[m, t, 0], ev[m, t, 0], lam)
            tot += ll(st[m, t, 1], ev[m, t, 1], mpf(NU))
            tot += ll(st[m, t, 2], ev[m, t, 2], r_ir)
    half_log2pi = mp.log(2 * mp.pi) / 2
    lp = -half_log2pi - mp.log(10) - a0 ** 2 / 200
    lp += -half_log2pi - beta ** 2 / 2
    lp += 3 * mp.log(10) - mp.loggamma(3) + 2 * mp.log(psi) - 10 * psi
    for x in alpha_t:
        lp += -half_log2pi - mp.log(mpf("0.005")) - x ** 2 / (2 * mpf("0.005") ** 2)
    lp += mp.log(2 / mp.pi) / 2 - mp.log(mpf("0.1")) - sig ** 2 / (2 * mpf("0.1") ** 2)
    quad = mpf(0)
    for i in range(M):
        for j in range(M):
            quad += sp[i] * mpf(float(k.Q[i, j])) * sp[j]
    Qm = mp.matrix(k.Q.tolist())
    lp += -quad / 2 + mp.log(mp.det(Qm)) / 2 - mpf(M) * half_log2pi
    for gam in (g0, g1):
        lp += -half_log2pi - mp.log(100) - gam ** 2 / 20000
    for i in range(2):
        lp += -mp.log1p(mp.e ** (-u[i]))
    return tot + lp
