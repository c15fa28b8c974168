"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the chain-binomial forward sampler.

PARITY UNPINNED against the reference: the arithmetic lives in gemlib's
`DiscreteTimeStateTransitionModel.sample` (git rev 9fa5e0ff, absent from /root/reference) and in
TensorFlow's binomial sampler; the reference seeds nothing (covid19uk/posterior/predict.py:50-64).
This file defines the build's own draw protocol, which libseirhip's `seir_simulate` follows
bit for bit on the shared Philox stream, and is itself pinned against scipy.stats.binom
(tests/test_simulate.py).

Follows:
  * covid19uk/model_spec.py:232-276   transition rates for absolute day t and state [M,4]
  * covid19uk/model_spec.py:278-285   DiscreteTimeStateTransitionModel(initial_step, num_steps)
  * SURVEY.md appendix A.2            `sample()`: day by day, y_x ~ Binomial(n_source, 1 - exp(-rate_x dt))
  * covid19uk/posterior/predict.py:13-70  predicted_incidence (in-sample / out-of-sample alpha_t)

Draw protocol (per draw d, simulated day s, metapopulation m, transition x in {0: S->E, 1: E->I,
2: I->R}): Philox4x32-10 with key = seed, counter = (attempt, 64 + x, s*M + m, d); one call
gives the two uniforms (u, v) of one attempt of the rejection loop (inversion uses u only).

Binomial(n, p):  n == 0 or p <= 0 -> 0;  p >= 1 -> n;  p > 1/2 -> n - Binomial(n, 1-p);
n p < 10 -> sequential inversion (BINV, Kachitvichyanukul & Schmeiser 1988);
otherwise BTRS (Hormann 1993, transformed rejection with squeeze).
"""
import math

import numpy as np

from . import mcmc_oracle as mo
from . import seir_oracle as so

RS_SIM_BASE = 64
BINV_MAX_MEAN = 10.0
BINV_MAX_X = 200
MAX_ATTEMPTS = 64


def _u(seed, draw, cell, x, attempt):
    a, b = mo.rng_uniform2(seed, draw, cell, RS_SIM_BASE + x, attempt)
    return float(a[0]), float(b[0])


def _lfact(k):
    return math.lgamma(k + 1.0)


def binomial(n, p, uniform):
    """One Binomial(n, p) variate; `uniform(attempt)` returns the attempt-th uniform pair of the cell."""
    n = int(n)
    if n <= 0 or not (p > 0.0):
        return 0
    if p >= 1.0:
        return n
    flip = p > 0.5
    pp = 1.0 - p if flip else p
    q = 1.0 - pp
    if n * pp < BINV_MAX_MEAN:
        s = pp / q
        a = (n + 1) * s
        r0 = math.exp(n * math.log1p(-pp))
        xmax = min(n, BINV_MAX_X)
        x = None
        for attempt in range(MAX_ATTEMPTS):
            u, _ = uniform(attempt)
            r, k = r0, 0
            while u > r and k <= xmax:
                u -= r
                k += 1
                r *= a / k - s
            if k <= xmax:
                x = k
                break
        if x is None:
            x = int(n * pp)
    else:
        spq = math.sqrt(n * pp * q)
        b = 1.15 + 2.53 * spq
        a = -0.0873 + 0.0248 * b + 0.01 * pp
        c = n * pp + 0.5
        vr = 0.92 - 4.2 / b
        alpha = (2.83 + 5.1 / b) * spq
        m = math.floor((n + 1) * pp)
        lpq = math.log(pp / q)
        h = _lfact(m) + _lfact(n - m)
        x = None
        for attempt in range(MAX_ATTEMPTS):
            u, v = uniform(attempt)
            u -= 0.5
            us = 0.5 - abs(u)
            k = math.floor((2.0 * a / us + b) * u + c)
            if k < 0 or k > n:
                continue
            if us >= 0.07 and v <= vr:
                x = k
                break
            v = math.log(v * alpha / (a / (us * us) + b))
            if v <= h - _lfact(k) - _lfact(n - k) + (k - m) * lpq:
                x = k
                break
        if x is None:
            x = int(m)
    return n - x if flip else int(x)


def _prob(rate_dt):
    """1 - exp(-rate dt); a (meaningless) negative rate gives a negative "probability", which the
    binomial sampler treats as 0 -- fp64 overflow of exp included, as on the device."""
    return float("-inf") if rate_dt < -700.0 else -math.expm1(-rate_dt)


def day_rates(state, par, a_t, W_t, wd_t, k: so.ModelConstants):
    """model_spec.py:257-274 for one day: state [M,4] -> (rate_se[M], rate_ei, rate_ir)."""
    psi, sigma, beta, g0, g1 = (float(par[i]) for i in range(5))
    spatial = np.asarray(par[5:], dtype=np.float64)
    I = state[:, 2]
    eta = a_t + beta * k.log_area_c + sigma * spatial
    F = k.Cstar @ (I / k.N)
    lam = np.exp(eta) * (I + psi * W_t * F) / k.N + so.RATE_FLOOR
    return lam, so.NU, math.exp(g0 + g1 * wd_t)


def simulate(k: so.ModelConstants, par, log_baseline, spatial, W, weekday_c, init_state, seed=0, first_draw_id=0):
    """events [n, M, S, 3] for n draws.

    par [n,5] (psi, sigma_space, beta_area, gamma0, gamma1), log_baseline [n,S] (a_t of every
    simulated day), spatial [n,M], W [S], weekday_c [S] (already centred), init_state [n,M,4]."""
    par = np.asarray(par, dtype=np.float64)
    n, S, M = par.shape[0], np.asarray(log_baseline).shape[1], k.M
    out = np.zeros((n, M, S, 3))
    dt = so.TIME_DELTA
    for d in range(n):
        state = np.array(init_state[d], dtype=np.float64)
        th = np.concatenate([par[d], np.asarray(spatial[d], dtype=np.float64)])
        for s in range(S):
            lam, nu, rir = day_rates(state, th, float(log_baseline[d][s]), float(W[s]), float(weekday_c[s]), k)
            p_ei = _prob(nu * dt)
            p_ir = _prob(rir * dt)
            for m in range(M):
                cell = s * M + m
                p_se = _prob(float(lam[m]) * dt)
                for x, (src, p) in enumerate(((0, p_se), (1, p_ei), (2, p_ir))):
                    y = binomial(state[m, src], p,
                                 lambda att, x=x: _u(seed, first_draw_id + d, cell, x, att))
                    out[d, m, s, x] = y
            state = state + out[d, :, s, :] @ so.STOICHIOMETRY
    return out


def log_baseline_path(alpha_0, alpha_t, initial_step, num_steps):
    """a_t for t = initial_step .. initial_step+num_steps-1 with the model's indexing
    (model_spec.py:245-256): alpha_0 at t == 0, else (alpha_0 + cumsum(alpha_t))[clip(t-1, 0, len-1)]."""
    alpha_t = np.asarray(alpha_t, dtype=np.float64)
    out = np.empty(num_steps)
    b = alpha_0 + np.cumsum(alpha_t)
    for s in range(num_steps):
        t = initial_step + s
        if t == 0 or alpha_t.size == 0:
            out[s] = alpha_0
        else:
            out[s] = b[min(max(t - 1, 0), alpha_t.size - 1)]
    return out
