/* CPU oracle (plain C, fp64) for the covid19uk SEIR joint log-probability.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg, never by the product package.
 *
 * PARITY UNPINNED: the reference's arithmetic for this path lives in gemlib
 * (pyproject.toml:15, rev 9fa5e0ff) and tensorflow-probability, neither of
 * which is under /root/reference; the reference ships no golden vector for
 * it.  This file restates the algorithm from the reference's call sites and
 * is pinned against oracle/seir_oracle.py (itself pinned by scipy.stats,
 * mpmath and hand cases, tests/test_oracle.py).
 *
 * Restated reference code:
 *   constants / Cstar etc.     covid19uk/model_spec.py:22-26, 216-230
 *   transition_rate_fn         covid19uk/model_spec.py:232-276
 *   chain-binomial log-prob    covid19uk/model_spec.py:278-285 (gemlib
 *                              DiscreteTimeStateTransitionModel.log_prob),
 *                              doc/lancs_space_model_concept.tex:256-275
 *   compute_state              covid19uk/inference/inference.py:500-510
 *   priors                     covid19uk/model_spec.py:140-198
 *   bijector + joint_log_prob  covid19uk/inference/inference.py:525-557
 *
 * Layout: events[M][T][3] (M-major, as the reference), u[P] with
 * P = 6 + (T-1) + M ordered psi, sigma_space, beta_area, gamma0, gamma1,
 * alpha_0, alpha_t[T-1], spatial_effect[M].
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define NU 0.28
#define RATE_FLOOR 1e-9
#define EPS64 2.220446049250313e-16
#define LOG_2PI 1.8378770664093453

#include "seir_oracle.h"

static double lgam(double x) { int s; return lgamma_r(x, &s); }

static double lbinom(double n, double k) {
    if (k < 0.0 || k > n) return -INFINITY;
    return lgam(n + 1.0) - lgam(k + 1.0) - lgam(n - k + 1.0);
}

/* multiply_no_nan(x, y) */
static double mnn(double x, double y) { return y == 0.0 ? 0.0 : x * y; }

/* formulation 0: as TFP/gemlib evaluate it (p = 1-exp(-r), log p, log(1-p));
 * formulation 1: log(1-p) = -r, log p = log(-expm1(-r)). */
static double ll_term(double n, double k, double r, int stable) {
    double lq, lp;
    if (stable) { lq = -r; lp = log(-expm1(-r)); }
    else { double p = 1.0 - exp(-r); lq = log(1.0 - p); lp = log(p); }
    return mnn(lq, n - k) + mnn(lp, k) + lbinom(n, k);
}

static double softplus(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
static double log_sigmoid(double x) { return -softplus(-x); }

static double priors(const oracle_consts *c, const double *th, double *Qs /* [M] out, may be NULL */) {
    const int M = c->M, T = c->T;
    const double psi = th[0], sig = th[1], beta = th[2], g0 = th[3], g1 = th[4], a0 = th[5];
    const double *at = th + 6, *sp = th + 6 + T - 1;
    double lp = -0.5 * a0 * a0 / 100.0 - log(10.0) - 0.5 * LOG_2PI;
    lp += -0.5 * beta * beta - 0.5 * LOG_2PI;
    lp += 3.0 * log(10.0) - lgam(3.0) + 2.0 * log(psi) - 10.0 * psi;
    for (int j = 0; j < T - 1; ++j)
        lp += -0.5 * (at[j] / 0.005) * (at[j] / 0.005) - log(0.005) - 0.5 * LOG_2PI;
    lp += sig >= 0 ? 0.5 * log(2.0 / M_PI) - log(0.1) - sig * sig / 0.02 : -INFINITY;
    double quad = 0.0;
    for (int i = 0; i < M; ++i) {
        double acc = 0.0;
        for (int j = 0; j < M; ++j) acc += c->car_Q[(size_t)i * M + j] * sp[j];
        if (Qs) Qs[i] = acc;
        quad += sp[i] * acc;
    }
    lp += -0.5 * quad + c->car_half_logdet - 0.5 * M * LOG_2PI;
    lp += -0.5 * g0 * g0 / 1e4 - log(100.0) - 0.5 * LOG_2PI;
    lp += -0.5 * g1 * g1 / 1e4 - log(100.0) - 0.5 * LOG_2PI;
    return lp;
}

/* value (grad == NULL) or value + gradient w.r.t. u.  Returns the joint log-prob. */
double seir_oracle_eval(const oracle_consts *c, const double *u, const double *events,
                        int stable, double *grad) {
    const int M = c->M, T = c->T, P = 6 + T - 1 + M;
    double *th = (double *)malloc(sizeof(double) * P);
    memcpy(th, u, sizeof(double) * P);
    th[0] = softplus(u[0]) + EPS64;
    th[1] = softplus(u[1]) + EPS64;
    const double psi = th[0], sig = th[1], beta = th[2], g0 = th[3], g1 = th[4], a0 = th[5];
    const double *at = th + 6, *sp = th + 6 + T - 1;

    double *S = (double *)malloc(sizeof(double) * M * T * 4);
    double *E = S + (size_t)M * T, *I = E + (size_t)M * T, *X = I + (size_t)M * T;
    double *F = (double *)calloc((size_t)M * T, sizeof(double));
    double *ea = (double *)malloc(sizeof(double) * T), *rir = (double *)malloc(sizeof(double) * T);
    double *eb = (double *)malloc(sizeof(double) * M);
    double *row = (double *)calloc(M, sizeof(double)), *col = (double *)calloc(T, sizeof(double));
    double *gr = (double *)calloc(T, sizeof(double));
    double *GE = grad ? (double *)malloc(sizeof(double) * M * T * 2) : NULL;
    double *GR = grad ? GE + (size_t)M * T : NULL;

    /* state at the start of each day: exclusive cumsum of events @ stoichiometry */
    for (int m = 0; m < M; ++m) {
        double s = c->init_state[m * 4 + 0], e = c->init_state[m * 4 + 1], i = c->init_state[m * 4 + 2];
        for (int t = 0; t < T; ++t) {
            const double *ev = events + ((size_t)m * T + t) * 3;
            S[(size_t)m * T + t] = s; E[(size_t)m * T + t] = e; I[(size_t)m * T + t] = i;
            X[(size_t)m * T + t] = i / c->N[m];
            s -= ev[0]; e += ev[0] - ev[1]; i += ev[1] - ev[2];
        }
    }
    /* F = Cstar @ X   (the matvec of model_spec.py:262 for every t) */
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        double *Fm = F + (size_t)m * T;
        for (int j = 0; j < M; ++j) {
            const double cj = c->Cstar[(size_t)m * M + j];
            const double *Xj = X + (size_t)j * T;
            for (int t = 0; t < T; ++t) Fm[t] += cj * Xj[t];
        }
    }
    double acc = a0;
    for (int t = 0; t < T; ++t) {
        if (t > 0) acc += at[t - 1];
        ea[t] = acc;                                  /* alpha_0 + cumsum(alpha_t)[t-1] */
        rir[t] = exp(g0 + g1 * c->weekday_c[t]);
    }
    for (int m = 0; m < M; ++m) eb[m] = beta * c->log_area_c[m] + sig * sp[m];

    double logL = 0.0, gpsi = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : logL, gpsi)
    for (int m = 0; m < M; ++m) {
        double rsum = 0.0;
        for (int t = 0; t < T; ++t) {
            const size_t q = (size_t)m * T + t;
            const double *ev = events + q * 3;
            const double expeta = exp(ea[t] + eb[m]);
            const double h = I[q] + psi * c->W[t] * F[q];
            const double lam0 = expeta * h / c->N[m];
            const double lam = lam0 + RATE_FLOOR;
            logL += ll_term(S[q], ev[0], lam, stable) + ll_term(E[q], ev[1], NU, stable)
                  + ll_term(I[q], ev[2], rir[t], stable);
            if (grad) {
                const double gl = mnn(1.0 / expm1(lam), ev[0]) - (S[q] - ev[0]);
                const double ge = gl * lam0;
                rsum += ge;
                GE[q] = ge;
                gpsi += gl * expeta * c->W[t] * F[q] / c->N[m];
                GR[q] = mnn(1.0 / expm1(rir[t]), ev[2]) - (I[q] - ev[2]);
            }
        }
        row[m] = rsum;
    }

    if (grad)
        for (int m = 0; m < M; ++m)
            for (int t = 0; t < T; ++t) {
                col[t] += GE[(size_t)m * T + t];
                gr[t] += GR[(size_t)m * T + t];
            }
    double *Qs = grad ? (double *)malloc(sizeof(double) * M) : NULL;
    double lp = priors(c, th, Qs) + logL + log_sigmoid(u[0]) + log_sigmoid(u[1]);

    if (grad) {
        double s0 = exp(log_sigmoid(u[0])), s1 = exp(log_sigmoid(u[1]));
        double gsig = 0.0, gbeta = 0.0, gg0 = 0.0, gg1 = 0.0, ga0 = 0.0;
        for (int m = 0; m < M; ++m) { gsig += sp[m] * row[m]; gbeta += c->log_area_c[m] * row[m]; }
        for (int t = 0; t < T; ++t) {
            gg0 += gr[t] * rir[t]; gg1 += gr[t] * rir[t] * c->weekday_c[t]; ga0 += col[t];
        }
        grad[0] = (gpsi + 2.0 / psi - 10.0) * s0 + (1.0 - s0);
        grad[1] = (gsig - sig / 0.01) * s1 + (1.0 - s1);
        grad[2] = gbeta - beta;
        grad[3] = gg0 - g0 / 1e4;
        grad[4] = gg1 - g1 / 1e4;
        grad[5] = ga0 - a0 / 100.0;
        double rc = 0.0;                      /* reverse cumsum of column sums */
        for (int t = T - 1; t >= 1; --t) {
            rc += col[t];
            grad[6 + t - 1] = rc - at[t - 1] / (0.005 * 0.005);
        }
        for (int m = 0; m < M; ++m) grad[6 + T - 1 + m] = sig * row[m] - Qs[m];
        free(Qs);
    }
    free(th); free(S); free(F); free(ea); free(rir); free(eb); free(row); free(col); free(gr); free(GE);
    return lp;
}

/* flat-argument entry point for ctypes */
double seir_oracle_eval_flat(int M, int T, const double *Cstar, const double *N, const double *W,
                             const double *weekday_c, const double *log_area_c, const double *car_Q,
                             double car_half_logdet, const double *init_state, const double *u,
                             const double *events, int stable, double *grad) {
    oracle_consts c = {M, T, Cstar, N, W, weekday_c, log_area_c, car_Q, car_half_logdet, init_state};
    return seir_oracle_eval(&c, u, events, stable, grad);
}

/* thread count of the OpenMP loops above (cpu_baseline reports it as "cores") */
void seir_oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int seir_oracle_max_threads(void) { return omp_get_max_threads(); }
