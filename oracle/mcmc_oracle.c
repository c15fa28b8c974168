/* CPU oracle of the Metropolis-within-Gibbs sweep in plain C (fp64, one chain): a restatement of
 * oracle/mcmc_oracle.py, operation for operation, on top of the C density of oracle/seir_oracle.c.
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/libseir_oracle.so, loaded by tests/ and by bench.py's
 * cpu_baseline leg (the same sweep on the host cores with no Python in the
 * timed loop), never by the product package.
 *
 * PARITY UNPINNED, like the Python file it follows: the kernels restated are gemlib's and TFP's, neither
 * under /root/reference; the semantics are the build's own, taken from the
 * reference's call sites:
 *   kernel wiring / config keys   covid19uk/inference/mcmc_kernel_factory.py:14-168
 *   sweep composition             covid19uk/inference/inference.py:86-101,219-228
 *   hmc kwargs, t_range           covid19uk/inference/inference.py:324-339
 *   trace layout                  covid19uk/inference/inference.py:245-282
 * Like the reference it evaluates the FULL joint log-prob for every proposal.
 * Pinned against oracle/mcmc_oracle.py draw by draw (tests/test_mcmc_oracle.py): integer draws, accept
 * flags and event tensors exactly; continuous quantities to rounding (NumPy's vectorised log / sin / cos
 * of the Box-Muller step are not glibc's to the last bit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MC_MMAX 4
#define RS_MOMENTUM 0u
#define RS_HMC_ACCEPT 1u
#define RS_MOVE_BASE 16u
#define MC_INF 0x7FFFFFFF

#include "seir_oracle.h"

typedef struct {
    int is_accepted;
    double target_log_prob, step_size, used_step_size, log_accept_ratio;
} mc_hmc_result;
typedef struct {
    int is_accepted, valid;
    double target_log_prob, log_q_ratio;
    int64_t proposed_delta[4][MC_MMAX];           /* m, t, delta_t, x_star (inference.py:266-273) */
} mc_move_result;
typedef struct {
    mc_hmc_result hmc;
    mc_move_result move[4];                       /* move S->E, move E->I, occult S->E, occult E->I: the LAST scan's */
} mc_sweep_result;

typedef struct {
    oracle_consts c;
    int M, T, P;
    int dmax, nmax, m, occult_nmax, n_scans, t_lo, t_hi, L;
    uint64_t seed;
    uint32_t chain, sweep;
    unsigned disabled;                            /* bit 0 hmc, 1..4 the four event kernels */
    double eps, *var;
    int adapt_step, adapt_mass, n_adapt;
    double target;
    double da_err, da_step, da_logavg, da_mu;
    double rv_n, *rv_mean, *rv_m2;
    double *u, *events, *ev_new, *st;             /* st: closed state [M][T+1][4] of `events` */
    int st_valid;
    double logp;
    long n_evals;
    double *q, *p, *g, *z;
    int *idx;                                     /* scratch [max(M, T)] */
} mc_chain;

/* ---- Philox4x32-10, counter (idx, stream, sweep, chain), key = seed (oracle/mcmc_oracle.py:38-83) ---- */
static void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint64_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = M0 * c0, p1 = M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static double u01(uint32_t hi, uint32_t lo) {
    const uint64_t x = (((uint64_t)hi << 32) | lo) >> 12;
    return ((double)x + 0.5) * 2.220446049250313e-16;
}
static void uniform2(const mc_chain *ch, uint32_t stream, uint32_t idx, double *a, double *b) {
    uint32_t r[4];
    philox(idx, stream, ch->sweep, ch->chain, (uint32_t)(ch->seed & 0xFFFFFFFFu), (uint32_t)(ch->seed >> 32), r);
    *a = u01(r[0], r[1]);
    *b = u01(r[2], r[3]);
}
static int rng_index(double u, int n) { const int i = (int)(u * n); return i < n - 1 ? i : n - 1; }

static double lp_value(mc_chain *ch, const double *u, const double *events) {
    ch->n_evals += 1;
    return seir_oracle_eval(&ch->c, u, events, 1, NULL);
}
static double lp_grad(mc_chain *ch, const double *u, double *g) {
    ch->n_evals += 1;
    return seir_oracle_eval(&ch->c, u, ch->events, 1, g);
}

/* ---- HMC with dual averaging and the running variance (mcmc_oracle.py: hmc_step) ---- */
static void hmc_step(mc_chain *ch, mc_hmc_result *out) {
    const int P = ch->P;
    const double eps = ch->eps, *var = ch->var;
    double *q = ch->q, *p = ch->p, *g = ch->g, *z = ch->z;
    memcpy(q, ch->u, sizeof(double) * P);
    const double lp0 = lp_grad(ch, q, g);
    for (int i = 0; i < (P + 1) / 2; ++i) {
        double u1, u2;
        uniform2(ch, RS_MOMENTUM, (uint32_t)i, &u1, &u2);
        const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586 * u2;
        z[2 * i] = rad * cos(ang);
        z[2 * i + 1] = rad * sin(ang);
    }
    double k0 = 0.0;
    for (int i = 0; i < P; ++i) { p[i] = z[i] / sqrt(var[i]); k0 += 0.5 * var[i] * p[i] * p[i]; }
    for (int i = 0; i < P; ++i) { p[i] += 0.5 * eps * g[i]; q[i] += eps * var[i] * p[i]; }
    for (int s = 1; s < ch->L; ++s) {
        (void)lp_grad(ch, q, g);                   /* (at the trajectory's position q; the chain's u moves on acceptance only) */
        for (int i = 0; i < P; ++i) { p[i] += eps * g[i]; q[i] += eps * var[i] * p[i]; }
    }
    const double lp1 = lp_grad(ch, q, g);
    double k1 = 0.0;
    for (int i = 0; i < P; ++i) { p[i] += 0.5 * eps * g[i]; k1 += 0.5 * var[i] * p[i] * p[i]; }
    const double log_ratio = (lp1 - lp0) - (k1 - k0);
    double ua, ub;
    uniform2(ch, RS_HMC_ACCEPT, 0u, &ua, &ub);
    const int acc = (log(ua) < log_ratio) && !(ch->disabled & 1u);      /* NaN -> reject */
    if (acc) { memcpy(ch->u, q, sizeof(double) * P); ch->logp = lp1; }
    else ch->logp = lp0;
    if (ch->adapt_step) {
        const double a = isfinite(log_ratio) ? fmin(1.0, exp(log_ratio)) : 0.0;
        const double prev = ch->da_step, n = prev + 1.0;
        ch->da_err += ch->target - a;
        const double log_step = ch->da_mu - ch->da_err * sqrt(n) / ((n + 10.0) * 0.05);
        const double eta = pow(n, -0.75);
        ch->da_logavg = eta * log_step + (1.0 - eta) * ch->da_logavg;
        ch->da_step = n;
        if (prev <= ch->n_adapt) ch->eps = prev < ch->n_adapt ? exp(log_step) : exp(ch->da_logavg);
    }
    if (ch->adapt_mass) {
        const double n1 = ch->rv_n + 1.0;
        for (int i = 0; i < P; ++i) {
            const double dlt = ch->u[i] - ch->rv_mean[i];
            ch->rv_mean[i] += dlt / n1;
            ch->rv_m2[i] += dlt * (ch->u[i] - ch->rv_mean[i]);
            ch->var[i] = ch->rv_m2[i] / n1;
        }
        ch->rv_n = n1;
    }
    out->is_accepted = acc;
    out->target_log_prob = ch->logp;
    out->step_size = ch->eps;                     /* the NEXT step's, as trace_results_fn reads it (inference.py:255-261) */
    out->used_step_size = eps;
    out->log_accept_ratio = log_ratio;
}

/* ---- event updates ---- */
static void closed_state(mc_chain *ch) {
    if (ch->st_valid) return;
    const int M = ch->M, T = ch->T;
    for (int m = 0; m < M; ++m) {
        double S = ch->c.init_state[m * 4 + 0], E = ch->c.init_state[m * 4 + 1], I = ch->c.init_state[m * 4 + 2],
               R = ch->c.init_state[m * 4 + 3];
        double *st = ch->st + (size_t)m * (T + 1) * 4;
        for (int t = 0; t <= T; ++t) {
            st[t * 4 + 0] = S; st[t * 4 + 1] = E; st[t * 4 + 2] = I; st[t * 4 + 3] = R;
            if (t < T) {
                const double *e = ch->events + ((size_t)m * T + t) * 3;
                S -= e[0]; E += e[0] - e[1]; I += e[1] - e[2]; R += e[2];
            }
        }
    }
    ch->st_valid = 1;
}
static int st_min(const mc_chain *ch, int m, int a, int b, int comp) {      /* min over days [a, b) of compartment comp */
    const double *st = ch->st + (size_t)m * (ch->T + 1) * 4;
    double v = st[a * 4 + comp];
    for (int t = a + 1; t < b; ++t) if (st[t * 4 + comp] < v) v = st[t * 4 + comp];
    return (int)v;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static int mh(mc_chain *ch, int valid, double logq, double logu) {
    int acc = 0;
    double lp_new = 0.0;
    if (valid) {
        lp_new = lp_value(ch, ch->u, ch->ev_new);
        acc = logu < (lp_new - ch->logp) + logq;
    }
    if (acc) {
        double *t = ch->events; ch->events = ch->ev_new; ch->ev_new = t;
        ch->logp = lp_new;
        ch->st_valid = 0;
    }
    return acc;
}

static void event_time_move(mc_chain *ch, int tgt, int scan, int slot, mc_move_result *out) {
    const int M = ch->M, T = ch->T;
    const uint32_t stream = RS_MOVE_BASE + (uint32_t)(scan * 4 + slot);
    double ua, ub;
    uniform2(ch, stream, 15u, &ua, &ub);
    const double logu = log(ua);
    closed_state(ch);
    const double *ev = ch->events;
    int *hot = ch->idx, H = 0;
    for (int m = 0; m < M; ++m) {
        double tot = 0.0;
        for (int t = 0; t < T; ++t) tot += ev[((size_t)m * T + t) * 3 + tgt];
        if (tot > 0.0) hot[H++] = m;
    }
    const int nsel = imin(imin(ch->m, MC_MMAX), H);
    int chosen[MC_MMAX], nch = 0, valid = 1;
    double logq = 0.0;
    memcpy(ch->ev_new, ev, sizeof(double) * (size_t)M * T * 3);
    memset(out->proposed_delta, 0, sizeof(out->proposed_delta));
    int *days = (int *)malloc(sizeof(int) * T);
    for (int j = 0; j < nsel; ++j) {
        double u_m, u_t, u_d, u_x;
        uniform2(ch, stream, (uint32_t)(2 * j), &u_m, &u_t);
        uniform2(ch, stream, (uint32_t)(2 * j + 1), &u_d, &u_x);
        int pos = rng_index(u_m, H - j);
        for (int a = 0; a < nch; ++a) if (pos >= chosen[a]) pos += 1;       /* chosen is kept sorted */
        int ins = nch;
        while (ins > 0 && chosen[ins - 1] > pos) { chosen[ins] = chosen[ins - 1]; --ins; }
        chosen[ins] = pos; ++nch;
        const int m = hot[pos];
        const double *K = ev + (size_t)m * T * 3 + tgt;
        int D = 0;
        for (int t = 0; t < T; ++t) if (K[t * 3] > 0.0) days[D++] = t;
        const int t = days[rng_index(u_t, D)];
        const int v = rng_index(u_d, 2 * ch->dmax);
        const int delta = v < ch->dmax ? v - ch->dmax : v - ch->dmax + 1;
        const int t2 = t + delta;
        if (t2 < 0 || t2 >= T) {
            valid = 0;
            out->proposed_delta[0][j] = m; out->proposed_delta[1][j] = t; out->proposed_delta[2][j] = delta;
            out->proposed_delta[3][j] = 0;
            continue;
        }
        const int lo = imin(t, t2), hi = imax(t, t2);
        const int src = tgt, dst = tgt + 1;
        const int dec = delta > 0 ? dst : src, inc = delta > 0 ? src : dst;
        const int min_dec = dec == 0 ? MC_INF : st_min(ch, m, lo + 1, hi + 1, dec);
        const int min_inc = inc == 0 ? MC_INF : st_min(ch, m, lo + 1, hi + 1, inc);
        const int kt = (int)K[t * 3], kt2 = (int)K[t2 * 3];
        const int xmax = imax(0, imin(imin(ch->nmax, kt), min_dec));
        const int x = rng_index(u_x, xmax + 1);
        const int Dn = D - ((x > 0 && x == kt) ? 1 : 0) + ((x > 0 && kt2 == 0) ? 1 : 0);
        const int binc = inc == 0 ? MC_INF : min_inc + x;
        const int xmax_r = imax(0, imin(imin(ch->nmax, kt2 + x), binc));
        /* a null sub-move (x == 0) is its own reverse: no correction */
        if (x > 0) logq += (-log((double)Dn) - log((double)(xmax_r + 1))) - (-log((double)D) - log((double)(xmax + 1)));
        ch->ev_new[((size_t)m * T + t) * 3 + tgt] -= x;
        ch->ev_new[((size_t)m * T + t2) * 3 + tgt] += x;
        out->proposed_delta[0][j] = m; out->proposed_delta[1][j] = t; out->proposed_delta[2][j] = delta;
        out->proposed_delta[3][j] = x;
    }
    free(days);
    const int enabled = !(ch->disabled & (1u << (1 + tgt)));
    out->is_accepted = mh(ch, valid && enabled, logq, logu);
    out->target_log_prob = ch->logp;
    out->log_q_ratio = logq;
    out->valid = valid;
}

static void occult_move(mc_chain *ch, int tgt, int scan, int slot, mc_move_result *out) {
    const int M = ch->M, T = ch->T, lo_r = ch->t_lo, hi_r = ch->t_hi, R = hi_r - lo_r, nmax = ch->occult_nmax;
    const uint32_t stream = RS_MOVE_BASE + (uint32_t)(scan * 4 + slot);
    double ua, ub, u_br, u_m, u_t, u_x;
    uniform2(ch, stream, 15u, &ua, &ub);
    const double logu = log(ua);
    uniform2(ch, stream, 0u, &u_br, &u_m);
    uniform2(ch, stream, 1u, &u_t, &u_x);
    closed_state(ch);
    const double *ev = ch->events;
    int *hotrows = ch->idx, Hd = 0;
    double *rng_tot = ch->z;                         /* scratch [>= M]: P >= M */
    for (int m = 0; m < M; ++m) {
        double tot = 0.0;
        for (int t = lo_r; t < hi_r; ++t) tot += ev[((size_t)m * T + t) * 3 + tgt];
        rng_tot[m] = tot;
        if (tot > 0.0) hotrows[Hd++] = m;
    }
    const int is_del = (u_br < 0.5) && Hd > 0;
    const int src = tgt, dst = tgt + 1;
    int m, t = 0;
    if (!is_del) { m = rng_index(u_m, M); t = lo_r + rng_index(u_t, R); }
    else m = hotrows[rng_index(u_m, Hd)];
    const double *K = ev + (size_t)m * T * 3 + tgt;
    int Dm = 0, *hotdays = (int *)malloc(sizeof(int) * (size_t)(R > 0 ? R : 1));
    for (int d = lo_r; d < hi_r; ++d) if (K[d * 3] > 0.0) hotdays[Dm++] = d - lo_r;
    if (is_del) t = lo_r + hotdays[rng_index(u_t, Dm)];
    free(hotdays);
    const int rt_m = (int)rng_tot[m];
    const int min_src = src == 0 ? MC_INF : st_min(ch, m, t + 1, T + 1, src);
    const int min_dst = st_min(ch, m, t + 1, T + 1, dst);
    const int kt = (int)K[t * 3];
    const double l2 = log(2.0), lM = log((double)M), lR = log((double)R);
    memcpy(ch->ev_new, ev, sizeof(double) * (size_t)M * T * 3);
    double qf, qr;
    int x;
    if (!is_del) {
        const int xmax = imax(0, imin(nmax, min_src));
        x = rng_index(u_x, xmax + 1);
        qf = (Hd > 0 ? -l2 : 0.0) - lM - lR - log((double)(xmax + 1));
        const int Hd2 = Hd + ((rt_m == 0 && x > 0) ? 1 : 0), Dm2 = Dm + ((kt == 0 && x > 0) ? 1 : 0);
        const int xmax_r = imax(0, imin(imin(nmax, kt + x), min_dst + x));
        qr = (Hd2 > 0 && kt + x > 0) ? (-l2 - log((double)Hd2) - log((double)Dm2) - log((double)(xmax_r + 1))) : -INFINITY;
        ch->ev_new[((size_t)m * T + t) * 3 + tgt] += x;
    } else {
        const int xmax = imax(0, imin(imin(nmax, kt), min_dst));
        x = rng_index(u_x, xmax + 1);
        qf = -l2 - log((double)Hd) - log((double)Dm) - log((double)(xmax + 1));
        const int Hd2 = Hd - ((x > 0 && rt_m == x) ? 1 : 0);
        const int bs = src == 0 ? MC_INF : min_src + x;
        const int xmax_r = imax(0, imin(nmax, bs));
        qr = (Hd2 > 0 ? -l2 : 0.0) - lM - lR - log((double)(xmax_r + 1));
        ch->ev_new[((size_t)m * T + t) * 3 + tgt] -= x;
    }
    const int enabled = !(ch->disabled & (1u << (3 + tgt)));
    out->is_accepted = mh(ch, enabled, qr - qf, logu);
    out->target_log_prob = ch->logp;
    out->log_q_ratio = qr - qf;
    out->valid = 1;
    memset(out->proposed_delta, 0, sizeof(out->proposed_delta));
    out->proposed_delta[0][0] = m; out->proposed_delta[1][0] = t; out->proposed_delta[2][0] = is_del ? -1 : 1;
    out->proposed_delta[3][0] = x;
}

/* ---- public (ctypes) ---- */
mc_chain *mcmc_oracle_create(int M, int T, const double *Cstar, const double *N, const double *W, const double *weekday_c,
                             const double *log_area_c, const double *car_Q, double car_half_logdet, const double *init_state,
                             const double *u, const double *events, const int *cfg /* dmax, nmax, m, occult_nmax, n_scans,
                             t_lo, t_hi, L, disabled mask */, uint64_t seed, uint32_t chain_id) {
    mc_chain *ch = (mc_chain *)calloc(1, sizeof(mc_chain));
    const int P = 6 + T - 1 + M;
    const size_t ne = (size_t)M * T * 3;
    ch->M = M; ch->T = T; ch->P = P;
    /* the constants are copied: the caller's arrays need not outlive the call */
    double *cs = (double *)malloc(sizeof(double) * ((size_t)2 * M * M + 3 * M + 2 * T + 4 * M));
    double *q = cs;
#define MC_COPY(dst, src, n) do { memcpy(q, (src), sizeof(double) * (n)); (dst) = q; q += (n); } while (0)
    MC_COPY(ch->c.Cstar, Cstar, (size_t)M * M); MC_COPY(ch->c.car_Q, car_Q, (size_t)M * M);
    MC_COPY(ch->c.N, N, M); MC_COPY(ch->c.log_area_c, log_area_c, M);
    MC_COPY(ch->c.W, W, T); MC_COPY(ch->c.weekday_c, weekday_c, T);
    MC_COPY(ch->c.init_state, init_state, 4 * (size_t)M);
#undef MC_COPY
    ch->c.M = M; ch->c.T = T; ch->c.car_half_logdet = car_half_logdet;
    ch->dmax = cfg[0]; ch->nmax = cfg[1]; ch->m = cfg[2]; ch->occult_nmax = cfg[3]; ch->n_scans = cfg[4];
    ch->t_lo = cfg[5]; ch->t_hi = cfg[6]; ch->L = cfg[7]; ch->disabled = (unsigned)cfg[8];
    ch->seed = seed; ch->chain = chain_id; ch->sweep = 0;
    ch->eps = 0.1; ch->target = 0.75;
    ch->da_mu = log(10.0 * ch->eps);
    ch->var = (double *)malloc(sizeof(double) * P);
    ch->rv_mean = (double *)calloc(P, sizeof(double)); ch->rv_m2 = (double *)calloc(P, sizeof(double));
    for (int i = 0; i < P; ++i) ch->var[i] = 1.0;
    ch->u = (double *)malloc(sizeof(double) * P); memcpy(ch->u, u, sizeof(double) * P);
    ch->events = (double *)malloc(sizeof(double) * ne); memcpy(ch->events, events, sizeof(double) * ne);
    ch->ev_new = (double *)malloc(sizeof(double) * ne);
    ch->st = (double *)malloc(sizeof(double) * (size_t)M * (T + 1) * 4);
    ch->q = (double *)malloc(sizeof(double) * P); ch->p = (double *)malloc(sizeof(double) * P);
    ch->g = (double *)malloc(sizeof(double) * P); ch->z = (double *)malloc(sizeof(double) * (P + 2));
    ch->idx = (int *)malloc(sizeof(int) * (size_t)(M > T ? M : T));
    ch->logp = seir_oracle_eval(&ch->c, ch->u, ch->events, 1, NULL);
    return ch;
}
void mcmc_oracle_destroy(mc_chain *ch) {
    if (!ch) return;
    free((void *)ch->c.Cstar);
    free(ch->var); free(ch->rv_mean); free(ch->rv_m2); free(ch->u); free(ch->events); free(ch->ev_new); free(ch->st);
    free(ch->q); free(ch->p); free(ch->g); free(ch->z); free(ch->idx);
    free(ch);
}
void mcmc_oracle_set_eps(mc_chain *ch, double eps) { ch->eps = eps; }
void mcmc_oracle_set_adaptation(mc_chain *ch, int adapt_step, int adapt_mass, int n_adapt, double target, double rv_count,
                                const double *rv_mean, const double *rv_var) {
    ch->adapt_step = adapt_step; ch->adapt_mass = adapt_mass; ch->n_adapt = n_adapt; ch->target = target;
    ch->da_err = ch->da_step = ch->da_logavg = 0.0;
    ch->da_mu = log(10.0 * ch->eps);
    if (adapt_mass) {
        ch->rv_n = rv_count;
        for (int i = 0; i < ch->P; ++i) { ch->rv_mean[i] = rv_mean[i]; ch->rv_m2[i] = rv_var[i] * rv_count; ch->var[i] = rv_var[i]; }
    }
}
/* one posterior draw: HMC on u | events, then n_scans x [move S->E, move E->I, occult S->E, occult E->I] */
void mcmc_oracle_sweep(mc_chain *ch, mc_sweep_result *out) {
    if (ch->disabled & 1u) {
        out->hmc.is_accepted = 0; out->hmc.target_log_prob = ch->logp; out->hmc.step_size = out->hmc.used_step_size = ch->eps;
        out->hmc.log_accept_ratio = NAN;
    } else {
        hmc_step(ch, &out->hmc);
    }
    for (int scan = 0; scan < ch->n_scans; ++scan) {
        event_time_move(ch, 0, scan, 0, &out->move[0]);
        event_time_move(ch, 1, scan, 1, &out->move[1]);
        occult_move(ch, 0, scan, 2, &out->move[2]);
        occult_move(ch, 1, scan, 3, &out->move[3]);
    }
    ch->sweep += 1;
}
/* n sweeps with nothing but C in the loop (bench.py's cpu_baseline) */
void mcmc_oracle_run(mc_chain *ch, int n) {
    mc_sweep_result r;
    for (int i = 0; i < n; ++i) mcmc_oracle_sweep(ch, &r);
}
void mcmc_oracle_get_state(const mc_chain *ch, double *u, double *events, double *scal /* logp, eps, n_evals */) {
    if (u) memcpy(u, ch->u, sizeof(double) * ch->P);
    if (events) memcpy(events, ch->events, sizeof(double) * (size_t)ch->M * ch->T * 3);
    if (scal) { scal[0] = ch->logp; scal[1] = ch->eps; scal[2] = (double)ch->n_evals; }
}
void mcmc_oracle_get_variance(const mc_chain *ch, double *var) { memcpy(var, ch->var, sizeof(double) * ch->P); }
