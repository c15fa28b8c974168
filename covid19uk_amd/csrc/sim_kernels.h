// Chain-binomial forward simulation: the device counterpart of
// DiscreteTimeStateTransitionModel.sample as used by covid19uk/posterior/predict.py:50-64 --
// for every parameter draw, num_steps days of
//     y_x[m] ~ Binomial(source_x[m], 1 - exp(-rate_x[m] dt)),  state += y . STOICHIOMETRY
// with the rates of model_spec.py:232-276.
//
// One workgroup per draw; the state of the draw lives in LDS for the whole simulation, a day
// is: x = I/N -> F = Cstar x (Cstar is symmetric, so thread m reads column m as row-major
// Cstar[j][m]: coalesced; zero entries of x are skipped, they are the common case early in an
// epidemic) -> rates -> three binomial draws per metapopulation -> state update.  Days are
// inherently sequential; draws are the parallel axis (hundreds to thousands per call).
//
// Random numbers: Philox4x32-10, key = seed, counter = (attempt, 64 + x, s*M + m, draw id); one
// call yields the uniform pair of one attempt, so every (draw, day, metapopulation, transition)
// has its own substream and the result does not depend on scheduling.  The CPU oracle
// (oracle/sim_oracle.py) runs the same protocol.
//
// Binomial(n, p): inversion (BINV) for n min(p,q) < 10, BTRS (Hormann 1993) above.
#pragma once
#include "logprob_kernels.h"
#include "philox.h"

namespace seir {

constexpr uint32_t RS_SIM_BASE = 64;
constexpr int SIM_THREADS = 512;
constexpr int SIM_MAX_ATTEMPTS = 64;
constexpr int SIM_BINV_MAX_X = 200;
constexpr int SIM_DAYS_STAGED = 8;        // days of output staged in LDS per flush

struct SimArgs {
    int n, S, first_draw;
    uint32_t k0, k1;
    const double *par;        // [n][5] psi, sigma_space, beta_area, gamma0, gamma1
    const double *a_path;     // [n][S]
    const double *spatial;    // [n][M]
    const double *W;          // [S]
    const double *wd;         // [S]
    const double *init;       // [n][M][4]
    double *events;           // [n][M][S][3]
};

__device__ __attribute__((noinline)) int sim_binomial(int n, double p, const RngKey &key, uint32_t stream) {
    if (n <= 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    const bool flip = p > 0.5;
    const double pp = flip ? 1.0 - p : p, q = 1.0 - pp;
    const double nd = (double)n;
    int x = -1;
    if (nd * pp < 10.0) {
        const double s = pp / q, a = (nd + 1.0) * s, r0 = exp(nd * log1p(-pp));
        const int xmax = min(n, SIM_BINV_MAX_X);
        for (int att = 0; att < SIM_MAX_ATTEMPTS && x < 0; ++att) {
            double u, v;
            rng_uniform2(key, stream, (uint32_t)att, u, v);
            double r = r0;
            int k = 0;
            while (u > r && k <= xmax) {
                u -= r;
                ++k;
                r *= a / (double)k - s;
            }
            if (k <= xmax) x = k;
        }
        if (x < 0) x = (int)(nd * pp);
    } else {
        const double spq = sqrt(nd * pp * q);
        const double b = 1.15 + 2.53 * spq;
        const double a = -0.0873 + 0.0248 * b + 0.01 * pp;
        const double c = nd * pp + 0.5;
        const double vr = 0.92 - 4.2 / b;
        const double alpha = (2.83 + 5.1 / b) * spq;
        const double m = floor((nd + 1.0) * pp);
        const double lpq = log(pp / q);
        const double h = lfact(m) + lfact(nd - m);
        for (int att = 0; att < SIM_MAX_ATTEMPTS && x < 0; ++att) {
            double u, v;
            rng_uniform2(key, stream, (uint32_t)att, u, v);
            u -= 0.5;
            const double us = 0.5 - fabs(u);
            const double k = floor((2.0 * a / us + b) * u + c);
            if (k < 0.0 || k > nd) continue;
            if (us >= 0.07 && v <= vr) { x = (int)k; break; }
            v = log(v * alpha / (a / (us * us) + b));
            if (v <= h - lfact(k) - lfact(nd - k) + (k - m) * lpq) x = (int)k;
        }
        if (x < 0) x = (int)m;
    }
    return flip ? n - x : x;
}

// dynamic LDS: x [Mp] fp64 | eb [Mp] fp64 | St [3][Mp] int | stage [SIM_DAYS_STAGED][3][Mp] int
__host__ __device__ inline size_t k_simulate_lds_bytes(const Dims &d) {
    return sizeof(double) * 2 * d.Mp + sizeof(int) * 3 * d.Mp + sizeof(int) * SIM_DAYS_STAGED * 3 * d.Mp;
}

__global__ __launch_bounds__(SIM_THREADS) void k_simulate(Dims d, Consts c, SimArgs a) {
    extern __shared__ double lds[];
    double *xs = lds, *ebs = lds + d.Mp;
    int *St = (int *)(ebs + d.Mp);
    int *stage = St + 3 * d.Mp;
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *par = a.par + (size_t)b * 5;
    const double psi = par[0], sig = par[1], beta = par[2], g0 = par[3], g1 = par[4];
    const int M = d.M, S = a.S;
    for (int m = tid; m < d.Mp; m += SIM_THREADS) {
        const double *in = a.init + ((size_t)b * M + m) * 4;
        St[m] = m < M ? (int)in[0] : 0;
        St[d.Mp + m] = m < M ? (int)in[1] : 0;
        St[2 * d.Mp + m] = m < M ? (int)in[2] : 0;
        // exp(beta l_m + sigma s_m) / N_m: constant over the simulated days
        ebs[m] = m < M ? exp(beta * c.la[m] + sig * a.spatial[(size_t)b * M + m]) * c.invN[m] : 0.0;
    }
    __syncthreads();
    RngKey key{a.k0, a.k1, (uint32_t)(a.first_draw + b), 0u};
    const double p_ei = -expm1(-d.nu * d.dt);
    for (int s0 = 0; s0 < S; s0 += SIM_DAYS_STAGED) {
        const int ns = min(SIM_DAYS_STAGED, S - s0);
        for (int ds = 0; ds < ns; ++ds) {
            const int s = s0 + ds;
            for (int m = tid; m < d.Mp; m += SIM_THREADS) xs[m] = m < M ? (double)St[2 * d.Mp + m] * c.invN[m] : 0.0;
            __syncthreads();
            const double ea = exp(a.a_path[(size_t)b * S + s]);
            const double psiW = psi * a.W[s];
            const double p_ir = -expm1(-exp(g0 + g1 * a.wd[s]) * d.dt);
            for (int m = tid; m < M; m += SIM_THREADS) {
                // four independent partial sums: a dependent fp64 FMA chain costs ~32 cycles a link
                double F0 = 0.0, F1 = 0.0, F2 = 0.0, F3 = 0.0;
                const double *col = c.Cstar + m;
                for (int j = 0; j < M; j += 4) {                  // xs and Cstar are zero-padded to Mp
                    const double x0 = xs[j], x1 = xs[j + 1], x2 = xs[j + 2], x3 = xs[j + 3];   // LDS broadcast
                    if ((x0 != 0.0) | (x1 != 0.0) | (x2 != 0.0) | (x3 != 0.0)) {               // uniform branch
                        const double *cj = col + (size_t)j * d.Kp0;
                        F0 = fma(cj[0], x0, F0);
                        F1 = fma(cj[d.Kp0], x1, F1);
                        F2 = fma(cj[2 * (size_t)d.Kp0], x2, F2);
                        F3 = fma(cj[3 * (size_t)d.Kp0], x3, F3);
                    }
                }
                const double F = (F0 + F1) + (F2 + F3);
                const double I = (double)St[2 * d.Mp + m];
                const double lam = ea * ebs[m] * (I + psiW * F) + d.rate_floor;
                const double p_se = -expm1(-lam * d.dt);
                key.sweep = (uint32_t)(s * M + m);
                const int y0 = sim_binomial(St[m], p_se, key, RS_SIM_BASE + 0);
                const int y1 = sim_binomial(St[d.Mp + m], p_ei, key, RS_SIM_BASE + 1);
                const int y2 = sim_binomial(St[2 * d.Mp + m], p_ir, key, RS_SIM_BASE + 2);
                stage[(ds * 3 + 0) * d.Mp + m] = y0;
                stage[(ds * 3 + 1) * d.Mp + m] = y1;
                stage[(ds * 3 + 2) * d.Mp + m] = y2;
            }
            __syncthreads();                                  // every thread has read xs / St of this day
            for (int m = tid; m < M; m += SIM_THREADS) {
                const int y0 = stage[(ds * 3 + 0) * d.Mp + m], y1 = stage[(ds * 3 + 1) * d.Mp + m],
                          y2 = stage[(ds * 3 + 2) * d.Mp + m];
                St[m] -= y0;
                St[d.Mp + m] += y0 - y1;
                St[2 * d.Mp + m] += y1 - y2;
            }
            __syncthreads();
        }
        // flush ns days: events[b][m][s0 + ds][x], runs of ns*3 consecutive doubles per row
        const int run = ns * 3;
        for (int idx = tid; idx < M * run; idx += SIM_THREADS) {
            const int m = idx / run, r = idx - m * run;       // r = ds*3 + x
            a.events[((size_t)b * M + m) * S * 3 + (size_t)s0 * 3 + r] = (double)stage[r * d.Mp + m];
        }
        __syncthreads();
    }
}

}  // namespace seir
