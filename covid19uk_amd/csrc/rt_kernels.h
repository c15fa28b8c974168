// Reproduction number R_it: column sums of the next-generation matrix for every posterior
// draw and day (covid19uk/model_spec.py:302-368; covid19uk/posterior/reproduction_number.py:13-44).
//
//   NGM_t[i][j] = S_it (1 - exp(-rate_ij)) / (1 - exp(-exp(gamma0)))
//   rate_ij     = exp(a_t + beta l_i + sigma s_j) (delta_ij + psi W_t Cstar_ij / N_j) / N_i
//   R_it[t][j]  = sum_i NGM_t[i][j]
// with a_t = alpha_0 (t = 0) or alpha_0 + cumsum(alpha_t)[min(t, T-2)] -- the reference indexes
// b_t with t here, not t-1 as the model does (model_spec.py:336-343 vs :245-256; kept).
// exp(eta) is separable: E_it = exp(a_t + beta l_i)/N_i per (i,t), f_j = exp(sigma s_j) per column.
//
// n T M^2 probability evaluations (7e10 for 2000 draws of UK-380 x 365): compute bound on fp64
// VALU.  Workgroup = 64 destination columns (lane = j) x RT_TT days x all sources i: Cstar[i][j]
// is loaded once per (i, lane) and reused for the RT_TT days held in registers; the per-(i,t)
// factors sit in LDS.  S_it comes from the state scan of the log-prob path (KS = (k_se, S-k_se)).
#pragma once
#include "logprob_kernels.h"

namespace seir {

constexpr int RT_TT = 16;      // days per workgroup

// a_t table with the NGM's indexing, one workgroup per draw; writes w.ea[b][t] = exp(a_t)
__global__ __launch_bounds__(256) void k_rt_tables(Dims d, Work w, const double *__restrict__ theta) {
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *th = theta + (size_t)b * d.P;
    const double a0 = th[5];
    const double *at = th + 6;
    const int n = d.T - 1;                               // alpha_t entries
    const int per = (n + 255) / 256;
    const int lo = tid * per, hi = min(n, lo + per);
    double s = 0.0;
    for (int j = lo; j < hi; ++j) s += at[j];
    double tot;
    double run = a0 + block_excl_scan_256(s, sh, tot);
    // b_t[j] = alpha_0 + cumsum(alpha_t)[j]; a_t = b_t[min(t, n-1)] for t >= 1
    for (int j = lo; j < hi; ++j) {
        run += at[j];
        if (j >= 1) w.ea[(size_t)b * d.Tp + j] = exp(run);                  // day t = j
        if (j == n - 1 && d.T - 1 >= 1) w.ea[(size_t)b * d.Tp + d.T - 1] = exp(run);   // t = T-1 clips to n-1
    }
    if (tid == 0) w.ea[(size_t)b * d.Tp] = exp(a0);
}

// 1 - exp(-x): short series near 0 (rates are ~1e-6), libm otherwise
__device__ __forceinline__ double prob_of_rate(double x) {
    if (fabs(x) < 0.0078125) return x * (1.0 - x * (0.5 - x * (0.16666666666666666 - x * (4.1666666666666664e-2 - x * 8.3333333333333332e-3))));
    return -expm1(-x);
}

__global__ __launch_bounds__(256) void k_rt(Dims d, Consts c, Work w, const double *__restrict__ theta,
                                             double *__restrict__ Rit) {
    extern __shared__ double lds[];                      // E [RT_TT][Mp] | S [RT_TT][Mp] | red [4][RT_TT][64]
    const int b = blockIdx.z, j = blockIdx.x * WAVE + (threadIdx.x & 63), t0 = blockIdx.y * RT_TT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *E = lds, *S = lds + RT_TT * d.Mp, *red = S + RT_TT * d.Mp;
    const double *th = theta + (size_t)b * d.P;
    const double psi = th[0], sig = th[1], beta = th[2], g0 = th[3];
    for (int idx = threadIdx.x; idx < RT_TT * d.Mp; idx += 256) {
        const int tt = idx / d.Mp, i = idx - tt * d.Mp, t = t0 + tt;
        double e = 0.0, sv = 0.0;
        if (i < d.M && t < d.T) {
            e = w.ea[(size_t)b * d.Tp + t] * exp(beta * c.la[i]) * c.invN[i];
            const int2 ks = w.KS[((size_t)b * d.Mp + i) * d.Tp + t];
            sv = (double)(ks.x + ks.y);                  // S at the start of day t
        }
        E[idx] = e;
        S[idx] = sv;
    }
    __syncthreads();
    double pw[RT_TT], acc[RT_TT];
#pragma unroll
    for (int tt = 0; tt < RT_TT; ++tt) { pw[tt] = (t0 + tt < d.T) ? psi * c.W[t0 + tt] : 0.0; acc[tt] = 0.0; }
    const bool jin = j < d.M;
    const double fj = jin ? exp(sig * th[6 + d.T - 1 + j]) : 0.0;
    const double inj = jin ? c.invN[j] : 0.0;
    for (int i = wave; i < d.M; i += 4) {
        const double cij = jin ? c.Cstar[(size_t)i * d.Kp0 + j] * inj : 0.0;
        const double dlt = (i == j) ? 1.0 : 0.0;
#pragma unroll
        for (int tt = 0; tt < RT_TT; ++tt) {
            const double x = E[tt * d.Mp + i] * fj * (dlt + pw[tt] * cij);
            acc[tt] += S[tt * d.Mp + i] * prob_of_rate(x);
        }
    }
#pragma unroll
    for (int tt = 0; tt < RT_TT; ++tt) red[(wave * RT_TT + tt) * WAVE + lane] = acc[tt];
    __syncthreads();
    const double period = 1.0 / (1.0 - exp(-exp(g0)));   // expected infectious period, model_spec.py:361-363
    for (int idx = threadIdx.x; idx < RT_TT * WAVE; idx += 256) {
        const int tt = idx / WAVE, l = idx - tt * WAVE, t = t0 + tt, jj = blockIdx.x * WAVE + l;
        if (t < d.T && jj < d.M) {
            const double v = (red[(0 * RT_TT + tt) * WAVE + l] + red[(1 * RT_TT + tt) * WAVE + l]) +
                             (red[(2 * RT_TT + tt) * WAVE + l] + red[(3 * RT_TT + tt) * WAVE + l]);
            Rit[((size_t)b * d.T + t) * d.M + jj] = v * period;
        }
    }
}

inline size_t k_rt_lds_bytes(const Dims &d) { return sizeof(double) * ((size_t)2 * RT_TT * d.Mp + 4 * RT_TT * WAVE); }

// Within-/between-location infection pressure of the last state of every draw
// (covid19uk/posterior/within_between.py:13-57):
//   within_m  = I_m - psi (I_m/N_m) W colsum(C)_m        = I_m + psi W Cstar_mm x_m
//   between_m = psi W ((C + C^T) x)_m                     = psi W ((Cstar x)_m - Cstar_mm x_m)
// returned as fractions of their sum.  One workgroup per draw, x = I/N in LDS, thread m walks
// column m of the symmetric Cstar (coalesced across m).
__global__ __launch_bounds__(256) void k_within_between(Dims d, Consts c, int n, const double *__restrict__ psi,
                                                         const double *__restrict__ I_last, double W,
                                                         double *__restrict__ within, double *__restrict__ between) {
    extern __shared__ double xs[];                       // [Mp]
    const int b = blockIdx.x;
    for (int m = threadIdx.x; m < d.Mp; m += 256) xs[m] = m < d.M ? I_last[(size_t)b * d.M + m] * c.invN[m] : 0.0;
    __syncthreads();
    const double pw = psi[b] * W;
    for (int m = threadIdx.x; m < d.M; m += 256) {
        double F0 = 0.0, F1 = 0.0, F2 = 0.0, F3 = 0.0;
        const double *col = c.Cstar + m;
        for (int j = 0; j < d.M; j += 4) {               // xs and Cstar are zero-padded to Mp
            const double *cj = col + (size_t)j * d.Kp0;
            // the diagonal (-colsum C) belongs to `within`: leave it out instead of subtracting it back
            F0 = fma(j == m ? 0.0 : cj[0], xs[j], F0);
            F1 = fma(j + 1 == m ? 0.0 : cj[d.Kp0], xs[j + 1], F1);
            F2 = fma(j + 2 == m ? 0.0 : cj[2 * (size_t)d.Kp0], xs[j + 2], F2);
            F3 = fma(j + 3 == m ? 0.0 : cj[3 * (size_t)d.Kp0], xs[j + 3], F3);
        }
        const double self = c.Cstar[(size_t)m * d.Kp0 + m] * xs[m];
        const double wi = I_last[(size_t)b * d.M + m] + pw * self;
        const double be = pw * ((F0 + F1) + (F2 + F3));
        const double tot = wi + be;
        within[(size_t)b * d.M + m] = wi / tot;
        between[(size_t)b * d.M + m] = be / tot;
    }
}

}  // namespace seir
