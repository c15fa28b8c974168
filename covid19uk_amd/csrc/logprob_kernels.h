// Kernels of the joint log-probability evaluation
//   joint_log_prob(u, events)            covid19uk/inference/inference.py:537-557
//   DiscreteTimeStateTransitionModel.log_prob  (call site covid19uk/model_spec.py:278-285)
//   transition_rate_fn                   covid19uk/model_spec.py:232-276
// split by what each stage depends on:
//   k_scan      events only  state prefix-sum over T, binomial coefficients, E->I term,
//                            X = I/N, per-day I->R sufficient statistics
//   k_colreduce events only  folds k_scan's per-block partials (exact integer sums)
//   k_gemm      events only  F = Cstar . X  (the matvec of model_spec.py:262 for all T), fp64 MFMA
//   k_params    parameters   softplus, exp(alpha_0 + cumsum alpha_t), exp(beta l + sigma s)/N, priors
//   k_se        both         S->E chain-binomial term (+ d/d eta row/column sums, d/d psi)
//   k_finish    both         reduction, I->R term, gradient assembly
//
// Per-chain workspace (HBM), rows padded to Mp = ceil16(M), days to Tp = ceil64(T), pads zero:
//   API path     Xn fp64 (I/N), KS int2 (k_se, S-k_se), F fp64
//   sampler path int32 planes K[3] (events), St[3] (S,E,I at start of day), F fp64
#pragma once
#include "device_math.h"

namespace seir {

constexpr int SCAN_ROWS = 8;    // rows per k_scan workgroup (2 per wave)
constexpr int SE_TM = 16;       // k_se tile: 16 rows x 64 days per workgroup (4 rows per wave)
constexpr int NSCAL = 16;       // per-chain scalar block
enum { SC_PSI = 0, SC_SIG, SC_BETA, SC_G0, SC_G1, SC_A0, SC_S0, SC_S1, SC_PRIOR, SC_JAC };

struct Dims {
    int M, T, Mp, Tp, Kp, P, Pp;
    int nrb_scan;           // row blocks of k_scan
    int nmt, ntc;           // k_se tiles: Mp/16 row tiles, Tp/64 day chunks
    double nu, dt, rate_floor, car_half_logdet;
};

struct Consts {
    const double *Cstar;   // [Mp][Kp]
    const double *N, *invN, *la;   // [Mp]
    const double *W, *wd;          // [Tp]
    const double *init;            // [Mp][4]
    const int *Qrow, *Qcol;        // CSR of car_Q
    const double *Qval;
};

struct Work {
    double *Xn, *F;        // [B][Mp][Tp]
    int2 *KS;              // [B][Mp][Tp]
    int *K[3], *St[3];     // sampler planes [B][Mp][Tp] (null on a plain context)
    int *rowtot;           // [B][2][Mp] row totals of S->E / E->I events (sampler)
    double *rowconst;      // [B][Mp]
    double *colIR;         // [B][nrb_scan][Tp][2]
    double *Kir, *Dir;     // [B][Tp]  sum_m k_ir, sum_m (I - k_ir)
    double *constsum;      // [B]      sum of binomial coefficients + E->I term
    double *ea;            // [B][Tp]   exp(alpha_0 + cumsum(alpha_t)[t-1])
    double *eb;            // [B][Mp]   exp(beta*l_m + sigma*s_m) / N_m
    double *rir;           // [B][Tp]   exp(gamma0 + gamma1*wd_t)
    double *scal;          // [B][NSCAL]
    double *Qs;            // [B][Mp]   car_Q . spatial_effect
    double *Lpart, *Ppart; // [B][nmt*ntc]
    double *Kpart;         // [B][nmt][Tp]
    double *Rpart;         // [B][ntc][Mp]
};

// ---------------------------------------------------------------------------
// k_scan: one wave per row (m); lanes over days in 64-day chunks with a carry.
// State at the START of day t (gemlib compute_state, call site inference.py:500-510).
// SRC 0: events fp64 [B][M][T][3] (reference layout); writes Xn, KS.
// SRC 1: events from the sampler's int32 planes; writes St planes, Xn, rowtot.
// ---------------------------------------------------------------------------
template <int SRC>
__global__ __launch_bounds__(256) void k_scan(Dims d, Consts c, Work w, const double *__restrict__ events) {
    extern __shared__ double lds[];                 // [4][Tp][2]
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *mycol = lds + (size_t)wave * d.Tp * 2;
    for (int i = lane; i < d.Tp * 2; i += WAVE) mycol[i] = 0.0;

    const double r_ei = d.nu * d.dt;
    const double L_ei = log1mexp(r_ei);
    const int nch = d.Tp / WAVE;
    constexpr int RPW = SCAN_ROWS / 4;
    for (int r = 0; r < RPW; ++r) {
        const int m = blockIdx.x * SCAN_ROWS + wave * RPW + r;
        if (m >= d.M) break;
        const double S0 = c.init[m * 4 + 0], E0 = c.init[m * 4 + 1], I0 = c.init[m * 4 + 2];
        const double invN = c.invN[m];
        const size_t rowoff = ((size_t)b * d.Mp + m) * d.Tp;
        const double *ev = SRC == 0 ? events + ((size_t)b * d.M + m) * d.T * 3 : nullptr;
        double cse = 0.0, cei = 0.0, cir = 0.0, rc = 0.0;
        for (int ch = 0; ch < nch; ++ch) {
            const int t = ch * WAVE + lane;
            const bool valid = t < d.T;
            double kse = 0.0, kei = 0.0, kir = 0.0;
            if (SRC == 0) {
                if (valid) {
                    kse = ev[(size_t)t * 3 + 0];
                    kei = ev[(size_t)t * 3 + 1];
                    kir = ev[(size_t)t * 3 + 2];
                }
            } else {
                kse = (double)w.K[0][rowoff + t];
                kei = (double)w.K[1][rowoff + t];
                kir = (double)w.K[2][rowoff + t];
            }
            const double ise = wave_incl_scan(kse, lane), iei = wave_incl_scan(kei, lane),
                         iir = wave_incl_scan(kir, lane);
            const double xse = cse + ise - kse, xei = cei + iei - kei, xir = cir + iir - kir;
            const double S = S0 - xse, E = E0 + xse - xei, I = I0 + xei - xir;
            w.Xn[rowoff + t] = valid ? I * invN : 0.0;
            if (SRC == 0) {
                w.KS[rowoff + t] = valid ? make_int2((int)kse, (int)(S - kse)) : make_int2(0, 0);
            } else {
                w.St[0][rowoff + t] = valid ? (int)S : 0;
                w.St[1][rowoff + t] = valid ? (int)E : 0;
                w.St[2][rowoff + t] = valid ? (int)I : 0;
            }
            if (valid) {
                rc += lbinom(S, kse) + lbinom(E, kei) + lbinom(I, kir);
                rc += kei * L_ei - (E - kei) * r_ei;
                mycol[t * 2 + 0] += kir;
                mycol[t * 2 + 1] += I - kir;
            }
            cse += __shfl(ise, WAVE - 1, WAVE);
            cei += __shfl(iei, WAVE - 1, WAVE);
            cir += __shfl(iir, WAVE - 1, WAVE);
        }
        rc = wave_sum(rc);
        if (lane == 0) {
            w.rowconst[(size_t)b * d.Mp + m] = rc;
            if (SRC == 1) {
                w.rowtot[((size_t)b * 2 + 0) * d.Mp + m] = (int)cse;
                w.rowtot[((size_t)b * 2 + 1) * d.Mp + m] = (int)cei;
            }
        }
    }
    __syncthreads();
    double *out = w.colIR + ((size_t)b * d.nrb_scan + blockIdx.x) * d.Tp * 2;
    const int n = d.Tp * 2;
    for (int i = threadIdx.x; i < n; i += 256) out[i] = lds[i] + lds[n + i] + lds[2 * n + i] + lds[3 * n + i];
}

// One workgroup per chain: Kir_t, Dir_t (integer-valued, exact in any order) and
// the sum of the row constants.
__global__ __launch_bounds__(256) void k_colreduce(Dims d, Work w) {
    __shared__ double sh[4];
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < d.Tp; t += 256) {
        double a = 0.0, e = 0.0;
        for (int rb = 0; rb < d.nrb_scan; ++rb) {
            const double *p = w.colIR + (((size_t)b * d.nrb_scan + rb) * d.Tp + t) * 2;
            a += p[0];
            e += p[1];
        }
        w.Kir[(size_t)b * d.Tp + t] = a;
        w.Dir[(size_t)b * d.Tp + t] = e;
    }
    double acc = 0.0;
    for (int m = threadIdx.x; m < d.M; m += 256) acc += w.rowconst[(size_t)b * d.Mp + m];
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) w.constsum[b] = acc;
}

// ---------------------------------------------------------------------------
// k_gemm: F[b] = Cstar[Mp x Kp] . Xn[b][Kp x Tp] with v_mfma_f64_16x16x4_f64.
// Workgroup = one 16-row panel of Cstar staged in LDS (row stride == 2 mod 32
// doubles: conflict-free ds_read_b64 for the A fragment) x GEMM_TT t-tiles per wave.
// A: lane l holds A[l&15][l>>4]; B: B[l>>4][l&15]; D: row (l>>4)+4r, col l&15.
// ---------------------------------------------------------------------------
constexpr int GEMM_TT = 2;      // 16-wide t-tiles per wave (share the A fragment)
constexpr int GEMM_KC = 512;    // K chunk staged in LDS (66 KB at the cap)
using d4 = __attribute__((ext_vector_type(4))) double;

__host__ __device__ inline int gemm_kc(int Kp) { return Kp < GEMM_KC ? Kp : GEMM_KC; }
__host__ __device__ inline int gemm_lda(int Kp) { return ((gemm_kc(Kp) + 31) / 32) * 32 + 2; }

__global__ __launch_bounds__(256) void k_gemm(Dims d, Consts c, Work w) {
    extern __shared__ double lds[];                 // [16][lda]
    const int b = blockIdx.z, m0 = blockIdx.y * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lda = gemm_lda(d.Kp), kc = gemm_kc(d.Kp);
    const int tt0 = (blockIdx.x * 4 + wave) * GEMM_TT;     // first 16-wide t tile of this wave
    const double *Xb = w.Xn + (size_t)b * d.Mp * d.Tp;
    const int ar = lane & 15, ak = lane >> 4;
    d4 acc[GEMM_TT];
    bool live[GEMM_TT];
#pragma unroll
    for (int j = 0; j < GEMM_TT; ++j) {
        acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
        live[j] = (tt0 + j) * 16 < d.Tp;
    }
    const double *ap = lds + ar * lda + ak;
    for (int kb = 0; kb < d.Kp; kb += kc) {
        const int kn = min(kc, d.Kp - kb);
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * kn; i += 256) {
            const int r = i / kn, k = i - r * kn;
            lds[r * lda + k] = c.Cstar[(size_t)(m0 + r) * d.Kp + kb + k];
        }
        __syncthreads();
        const double *bp = Xb + (size_t)(kb + ak) * d.Tp + tt0 * 16 + ar;
#pragma unroll 4
        for (int k0 = 0; k0 < kn; k0 += 4) {
            const double a = ap[k0];
            const double *bk = bp + (size_t)k0 * d.Tp;
#pragma unroll
            for (int j = 0; j < GEMM_TT; ++j) {
                const double bv = live[j] ? bk[j * 16] : 0.0;
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[j], 0, 0, 0);
            }
        }
    }
    double *Fb = w.F + (size_t)b * d.Mp * d.Tp;
#pragma unroll
    for (int j = 0; j < GEMM_TT; ++j) {
        if (!live[j]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Fb[(size_t)(m0 + ak + 4 * r) * d.Tp + (tt0 + j) * 16 + ar] = acc[j][r];
    }
}

// ---------------------------------------------------------------------------
// Parameter tables for one chain, executed by one 256-thread workgroup.
// Bijector (inference.py:525-535), the rate tables of transition_rate_fn
// (model_spec.py:242-257, 271-274) and the priors (model_spec.py:140-198) with
// the CAR prior in precision form.  `u` is the chain's unconstrained vector.
// ---------------------------------------------------------------------------
__device__ inline void param_tables(const Dims &d, const Consts &c, const Work &w, int b,
                                    const double *__restrict__ u, double *seg /*[256]*/, double *sh /*[4]*/) {
    const int tid = threadIdx.x;
    const double eps = 2.220446049250313e-16;
    const double psi = softplus(u[0]) + eps, sig = softplus(u[1]) + eps;
    const double beta = u[2], g0 = u[3], g1 = u[4], a0 = u[5];
    const double *at = u + 6, *sp = u + 6 + d.T - 1;

    // a_t = alpha_0 + sum_{j<t} alpha_t[j]: per-thread contiguous segments + block scan
    const int per = (d.T + 255) / 256;
    const int t_lo = tid * per, t_hi = min(d.T, t_lo + per);
    double s = 0.0, q_at = 0.0;
    for (int t = t_lo; t < t_hi; ++t)
        if (t >= 1) { const double v = at[t - 1]; s += v; q_at += v * v; }
    __syncthreads();
    seg[tid] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {           // Hillis-Steele inclusive scan
        const double v = tid >= o ? seg[tid - o] : 0.0;
        __syncthreads();
        seg[tid] += v;
        __syncthreads();
    }
    double acc = a0 + (tid ? seg[tid - 1] : 0.0);
    for (int t = t_lo; t < t_hi; ++t) {
        if (t >= 1) acc += at[t - 1];
        w.ea[(size_t)b * d.Tp + t] = exp(acc);
        w.rir[(size_t)b * d.Tp + t] = exp(g0 + g1 * c.wd[t]);
    }
    double quad = 0.0;
    for (int m = tid; m < d.M; m += 256) {
        w.eb[(size_t)b * d.Mp + m] = exp(beta * c.la[m] + sig * sp[m]) * c.invN[m];
        double qs = 0.0;
        for (int e = c.Qrow[m]; e < c.Qrow[m + 1]; ++e) qs += c.Qval[e] * sp[c.Qcol[e]];
        w.Qs[(size_t)b * d.Mp + m] = qs;
        quad += sp[m] * qs;
    }
    quad = block_sum_256(quad, sh);
    q_at = block_sum_256(q_at, sh);
    if (tid == 0) {
        const double LOG_2PI = 1.8378770664093453;
        double lp = -0.5 * a0 * a0 / 100.0 - log(10.0) - 0.5 * LOG_2PI;              // alpha_0 ~ N(0,10)
        lp += -0.5 * beta * beta - 0.5 * LOG_2PI;                                    // beta_area ~ N(0,1)
        lp += 3.0 * log(10.0) - 0.6931471805599453 + 2.0 * log(psi) - 10.0 * psi;    // psi ~ Gamma(3,10)
        lp += -0.5 * q_at / (0.005 * 0.005) - (d.T - 1) * (log(0.005) + 0.5 * LOG_2PI);
        lp += 0.5 * log(2.0 / M_PI) - log(0.1) - sig * sig / 0.02;                   // HalfNormal(0.1)
        lp += -0.5 * quad + d.car_half_logdet - 0.5 * d.M * LOG_2PI;                 // CAR
        lp += -0.5 * g0 * g0 / 1.0e4 - log(100.0) - 0.5 * LOG_2PI;
        lp += -0.5 * g1 * g1 / 1.0e4 - log(100.0) - 0.5 * LOG_2PI;
        double *sc = w.scal + (size_t)b * NSCAL;
        sc[SC_PSI] = psi; sc[SC_SIG] = sig; sc[SC_BETA] = beta; sc[SC_G0] = g0; sc[SC_G1] = g1;
        sc[SC_A0] = a0;
        const double ls0 = -softplus(-u[0]), ls1 = -softplus(-u[1]);
        sc[SC_S0] = exp(ls0); sc[SC_S1] = exp(ls1);
        sc[SC_PRIOR] = lp;
        sc[SC_JAC] = ls0 + ls1;                     // inverse_log_det_jacobian, inference.py:555-557
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_params(Dims d, Consts c, Work w, const double *__restrict__ u_all) {
    __shared__ double sh[4];
    __shared__ double seg[256];
    param_tables(d, c, w, blockIdx.x, u_all + (size_t)blockIdx.x * d.P, seg, sh);
}

// ---------------------------------------------------------------------------
// k_se: the S->E chain-binomial term and its eta/psi derivatives.
//   lambda_mt = exp(eta_mt) (I + psi W_t F_mt)/N_m + 1e-9      model_spec.py:258-266
//   ll = k log(1-exp(-r)) - (S-k) r,  r = lambda dt   (multiply_no_nan: k==0 drops the log)
// Tile = 16 rows x 64 days per workgroup; wave = 4 rows, lane = day: every
// plane is read as 512-B coalesced row segments, 4 independent cells per lane.
// ---------------------------------------------------------------------------
template <bool GRAD, int SRC>
__global__ __launch_bounds__(256) void k_se(Dims d, Consts c, Work w) {
    __shared__ double colbuf[4][WAVE];
    __shared__ double shl[4], shp[4];
    const int b = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * WAVE + lane;
    const int m0 = blockIdx.y * SE_TM + wave * 4;
    const bool valid = t < d.T;
    const double psi = w.scal[(size_t)b * NSCAL + SC_PSI];
    const double ea_t = valid ? w.ea[(size_t)b * d.Tp + t] : 0.0;
    const double Wt = valid ? c.W[t] : 0.0;
    const double psiW = psi * Wt;
    double ll = 0.0, gpsi = 0.0, colacc = 0.0, rowacc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rowacc[r] = 0.0;
        const int m = m0 + r;
        if (m >= d.M) continue;                       // wave-uniform
        const size_t q = ((size_t)b * d.Mp + m) * d.Tp + t;
        const double F = w.F[q];
        double I, kse, snk;
        int ki;
        if (SRC == 0) {
            const int2 ks = w.KS[q];
            I = rint(w.Xn[q] * c.N[m]);
            ki = ks.x; kse = (double)ks.x; snk = (double)ks.y;
        } else {
            ki = w.K[0][q];
            I = (double)w.St[2][q];
            kse = (double)ki; snk = (double)(w.St[0][q] - ki);
        }
        if (!valid) continue;
        const double ee = ea_t * w.eb[(size_t)b * d.Mp + m];
        const double lam0 = ee * (I + psiW * F);
        const double rr = (lam0 + d.rate_floor) * d.dt;
        const double em1 = expm1(-rr);
        ll += (ki != 0 ? kse * log(-em1) : 0.0) - snk * rr;
        if (GRAD) {
            const double gl = d.dt * ((ki != 0 ? kse * (1.0 + em1) / (-em1) : 0.0) - snk);
            const double ge = gl * lam0;
            rowacc[r] = ge;
            colacc += ge;
            gpsi += gl * ee * Wt * F;
        }
    }
    ll = wave_sum(ll);
    if (GRAD) {
        gpsi = wave_sum(gpsi);
        colbuf[wave][lane] = colacc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = wave_sum(rowacc[r]);
            if (lane == 0 && m0 + r < d.M) w.Rpart[((size_t)b * d.ntc + blockIdx.x) * d.Mp + m0 + r] = v;
        }
    }
    if (lane == 0) { shl[wave] = ll; shp[wave] = gpsi; }
    __syncthreads();
    const size_t tile = (size_t)b * d.nmt * d.ntc + (size_t)blockIdx.y * d.ntc + blockIdx.x;
    if (threadIdx.x == 0) {
        w.Lpart[tile] = shl[0] + shl[1] + shl[2] + shl[3];
        if (GRAD) w.Ppart[tile] = shp[0] + shp[1] + shp[2] + shp[3];
    }
    if (GRAD && threadIdx.x < WAVE)
        w.Kpart[((size_t)b * d.nmt + blockIdx.y) * d.Tp + t] =
            colbuf[0][lane] + colbuf[1][lane] + colbuf[2][lane] + colbuf[3][lane];
}

// ---------------------------------------------------------------------------
// Reduction of k_se's partials for one chain by one 256-thread workgroup:
// returns (in every thread) the parameter-dependent log-likelihood
//   lp_se + lp_ir,   lp_ir = sum_t [Kir_t log(1-exp(-r_t)) - Dir_t r_t]
// and, if GRAD, writes d/du of (likelihood + priors + Jacobian) to g[P].
// lds_col needs Tp doubles.
// ---------------------------------------------------------------------------
template <bool GRAD>
__device__ inline double reduce_chain(const Dims &d, const Consts &c, const Work &w, int b,
                                      const double *__restrict__ u, double *__restrict__ g,
                                      double *lds_col, double *seg, double *sh) {
    const int tid = threadIdx.x;
    const double *sc = w.scal + (size_t)b * NSCAL;
    double acc = 0.0, gg0 = 0.0, gg1 = 0.0;
    const int ntile = d.nmt * d.ntc;
    for (int i = tid; i < ntile; i += 256) acc += w.Lpart[(size_t)b * ntile + i];
    for (int t = tid; t < d.Tp; t += 256) {
        double col = 0.0;
        if (t < d.T) {
            const double kir = w.Kir[(size_t)b * d.Tp + t], dir = w.Dir[(size_t)b * d.Tp + t];
            const double rate = w.rir[(size_t)b * d.Tp + t];
            const double r = rate * d.dt;
            const double em1 = expm1(-r);
            acc += (kir != 0.0 ? kir * log(-em1) : 0.0) - dir * r;
            if (GRAD) {
                const double gr = d.dt * ((kir != 0.0 ? kir * (1.0 + em1) / (-em1) : 0.0) - dir);
                gg0 += gr * rate;
                gg1 += gr * rate * c.wd[t];
                for (int ty = 0; ty < d.nmt; ++ty) col += w.Kpart[((size_t)b * d.nmt + ty) * d.Tp + t];
            }
        }
        if (GRAD) lds_col[t] = col;
    }
    acc = block_sum_256(acc, sh);
    if (!GRAD) return acc;

    const double psi = sc[SC_PSI], sig = sc[SC_SIG], beta = sc[SC_BETA];
    const double *at = u + 6, *sp = u + 6 + d.T - 1;
    // suffix sums of the column sums: d/d alpha_t[j] = sum_{t>j} col[t]
    const int per = (d.T + 255) / 256;
    const int t_lo = tid * per, t_hi = min(d.T, t_lo + per);
    double csum = 0.0;
    __syncthreads();
    for (int t = t_lo; t < t_hi; ++t) csum += lds_col[t];
    seg[tid] = csum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {             // inclusive suffix scan
        const double v = tid + o < 256 ? seg[tid + o] : 0.0;
        __syncthreads();
        seg[tid] += v;
        __syncthreads();
    }
    const double total = seg[0];
    double run = tid + 1 < 256 ? seg[tid + 1] : 0.0;
    for (int t = t_hi - 1; t >= t_lo; --t) {
        run += lds_col[t];
        if (t >= 1) g[6 + t - 1] = run - at[t - 1] / (0.005 * 0.005);
    }
    double gsig = 0.0, gbeta = 0.0;
    for (int m = tid; m < d.M; m += 256) {
        double R = 0.0;
        for (int tx = 0; tx < d.ntc; ++tx) R += w.Rpart[((size_t)b * d.ntc + tx) * d.Mp + m];
        gsig += sp[m] * R;
        gbeta += c.la[m] * R;
        g[6 + d.T - 1 + m] = sig * R - w.Qs[(size_t)b * d.Mp + m];
    }
    double gpsi = 0.0;
    for (int i = tid; i < ntile; i += 256) gpsi += w.Ppart[(size_t)b * ntile + i];
    gsig = block_sum_256(gsig, sh);
    gbeta = block_sum_256(gbeta, sh);
    gpsi = block_sum_256(gpsi, sh);
    gg0 = block_sum_256(gg0, sh);
    gg1 = block_sum_256(gg1, sh);
    if (tid == 0) {
        const double s0 = sc[SC_S0], s1 = sc[SC_S1];
        g[0] = (gpsi + 2.0 / psi - 10.0) * s0 + (1.0 - s0);
        g[1] = (gsig - sig / 0.01) * s1 + (1.0 - s1);
        g[2] = gbeta - beta;
        g[3] = gg0 - sc[SC_G0] / 1.0e4;
        g[4] = gg1 - sc[SC_G1] / 1.0e4;
        g[5] = total - sc[SC_A0] / 100.0;
    }
    return acc;
}

// k_finish: joint log-prob (+ gradient) of the API path, one workgroup per chain.
template <bool GRAD>
__global__ __launch_bounds__(256) void k_finish(Dims d, Consts c, Work w, const double *__restrict__ u_all,
                                               double *__restrict__ logp, double *__restrict__ grad) {
    extern __shared__ double lds_col[];             // [Tp]
    __shared__ double sh[4];
    __shared__ double seg[256];
    const int b = blockIdx.x;
    const double lp = reduce_chain<GRAD>(d, c, w, b, u_all + (size_t)b * d.P,
                                         GRAD ? grad + (size_t)b * d.P : nullptr, lds_col, seg, sh);
    if (threadIdx.x == 0) {
        const double *sc = w.scal + (size_t)b * NSCAL;
        logp[b] = lp + w.constsum[b] + sc[SC_PRIOR] + sc[SC_JAC];
    }
}

}  // namespace seir
