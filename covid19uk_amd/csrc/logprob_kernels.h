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
// Per-chain workspace (HBM), rows padded to Mp = ceil64(M), days to Tp = ceil64(T), pads zero
// (zero pads contribute exactly 0 to every sum, so the cell kernels carry no bounds masks):
//   API path     Xn fp64 (I/N), KS int2 (k_se, S-k_se), F fp64
//   sampler path int32 planes K[3] (events), St[3] (S,E,I at start of day), F fp64
#pragma once
#include "device_math.h"

namespace seir {

constexpr int SCAN_ROWS = 8;    // rows per k_scan workgroup
constexpr int SCAN_WAVES = 8;   // one row per wave: 3 waves per SIMD at UK-380 x 8 chains, evenly
constexpr int SCAN_CB = 6;      // k_scan: 64-day chunks fetched per batch (6 = one batch at T <= 384)
constexpr int SCAN_LFT = 2048;  // k_scan: entries of the log-factorial table held in LDS (16 KB)
constexpr int SE_RW = 4;        // k_se: rows per wave; tile = (4*SE_RW) rows x 64 days per workgroup
constexpr int SE_TM = 4 * SE_RW;
constexpr int NSCAL = 16;       // per-chain scalar block
enum { SC_PSI = 0, SC_SIG, SC_BETA, SC_G0, SC_G1, SC_A0, SC_S0, SC_S1, SC_PRIOR, SC_JAC };

struct Dims {
    int M, T, Mp, Tp, Kp, P, Pp;
    int Kp0;                // row stride of the padded Cstar [Kp rows used][Kp0 = Mp columns]
    int b0;                 // first chain handled by this launch (chain groups on separate streams)
    int nrb_scan;           // row blocks of k_scan
    int nmt, ntc;           // k_se tiles: Mp/SE_TM row tiles, Tp/64 day chunks
    int chunked;            // sampler: k_se also writes the tile scalars of the chunked leapfrog (Work::TS):
                            // 1 = column scalars only (the M-chunks sum the row partials themselves), 2 = all four
    int sp_par;             // which of the two Work::sp / Work::gst buffers holds the current position
    int skew;               // test hook (seir_set_option SEIR_OPT_DEBUG_SKEW = 1..3): a third of the workgroups of every launch starts ~30 us late
    int aff_nb;             // 0 = natural grids (tile, chain); > 0 = 1-D grids of tiles*aff_nb blocks with chain <-> XCD affinity
    int nlive;              // k_se_chunk, k_move_pair: > 0 = the grid is laid out for aff_nb / nbk chains but only the first nlive
                            // exist (fewer than 8 chains in the 8-chain layout: every chain whole on one XCD); their blocks retire at once
    double nu, dt, rate_floor, car_half_logdet;
    double L_ei;            // log(1 - exp(-nu dt))
    double prior_const;     // parameter-free part of the summed prior log-densities
};

// Test hook: delays a pseudo-random third of the workgroups of a launch.  Results must not depend on it
// (tests/test_sampler_gpu.py): nothing may be read that another workgroup of the same launch writes.
__device__ __forceinline__ void debug_skew(const Dims &d) {
    if (d.skew == 0) return;
    const unsigned id = blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u;
    if (((id * 2654435761u) >> 16) % 3u == (unsigned)(d.skew - 1))
        for (int i = 0; i < 10; ++i) __builtin_amdgcn_s_sleep(127);
}

struct Consts {
    const double *Cstar;   // [Mp][Kp0], Kp0 = Mp: zero-padded, symmetric
    const float *Cstar32;  // the same rounded to fp32 (k_gemm_f32; allocated when SEIR_OPT_GEMM_F32 is first set)
    const double *N, *invN, *la;   // [Mp]
    const double *W, *wd;          // [Tp]
    const double *init;            // [Mp][4]
    const int *Qrow, *Qcol;        // CSR of car_Q
    const double *Qval;
    const double2 *logtab;         // [LDSTAB_N] (1/c, log c) of device_math.h fast_log, then log(n!) pairs
    const double *lfact_big;       // [SCAN_LFT] log(n!) for n < SCAN_LFT: k_scan's LDS table (E, I and the event counts
                                   //            are mostly below it; Stirling above)
    int qw;                        // ELL width of car_Q (0: use the CSR arrays)
    const int *Qell_col;           // [qw][Mp]
    const double *Qell_val;        // [qw][Mp]
};

struct Work {
    double *Xn, *F;        // [B][Mp][Tp]
    float *Xn32;           // [B][Mp][Tp] I/N rounded to fp32 for k_gemm_f32 (null unless SEIR_OPT_GEMM_F32 was set)
    int2 *KS;              // [B][Mp][Tp]
    int *K[3], *St[3];     // sampler planes [B][Mp][Tp] (null on a plain context)
    int *rowtot;           // [B][2][Mp] row totals of S->E / E->I events (sampler)
    int *rngtot;           // [B][2][Mp] the same inside the occult range [tr_lo, tr_hi) (sampler): kept up to date
                           //            by every accepted update, so a proposal reads one int per row instead of 21
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
    double *Lpart0;        // [B][nmt*ntc] sampler, k_leap: the S->E term's partial sums at the START point of a trajectory (Lpart
                           //              is overwritten by the end point's)
    double *Kpart;         // [B][nmt][Tp]
    double *Rpart;         // [B][ntc][Mp]
    // chunked leapfrog (sampler only; null on a plain context) -- see k_hmc_chunk in sampler_kernels.h
    double *TS;            // [B][nmt*ntc][4] per k_se tile: sum col, sum col*V(t), sum_m l_m*row, sum_m s_m*row
    double *sp;            // [B][2][Mp] spatial effect at the position the tables were built for (double-buffered)
    double *gst;           // [B][2][GST_N] the six global parameters, their momenta, psi, sigma and the two sigmoids
    double *Vt;            // [B][Tp]  V(t) = sum_{s=1..t} var[alpha_t[s-1]]
    double *acur;          // [B][Tp]  a_t = alpha_0 + cumsum(alpha_t)[t-1] at the current position
    double *rirc;          // (unused since the chunked steps hand each other the I->R gradient parts, Work::CG)
    double *CT;            // [B][2][CT_MAXC][4] per 64-day chunk: sum alpha, sum v p, sum v alpha (double-buffered)
    double *CG;            // [B][2][CT_MAXC][2] per 64-day chunk: its part of d/d gamma0, d/d gamma1 of the I->R term at the
                           //            position the step arrives at (double-buffered like CT)
};
constexpr int CT_MAXC = 16;     // Tp/64 <= 16 (T <= 1024)
constexpr int GST_N = 16;       // q[0..5], p[0..5], psi, sigma_space, sigmoid(u0), sigmoid(u1)

// ---------------------------------------------------------------------------
// k_scan: one wave per row (m); lanes over days in 64-day chunks with a carry.
// State at the START of day t (gemlib compute_state, call site inference.py:500-510).
// SRC 0: events fp64 [B][M][T][3] (reference layout); writes Xn, KS.
// SRC 1: events from the sampler's int32 planes; writes St planes, Xn, rowtot.
// PART 0: everything.  The fused evaluation (k_state_params / k_eval_tiles) splits the work by what the
// contraction waits for: PART 1 = the state only (Xn, KS, the per-day I->R partials) -- a short, HBM-bound
// pass -- and PART 2 = the row constants only (binomial coefficients and the E->I term: the fp64-VALU-heavy
// part, which nothing but the final sum needs), run beside the matrix-core tiles.  Row block bx of chain by.
// ---------------------------------------------------------------------------
template <int SRC, int PART = 0>
__device__ __forceinline__ void scan_rows(const Dims &d, const Consts &c, const Work &w, const double *__restrict__ events,
                                          int bx, int by) {
    extern __shared__ double lds[];                 // PART 0: [SCAN_WAVES][Tp][2] | lft [SCAN_LFT]; 1: no lft; 2: lft only
    __shared__ double2 ltab[LDSTAB_N];
    debug_skew(d);
    constexpr bool STATE = PART != 2, ROWC = PART != 1;
    const int b = d.b0 + by, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *mycol = lds + (size_t)wave * d.Tp * 2;
    double *lft = lds + (PART == 2 ? 0 : (size_t)SCAN_WAVES * d.Tp * 2);
    if (ROWC)
        for (int i = threadIdx.x; i < SCAN_LFT; i += SCAN_WAVES * WAVE) lft[i] = c.lfact_big[i];
    if (STATE)
        for (int i = lane; i < d.Tp * 2; i += WAVE) mycol[i] = 0.0;
    if (ROWC) log_table_to_lds(ltab, c.logtab);
    else __syncthreads();
    // log(n!): LDS table below SCAN_LFT (one ds_read instead of ~40 dependent fp64 operations), Stirling above
    auto lf = [&](double n) { return n < (double)SCAN_LFT ? lft[(int)n] : lfact(n, ltab); };
    auto lb = [&](double n, double k) { return (k < 0.0 || k > n) ? -INFINITY : lf(n) - lf(k) - lf(n - k); };

    const double r_ei = d.nu * d.dt;
    const double L_ei = d.L_ei;
    const int nch = d.Tp / WAVE;
    constexpr int RPW = SCAN_ROWS / SCAN_WAVES;
    constexpr int CB = SCAN_CB;                      // day chunks loaded per batch, before any arithmetic
    for (int r = 0; r < RPW; ++r) {
        const int m = bx * SCAN_ROWS + wave * RPW + r;
        if (m >= d.M) break;
        const double S0 = c.init[m * 4 + 0], E0 = c.init[m * 4 + 1], I0 = c.init[m * 4 + 2];
        const double invN = c.invN[m];
        const size_t rowoff = ((size_t)b * d.Mp + m) * d.Tp;
        const double *ev = SRC == 0 ? events + ((size_t)b * d.M + m) * d.T * 3 : nullptr;
        double cse = 0.0, cei = 0.0, cir = 0.0, rc = 0.0;
        for (int ch0 = 0; ch0 < nch; ch0 += CB) {
            // all loads of the batch first: the chunks are independent up to the carried prefix,
            // so nothing below waits on memory more than once per batch
            double kse[CB], kei[CB], kir[CB];
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                const int t = (ch0 + j) * WAVE + lane;
                kse[j] = kei[j] = kir[j] = 0.0;
                if (ch0 + j < nch) {
                    if (SRC == 0) {
                        if (t < d.T) {
                            kse[j] = ev[(size_t)t * 3 + 0];
                            kei[j] = ev[(size_t)t * 3 + 1];
                            kir[j] = ev[(size_t)t * 3 + 2];
                        }
                    } else {
                        kse[j] = (double)w.K[0][rowoff + t];
                        kei[j] = (double)w.K[1][rowoff + t];
                        kir[j] = (double)w.K[2][rowoff + t];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                if (ch0 + j >= nch) break;
                const int t = (ch0 + j) * WAVE + lane;
                const bool valid = t < d.T;
                const double ise = wave_incl_scan(kse[j], lane), iei = wave_incl_scan(kei[j], lane),
                             iir = wave_incl_scan(kir[j], lane);
                const double xse = cse + ise - kse[j], xei = cei + iei - kei[j], xir = cir + iir - kir[j];
                const double S = S0 - xse, E = E0 + xse - xei, I = I0 + xei - xir;
                if (STATE) {
                    w.Xn[rowoff + t] = valid ? I * invN : 0.0;
                    if (w.Xn32 != nullptr) w.Xn32[rowoff + t] = valid ? (float)(I * invN) : 0.f;
                    if (SRC == 0) {
                        w.KS[rowoff + t] = valid ? make_int2((int)kse[j], (int)(S - kse[j])) : make_int2(0, 0);
                    } else {
                        w.St[0][rowoff + t] = valid ? (int)S : 0;
                        w.St[1][rowoff + t] = valid ? (int)E : 0;
                        w.St[2][rowoff + t] = valid ? (int)I : 0;
                    }
                    if (valid) {
                        mycol[t * 2 + 0] += kir[j];
                        mycol[t * 2 + 1] += I - kir[j];
                    }
                }
                if (ROWC && valid) {
                    // S only ever loses its events (S_{t+1} = S_t - k_t), so its binomial coefficients telescope:
                    //   sum_t [lf(S_t) - lf(k_t) - lf(S_t - k_t)] = lf(S_0) - lf(S_T) - sum_t lf(k_t)
                    // -- one small-argument term per cell here, the two large ones once per row below; E and I
                    // also gain events and do not telescope.  Infeasible counts (k < 0 or k > S) still give -inf.
                    rc += (kse[j] < 0.0 || kse[j] > S) ? -INFINITY : -lf(kse[j]);
                    rc += lb(E, kei[j]) + lb(I, kir[j]);
                    rc += kei[j] * L_ei - (E - kei[j]) * r_ei;
                }
                cse += __shfl(ise, WAVE - 1, WAVE);
                cei += __shfl(iei, WAVE - 1, WAVE);
                cir += __shfl(iir, WAVE - 1, WAVE);
            }
        }
        if (ROWC) rc = wave_sum(rc);
        if (lane == 0) {
            // the telescoped ends (cse = all S->E events of the row); an exhausted S already produced -inf above
            if (ROWC) {
                rc += S0 - cse >= 0.0 ? lfact(S0, ltab) - lfact(S0 - cse, ltab) : -INFINITY;
                w.rowconst[(size_t)b * d.Mp + m] = rc;
            }
            if (SRC == 1) {
                w.rowtot[((size_t)b * 2 + 0) * d.Mp + m] = (int)cse;
                w.rowtot[((size_t)b * 2 + 1) * d.Mp + m] = (int)cei;
            }
        }
    }
    if (!STATE) return;
    __syncthreads();
    double *out = w.colIR + ((size_t)b * d.nrb_scan + bx) * d.Tp * 2;
    const int n = d.Tp * 2;
    for (int i = threadIdx.x; i < n; i += SCAN_WAVES * WAVE) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < SCAN_WAVES; ++k) a += lds[k * n + i];
        out[i] = a;
    }
}

// The state part (scan_rows<0, 1>) of TWO row blocks of a chain by one workgroup, for k_eval_all's tile workgroups, which
// scan two blocks each: the second block's events are fetched before the first block's prefix sums are formed, so that
// their latency (first touch: HBM or the Infinity Cache, the events are read once per evaluation) runs under the first
// block's arithmetic instead of after it.  Per block the loads, the operations and their order are scan_rows': the same
// bits.  One batch of day chunks per row (T <= 64 SCAN_CB); the caller falls back to two calls otherwise.
__device__ __forceinline__ void scan_rows_state2(const Dims &d, const Consts &c, const Work &w, const double *__restrict__ events,
                                                 int bx0, int bx1, int by) {
    extern __shared__ double lds[];                 // [SCAN_WAVES][Tp][2]
    static_assert(SCAN_ROWS == SCAN_WAVES, "one row per wave");
    const int b = d.b0 + by, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *mycol = lds + (size_t)wave * d.Tp * 2;
    const int nch = d.Tp / WAVE;
    constexpr int CB = SCAN_CB;
    double ev_[2][3][CB];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int m = (q == 0 ? bx0 : bx1) * SCAN_ROWS + wave;
        const double *ev = events + ((size_t)b * d.M + (m < d.M ? m : 0)) * d.T * 3;
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int t = j * WAVE + lane;
            const bool in = m < d.M && j < nch && t < d.T;
            ev_[q][0][j] = in ? ev[(size_t)t * 3 + 0] : 0.0;
            ev_[q][1][j] = in ? ev[(size_t)t * 3 + 1] : 0.0;
            ev_[q][2][j] = in ? ev[(size_t)t * 3 + 2] : 0.0;
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int bx = q == 0 ? bx0 : bx1;
        const int m = bx * SCAN_ROWS + wave;
        for (int i = lane; i < d.Tp * 2; i += WAVE) mycol[i] = 0.0;
        __syncthreads();
        if (m < d.M) {
            const double S0 = c.init[m * 4 + 0], E0 = c.init[m * 4 + 1], I0 = c.init[m * 4 + 2];
            const double invN = c.invN[m];
            const size_t rowoff = ((size_t)b * d.Mp + m) * d.Tp;
            double cse = 0.0, cei = 0.0, cir = 0.0;
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                if (j >= nch) break;
                const int t = j * WAVE + lane;
                const bool valid = t < d.T;
                const double kse = ev_[q][0][j], kei = ev_[q][1][j], kir = ev_[q][2][j];
                const double ise = wave_incl_scan(kse, lane), iei = wave_incl_scan(kei, lane), iir = wave_incl_scan(kir, lane);
                const double xse = cse + ise - kse, xei = cei + iei - kei, xir = cir + iir - kir;
                const double S = S0 - xse, I = I0 + xei - xir;
                w.Xn[rowoff + t] = valid ? I * invN : 0.0;
                if (w.Xn32 != nullptr) w.Xn32[rowoff + t] = valid ? (float)(I * invN) : 0.f;
                w.KS[rowoff + t] = valid ? make_int2((int)kse, (int)(S - kse)) : make_int2(0, 0);
                if (valid) {
                    mycol[t * 2 + 0] += kir;
                    mycol[t * 2 + 1] += I - kir;
                }
                cse += __shfl(ise, WAVE - 1, WAVE);
                cei += __shfl(iei, WAVE - 1, WAVE);
                cir += __shfl(iir, WAVE - 1, WAVE);
                (void)E0;
            }
        }
        __syncthreads();
        double *out = w.colIR + ((size_t)b * d.nrb_scan + bx) * d.Tp * 2;
        const int n = d.Tp * 2;
        for (int i = threadIdx.x; i < n; i += SCAN_WAVES * WAVE) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < SCAN_WAVES; ++k) a += lds[k * n + i];
            out[i] = a;
        }
        __syncthreads();                                // the next block reuses the LDS
    }
}

template <int SRC>
__global__ __launch_bounds__(SCAN_WAVES * WAVE) void k_scan(Dims d, Consts c, Work w, const double *__restrict__ events) {
    scan_rows<SRC>(d, c, w, events, blockIdx.x, blockIdx.y);
}

// Grid (Tp/64, chains): Kir_t, Dir_t of 64 days per workgroup (integer-valued, exact in any
// order); block 0 of a chain also sums the row constants.
__device__ __forceinline__ void colreduce_block(const Dims &d, const Work &w, int bx, int by, bool with_const = true) {
    __shared__ double sh[4];
    __shared__ double2 part[4][WAVE];
    debug_skew(d);
    const int b = d.b0 + by, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = bx * WAVE + lane;
    // wave w folds the partial rows rb = w, w+4, ...; 16 loads in flight per batch
    double a = 0.0, e = 0.0;
    const double2 *p = (const double2 *)w.colIR + (size_t)b * d.nrb_scan * d.Tp + t;
    for (int rb0 = wave; rb0 < d.nrb_scan; rb0 += 64) {
        double2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int rb = rb0 + 4 * j;
            v[j] = rb < d.nrb_scan ? p[(size_t)rb * d.Tp] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) { a += v[j].x; e += v[j].y; }      // integer-valued sums: any order is exact
    }
    part[wave][lane] = make_double2(a, e);
    __syncthreads();
    if (wave == 0) {
        const double2 p0 = part[0][lane], p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
        w.Kir[(size_t)b * d.Tp + t] = (p0.x + p1.x) + (p2.x + p3.x);
        w.Dir[(size_t)b * d.Tp + t] = (p0.y + p1.y) + (p2.y + p3.y);
    }
    if (bx == 0 && with_const) {
        double acc = 0.0;
        for (int m = threadIdx.x; m < d.M; m += 256) acc += w.rowconst[(size_t)b * d.Mp + m];
        acc = block_sum_256(acc, sh);
        if (threadIdx.x == 0) w.constsum[b] = acc;
    }
}
__global__ __launch_bounds__(256) void k_colreduce(Dims d, Work w) { colreduce_block(d, w, blockIdx.x, blockIdx.y); }

// ---------------------------------------------------------------------------
// k_gemm: F[b] = Cstar[Mp x Kp] . Xn[b][Kp x Tp] for all T at once (the matvec of
// model_spec.py:262 for every day), fp64 on the matrix cores: v_mfma_f64_16x16x4_f64
// (A: lane l holds A[l&15][l>>4]; B: B[l>>4][l&15]; D: row (l>>4)+4r, col l&15).
//
// Workgroup tile 64 (rows) x 64 (days), 4 waves as 2 x 2, each wave a 32 x 32 sub-tile =
// 2 x 2 MFMA tiles with independent accumulators (4 MFMAs per pair of A and pair of B fragments,
// back-to-back issue).  K runs in chunks of 64 staged in LDS; the next chunk's global loads are
// issued into registers before the 64 MFMAs of the current chunk (~1.7 us of matrix work, longer
// than the load latency) and written to LDS after them.  Cstar is symmetric
// (C + C^T with a diagonal), so the A chunk is read as Cstar[k][m0..m0+63] -- contiguous rows --
// and lands in LDS already "transposed" for the A fragment; row stride 80 doubles (== 16 mod 32)
// makes both fragment reads conflict-free ds_read_b64.
// ---------------------------------------------------------------------------
constexpr int GEMM_TM = 64, GEMM_KC = 64, GEMM_RS = 80;
using d4 = __attribute__((ext_vector_type(4))) double;

// TN = 64: 4 waves (2 x 2), TN = 96: 6 waves (2 x 3) -- at Tp = 384 (UK) 36 tiles of 64 x 64 per chain
// are 288 workgroups on 256 CUs, two rounds; 24 tiles of 64 x 96 are 192, one round with 1.5 waves
// per SIMD.  The B panel's row stride is TN + 16 (== 16 mod 32 like the A panel's 80).
template <int TN> __host__ __device__ constexpr int gemm_rsb() { return TN + 16; }
template <int TN> __host__ __device__ constexpr int gemm_threads() { return (GEMM_TM / 32) * (TN / 32) * WAVE; }
template <int TN>
__host__ __device__ inline size_t gemm_lds_bytes() {
    return (size_t)GEMM_KC * (GEMM_RS + gemm_rsb<TN>()) * sizeof(double);
}

template <int TN>
__global__ __launch_bounds__(gemm_threads<TN>()) void k_gemm(Dims d, Consts c, Work w) {
    extern __shared__ double lds[];                 // A [KC][RS] | B [KC][RSB]
    constexpr int NT = gemm_threads<TN>(), RSB = gemm_rsb<TN>(), NWC = TN / 32;
    debug_skew(d);
    const int b = d.b0 + blockIdx.z, m0 = blockIdx.y * GEMM_TM, t0 = blockIdx.x * TN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave / NWC, wc = wave % NWC;
    const double *Xb = w.Xn + (size_t)b * d.Mp * d.Tp;
    double *A = lds, *Bm = lds + GEMM_KC * GEMM_RS;
    // staging in double2 elements: A chunk KC x 64, B chunk KC x TN
    constexpr int EA = GEMM_KC * (GEMM_TM / 2), EB = GEMM_KC * (TN / 2);
    constexpr int NA = (EA + NT - 1) / NT, NB = (EB + NT - 1) / NT;
    double2 ra[NA], rb[NB];
    auto load_chunk = [&](int kb) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + i * NT, row = e / (GEMM_TM / 2), col = (e % (GEMM_TM / 2)) * 2, k = kb + row;
            ra[i] = (e < EA && k < d.Kp) ? *(const double2 *)(c.Cstar + (size_t)k * d.Kp0 + m0 + col) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + i * NT, row = e / (TN / 2), col = (e % (TN / 2)) * 2, k = kb + row;
            rb[i] = (e < EB && k < d.Kp) ? *(const double2 *)(Xb + (size_t)k * d.Tp + t0 + col) : make_double2(0.0, 0.0);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + i * NT, row = e / (GEMM_TM / 2), col = (e % (GEMM_TM / 2)) * 2;
            if (e < EA) *(double2 *)(A + row * GEMM_RS + col) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + i * NT, row = e / (TN / 2), col = (e % (TN / 2)) * 2;
            if (e < EB) *(double2 *)(Bm + row * RSB + col) = rb[i];
        }
    };
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
    const int ar = lane & 15, ak = lane >> 4;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int kb = 0; kb < d.Kp; kb += GEMM_KC) {
        const bool more = kb + GEMM_KC < d.Kp;
        if (more) load_chunk(kb + GEMM_KC);          // in flight behind the MFMAs of this chunk
#pragma unroll
        for (int kk = 0; kk < GEMM_KC; kk += 4) {
            const double *ap = A + (kk + ak) * GEMM_RS + wr * 32 + ar;
            const double *bp = Bm + (kk + ak) * RSB + wc * 32 + ar;
            const double a0 = ap[0], a1 = ap[16], b0 = bp[0], b1 = bp[16];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();                             // every wave is done with the chunk in LDS
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }
    double *Fb = w.F + (size_t)b * d.Mp * d.Tp;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Fb[(size_t)(m0 + wr * 32 + i * 16 + ak + 4 * r) * d.Tp + t0 + wc * 32 + j * 16 + ar] = acc[i][j][r];
}

// k_gemm_w8: the 64 x 96 tile with EIGHT waves, 4 (row blocks of 16) x 2 (column groups of 48), i.e.
// 1 x 3 MFMA tiles per wave.  The six-wave form of k_gemm<96> puts 2,2,1,1 waves on the four SIMDs of
// a CU and the two doubly loaded SIMDs set the time; eight waves load every SIMD alike (2 each, which
// also hides more of the fp64 MFMA's issue latency).  Same staging, LDS layout and arithmetic order
// per output element as k_gemm (K ascending in chunks of 64), so the results are bit-identical.
template <int NCG>      // column groups per tile: 2 -> 8 waves of 1 x 3 MFMA tiles, 3 -> 12 waves of 1 x 2
__global__ __launch_bounds__(256 * NCG) void k_gemm_w8(Dims d, Consts c, Work w) {
    extern __shared__ double lds[];                 // A [KC][RS] | B [KC][RSB]
    constexpr int TN = 96, NT = 256 * NCG, RSB = gemm_rsb<TN>(), CW = TN / NCG, NJ = CW / 16;
    debug_skew(d);
    const int b = d.b0 + blockIdx.z, m0 = blockIdx.y * GEMM_TM, t0 = blockIdx.x * TN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave / NCG, wc = wave % NCG;        // row block (16 rows), column group (CW columns)
    const double *Xb = w.Xn + (size_t)b * d.Mp * d.Tp;
    double *A = lds, *Bm = lds + GEMM_KC * GEMM_RS;
    constexpr int EA = GEMM_KC * (GEMM_TM / 2), EB = GEMM_KC * (TN / 2);
    constexpr int NA = (EA + NT - 1) / NT, NB = (EB + NT - 1) / NT;
    double2 ra[NA], rb[NB];
    auto load_chunk = [&](int kb) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + i * NT, row = e / (GEMM_TM / 2), col = (e % (GEMM_TM / 2)) * 2, k = kb + row;
            ra[i] = (e < EA && k < d.Kp) ? *(const double2 *)(c.Cstar + (size_t)k * d.Kp0 + m0 + col) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + i * NT, row = e / (TN / 2), col = (e % (TN / 2)) * 2, k = kb + row;
            rb[i] = (e < EB && k < d.Kp) ? *(const double2 *)(Xb + (size_t)k * d.Tp + t0 + col) : make_double2(0.0, 0.0);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + i * NT, row = e / (GEMM_TM / 2), col = (e % (GEMM_TM / 2)) * 2;
            if (e < EA) *(double2 *)(A + row * GEMM_RS + col) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + i * NT, row = e / (TN / 2), col = (e % (TN / 2)) * 2;
            if (e < EB) *(double2 *)(Bm + row * RSB + col) = rb[i];
        }
    };
    d4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    const int ar = lane & 15, ak = lane >> 4;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int kb = 0; kb < d.Kp; kb += GEMM_KC) {
        const bool more = kb + GEMM_KC < d.Kp;
        if (more) load_chunk(kb + GEMM_KC);          // in flight behind the MFMAs of this chunk
#pragma unroll
        for (int kk = 0; kk < GEMM_KC; kk += 4) {
            const double a0 = A[(kk + ak) * GEMM_RS + wr * 16 + ar];
            const double *bp = Bm + (kk + ak) * RSB + wc * CW + ar;
            double bf[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = bp[16 * j];
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bf[j], acc[j], 0, 0, 0);
        }
        __syncthreads();                             // every wave is done with the chunk in LDS
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }
    double *Fb = w.F + (size_t)b * d.Mp * d.Tp;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Fb[(size_t)(m0 + wr * 16 + ak + 4 * r) * d.Tp + t0 + wc * CW + j * 16 + ar] = acc[j][r];
}

// k_gemm_f32: the same contraction with fp32 operands on v_mfma_f32_32x32x2_f32 (BASELINE config 5:
// "fp32 MFMA mobility matvec" at 2048 regions x 730 days).  Cstar is kept as an fp32 copy (Consts::Cstar32),
// X = I/N is written in fp32 by the state scan (Work::Xn32), products accumulate in fp32 (a k-ordered fmaf chain,
// cdna_hip_programming.md section 3) and F is written back as fp64 for the fp64 likelihood kernels.  The
// result carries ~1e-7 relative error in F (measured through the log-prob: tests/test_logprob_gpu.py), which
// is why it is an option (SEIR_OPT_GEMM_F32) and not the default: the stated 1e-9 needs the fp64 kernel.
// Tile 128 x 128, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D: col = l&31, row = (r&3) + 8 (r>>2) + 4 (l>>5)); K in chunks of 16 through two LDS buffers (one barrier
// per chunk; the next chunk's global loads are in flight behind the 32 MFMAs of the current one).
constexpr int GF_T = 128, GF_KC = 16, GF_RS = GF_T + 4;
using f16v = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void k_gemm_f32(Dims d, Consts c, Work w) {
    __shared__ float As[2][GF_KC][GF_RS], Bs[2][GF_KC][GF_RS];
    debug_skew(d);
    const int b = d.b0 + blockIdx.z, m0 = blockIdx.y * GF_T, t0 = blockIdx.x * GF_T;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const float *Xb = w.Xn32 + (size_t)b * d.Mp * d.Tp;
    // staging: A and B chunks 16 x 128 floats = 512 float4 each (2 + 2 per thread)
    float4 ra[2], rb[2];
    auto load_chunk = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256, row = e >> 5, col = (e & 31) * 4, k = kb + row;
            ra[i] = k < d.Kp ? *(const float4 *)(c.Cstar32 + (size_t)k * d.Kp0 + m0 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256, row = e >> 5, col = (e & 31) * 4, k = kb + row;
            rb[i] = k < d.Kp ? *(const float4 *)(Xb + (size_t)k * d.Tp + t0 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256, row = e >> 5, col = (e & 31) * 4;
            *(float4 *)&As[buf][row][col] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256, row = e >> 5, col = (e & 31) * 4;
            *(float4 *)&Bs[buf][row][col] = rb[i];
        }
    };
    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int ln = lane & 31, lk = lane >> 5;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    int buf = 0;
    for (int kb = 0; kb < d.Kp; kb += GF_KC) {
        const bool more = kb + GF_KC < d.Kp;
        if (more) load_chunk(kb + GF_KC);
#pragma unroll
        for (int kk = 0; kk < GF_KC; kk += 2) {
            const float a0 = As[buf][kk + lk][wr * 64 + ln], a1 = As[buf][kk + lk][wr * 64 + 32 + ln];
            const float b0 = Bs[buf][kk + lk][wc * 64 + ln], b1 = Bs[buf][kk + lk][wc * 64 + 32 + ln];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) store_chunk(buf ^ 1);              // the other buffer: nobody reads it during this chunk
        __syncthreads();
        buf ^= 1;
    }
    double *Fb = w.F + (size_t)b * d.Mp * d.Tp;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                Fb[(size_t)row * d.Tp + t0 + wc * 64 + j * 32 + ln] = (double)acc[i][j][r];
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
    double tot_unused;
    double acc = a0 + block_excl_scan_256(s, seg, tot_unused);
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
        // model_spec.py:140-198; the parameter-free normalisers are folded into d.prior_const
        double lp = d.prior_const;
        lp += -0.5 * a0 * a0 / 100.0;                                 // alpha_0 ~ N(0,10)
        lp += -0.5 * beta * beta;                                     // beta_area ~ N(0,1)
        lp += 2.0 * cold_log(psi) - 10.0 * psi;                       // psi ~ Gamma(3,10)
        lp += -0.5 * q_at / (0.005 * 0.005);                          // alpha_t ~ N(0, 0.005)
        lp += -sig * sig / 0.02;                                      // sigma_space ~ HalfNormal(0.1)
        lp += -0.5 * quad;                                            // CAR, precision form
        lp += -0.5 * (g0 * g0 + g1 * g1) / 1.0e4;                     // gamma0, gamma1 ~ N(0,100)
        double *sc = w.scal + (size_t)b * NSCAL;
        sc[SC_PSI] = psi; sc[SC_SIG] = sig; sc[SC_BETA] = beta; sc[SC_G0] = g0; sc[SC_G1] = g1;
        sc[SC_A0] = a0;
        const double ls0 = u[0] - (psi - eps), ls1 = u[1] - (sig - eps);   // log sigmoid(u) = u - softplus(u)
        sc[SC_S0] = cold_exp(ls0); sc[SC_S1] = cold_exp(ls1);
        sc[SC_PRIOR] = lp;
        sc[SC_JAC] = ls0 + ls1;                     // inverse_log_det_jacobian, inference.py:555-557
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_params(Dims d, Consts c, Work w, const double *__restrict__ u_all) {
    __shared__ double sh[4];
    __shared__ double seg[256];
    param_tables(d, c, w, d.b0 + blockIdx.x, u_all + (size_t)(d.b0 + blockIdx.x) * d.P, seg, sh);
}

// ---------------------------------------------------------------------------
// k_se: the S->E chain-binomial term and its eta/psi derivatives.
//   lambda_mt = exp(eta_mt) (I + psi W_t F_mt)/N_m + 1e-9      model_spec.py:258-266
//   ll = k log(1-exp(-r)) - (S-k) r,  r = lambda dt   (multiply_no_nan: k==0 drops the log)
// Tile = SE_TM rows x 64 days per workgroup; wave = SE_RW rows, lane = day: every plane is
// read as 512-B coalesced row segments.  All of a lane's loads are issued before any
// arithmetic and the SE_RW cells are independent instruction streams, which is what
// fills the fp64 pipe (a dependent fp64 op has ~32 cycles latency on gfx950).  Zero pads
// contribute exactly zero, so there are no bounds masks.
// ---------------------------------------------------------------------------
constexpr int SE_RS = 72;       // LDS row stride (doubles) of the row-sum transpose: conflict-free b64 reads

// Chain <-> XCD affinity (speed only, never correctness): workgroups are dealt round-robin over
// the 8 XCDs, so blocks L and L+8 share one.  With at most 8 chains in a launch every block of a
// chain gets an id with the same L % 8, and so do the chain's single-workgroup kernels (block
// id = chain): partial sums, proposal descriptors and the planes a chain's kernels hand to each
// other are then found in that XCD's L2 instead of at the cross-XCD rate.
// Maps the linear block id L of a 1-D grid of per*nb blocks to (chain, tile); the host only sets
// aff_nb when nb is 1, 2, 4 or a multiple of 8 (then several chains share an XCD, each still whole on one) and
// per*nb is a multiple of 8 (otherwise the natural 2-D grid).
__device__ __forceinline__ void xcd_affine(int L, int per, int nb, int &chain, int &tile) {
    const int l = L & 7;
    if (nb == 8) {
        chain = l;
        tile = L >> 3;
    } else if (nb > 8) {                 // a multiple of 8: chain mod 8 = block id mod 8
        chain = L % nb;
        tile = L / nb;
    } else {
        chain = l % nb;
        tile = (L >> 3) * (8 / nb) + l / nb;
    }
}
inline bool xcd_affinity_applies(int per, int nb) {
    return (nb == 1 || nb == 2 || nb == 4 || (nb > 0 && nb % 8 == 0)) && ((long long)per * nb) % 8 == 0;
}

// The S->E cells of one lane (NR rows, one day): rate r = (exp(a_t) exp(b_m)/N_m (I + psi W_t F) + floor) dt, L = log(1 - e^-r),
// inv = 1/expm1(r).  All NR cells go through the branch-free series with its literals in scalar registers (device_math.h:
// SeK) -- straight-line code, so the compiler interleaves the cells' dependent fp64 chains (~32 cycles of latency against 4 of
// issue) -- and a cell whose rate is outside the series' range (rare: daily hazards are 1e-5..1e-2) is redone by the full
// l1me_inv in a cold block.  One definition for every kernel that evaluates the term (k_se, k_se_chunk, k_leap): the
// sampler's launch forms must give the same bits.
// WANT_L = false (the inner leapfrog steps of k_leap: only the gradient of the term is needed): L is left at 0 and its
// chain -- the table logarithm, a third of a cell's instructions -- is not evaluated.
template <int NR, bool WANT_L = true>
__device__ __forceinline__ void se_cells(double ea_t, const double (&eb)[NR], const double (&I)[NR], double psiW, const double (&F)[NR],
                                         double rate_floor, double dt, const double2 *ltab, const SeK &sk,
                                         double (&ee)[NR], double (&lam0)[NR], double (&rr)[NR], double (&L)[NR], double (&inv)[NR]) {
    bool odd = false;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        ee[r] = ea_t * eb[r];
        lam0[r] = ee[r] * (I[r] + psiW * F[r]);
        rr[r] = (lam0[r] + rate_floor) * dt;
        l1me_inv_series_k(rr[r], L[r], inv[r], ltab, sk);
        if (!WANT_L) L[r] = 0.0;
        odd = odd || !(rr[r] >= L1ME_SERIES_MIN && rr[r] <= L1ME_SERIES_MAX);
    }
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (!(rr[r] >= L1ME_SERIES_MIN && rr[r] <= L1ME_SERIES_MAX)) l1me_inv(rr[r], L[r], inv[r], ltab);
    }
}

// L = log(1 - e^-r) of cells whose rates se_cells<NR, false> has formed: the end points of a trajectory inside k_leap
// (a cold block there; the same series and the same fallback as se_cells)
// (the arguments outside the series' range -- a daily hazard above 1/8: rare -- go through libm OUT OF LINE: inlined, exp and
// log cost ~40 registers in the middle of k_leap's step loop, whose tile workgroups hold their cells in registers across
// it; the allocator then parked loop-invariant cell registers in scratch and reloaded them in every step)
__device__ __attribute__((noinline)) double l1me_L_slow(double r) {
    const double e = exp(-r), om = 1.0 - e;           // l1me_inv's branch, operation for operation
    return log(om);
}
template <int NR>
__device__ __forceinline__ void se_cells_L(const double (&rr)[NR], const double2 *ltab, const SeK &sk, double (&L)[NR]) {
    bool odd = false;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double x = rr[r];
        const double x2 = x * x;
        L[r] = fast_log_k(x, ltab, sk) + x * (sk.mhalf + x * (sk.l1 - x2 * (sk.l2 - x2 * (sk.l3 - x2 * sk.l4))));
        odd = odd || !(x >= L1ME_SERIES_MIN && x <= L1ME_SERIES_MAX);
    }
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (!(rr[r] >= L1ME_SERIES_MIN && rr[r] <= L1ME_SERIES_MAX)) L[r] = l1me_L_slow(rr[r]);
    }
}

// TSM (sampler, GRAD, SRC 1): tile scalars for the chunked leapfrog -- 0 none, 1 column scalars,
// 2 column and row scalars (Work::TS); compile-time so that the plain kernel carries none of it.
template <bool GRAD, int SRC, int TSM>
__device__ __forceinline__ void se_tile(const Dims &d, const Consts &c, const Work &w, int bx, int by, int bz) {
    __shared__ double colbuf[4][WAVE];
    __shared__ double llbuf[4][WAVE], psibuf[4][WAVE];
    __shared__ double rowbuf[GRAD ? 4 * SE_RW * SE_RS : 1];
    __shared__ double rlbuf[4][WAVE], rsbuf[4][WAVE];
    __shared__ double2 ltab[LDSTAB_N];
    debug_skew(d);
    const int b = d.b0 + bz, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = bx * WAVE + lane;
    const int m0 = by * SE_TM + wave * SE_RW;
#ifdef SE_STAMPS
    // developer timeline (tools/dev/se_timeline.py): start / end of every tile workgroup in the two free tile scalars
    unsigned long long se_t0 = 0;
    if (GRAD && SRC == 1 && TSM == 1) se_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (threadIdx.x < LOGTAB_N) ltab[threadIdx.x] = c.logtab[threadIdx.x];   // barrier below, after the loads
    const double psi = w.scal[(size_t)b * NSCAL + SC_PSI];
    const double ea_t = w.ea[(size_t)b * d.Tp + t];
    const double Wt = c.W[t];
    const double psiW = psi * Wt;
    const size_t q0 = ((size_t)b * d.Mp + m0) * d.Tp + t;
    double F[SE_RW], I[SE_RW], kse[SE_RW], snk[SE_RW], eb[SE_RW];
#pragma unroll
    for (int r = 0; r < SE_RW; ++r) {
        const size_t q = q0 + (size_t)r * d.Tp;
        F[r] = w.F[q];
        eb[r] = w.eb[(size_t)b * d.Mp + m0 + r];
        if (SRC == 0) {
            const int2 ks = w.KS[q];
            I[r] = rint(w.Xn[q] * c.N[m0 + r]);
            kse[r] = (double)ks.x; snk[r] = (double)ks.y;
        } else {
            const int ki = w.K[0][q];
            I[r] = (double)w.St[2][q];
            kse[r] = (double)ki; snk[r] = (double)(w.St[0][q] - ki);
        }
    }
    // tile scalars for the chunked leapfrog: operands fetched with the other loads
    constexpr bool ts_on = GRAD && SRC == 1 && TSM != 0;
    constexpr bool ts_rows = GRAD && SRC == 1 && TSM == 2;
    double ts_vt = 0.0, ts_l[SE_RW], ts_s[SE_RW];
    // the wave's rows are the same for all its lanes: scalar loads, the values live in SGPRs
    const int m0u = by * SE_TM + __builtin_amdgcn_readfirstlane(wave) * SE_RW;
#pragma unroll
    for (int r = 0; r < SE_RW; ++r) {
        ts_l[r] = ts_rows ? c.la[m0u + r] : 0.0;
        ts_s[r] = ts_rows ? w.sp[((size_t)b * 2 + d.sp_par) * d.Mp + m0u + r] : 0.0;
    }
    if (ts_on) ts_vt = w.Vt[(size_t)b * d.Tp + t];
    __syncthreads();
    double ll = 0.0, gpsi = 0.0, colacc = 0.0, rlacc = 0.0, rsacc = 0.0;
    double *myrow = rowbuf + (GRAD ? wave * SE_RW * SE_RS : 0);
    // the wave's SE_RW cells side by side: rates, log(1 - e^-r) and 1/expm1(r) by the branch-free series (se_cells), then the sums
    SeK sk;
    sk.load();
    double ee[SE_RW], lam0[SE_RW], rr[SE_RW], L[SE_RW], inv[SE_RW];
    se_cells<SE_RW>(ea_t, eb, I, psiW, F, d.rate_floor, d.dt, ltab, sk, ee, lam0, rr, L, inv);
#pragma unroll
    for (int r = 0; r < SE_RW; ++r) {
        const bool has = kse[r] != 0.0;
        ll += (has ? kse[r] * L[r] : 0.0) - snk[r] * rr[r];
        if (GRAD) {
            const double gl = d.dt * ((has ? kse[r] * inv[r] : 0.0) - snk[r]);
            const double ge = gl * lam0[r];
            myrow[r * SE_RS + lane] = ge;
            colacc += ge;
            if (ts_rows) {                                // uniform
                rlacc = fma(ge, ts_l[r], rlacc);          // sum_m l_m (row sum)_m and sum_m s_m (row sum)_m, cell by cell
                rsacc = fma(ge, ts_s[r], rsacc);
            }
            gpsi += gl * ee[r] * Wt * F[r];
        }
    }
    llbuf[wave][lane] = ll;
    if (GRAD) {
        psibuf[wave][lane] = gpsi;
        colbuf[wave][lane] = colacc;
        // row sums: lane (r = lane>>3, s = lane&7) adds the 8 entries j*8+s of row r, then the
        // 8 lanes of a row combine by shuffles -- one LDS transpose instead of SE_RW wave reductions
        constexpr int LPR = WAVE / SE_RW;                 // lanes per row; each adds SE_RW entries
        static_assert(SE_RW == 4 || SE_RW == 8 || SE_RW == 16, "rows per wave");
        const int rr_ = lane / LPR, ss = lane % LPR;
        const double *src = myrow + rr_ * SE_RS + ss;
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < SE_RW; j += 2) v += src[j * LPR] + src[(j + 1) * LPR];
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, WAVE);
        if (ss == 0) w.Rpart[((size_t)b * d.ntc + bx) * d.Mp + m0 + rr_] = v;
        if (ts_rows) { rlbuf[wave][lane] = rlacc; rsbuf[wave][lane] = rsacc; }
    }
    __syncthreads();
    const size_t tile = (size_t)b * d.nmt * d.ntc + (size_t)by * d.ntc + bx;
    if (wave == 0) {
        const double v = wave_sum((llbuf[0][lane] + llbuf[1][lane]) + (llbuf[2][lane] + llbuf[3][lane]));
        if (lane == 0) w.Lpart[tile] = v;
    } else if (GRAD && wave == 1) {
        const double v = wave_sum((psibuf[0][lane] + psibuf[1][lane]) + (psibuf[2][lane] + psibuf[3][lane]));
        if (lane == 0) w.Ppart[tile] = v;
    } else if (GRAD && wave == 2) {
        const double cs = (colbuf[0][lane] + colbuf[1][lane]) + (colbuf[2][lane] + colbuf[3][lane]);
        w.Kpart[((size_t)b * d.nmt + by) * d.Tp + t] = cs;
        if (ts_on && ts_rows) {                            // TSM 2: wave 3 is busy with the row scalars
            const double bs = wave_sum(cs), as = wave_sum(cs * ts_vt);
            if (lane == 0) { w.TS[tile * 4 + 0] = bs; w.TS[tile * 4 + 1] = as; }
        }
    } else if (GRAD && wave == 3) {
        if (ts_on && !ts_rows) {                           // TSM 1: the column scalars beside wave 2's store
            const double cs = (colbuf[0][lane] + colbuf[1][lane]) + (colbuf[2][lane] + colbuf[3][lane]);
            const double bs = wave_sum(cs), as = wave_sum(cs * ts_vt);
            if (lane == 0) { w.TS[tile * 4 + 0] = bs; w.TS[tile * 4 + 1] = as; }
        }
        if (ts_rows) {
            const double rl = wave_sum((rlbuf[0][lane] + rlbuf[1][lane]) + (rlbuf[2][lane] + rlbuf[3][lane]));
            const double rs = wave_sum((rsbuf[0][lane] + rsbuf[1][lane]) + (rsbuf[2][lane] + rsbuf[3][lane]));
            if (lane == 0) { w.TS[tile * 4 + 2] = rl; w.TS[tile * 4 + 3] = rs; }
        }
    }
#ifdef SE_STAMPS
    if (GRAD && SRC == 1 && TSM == 1) {
        __syncthreads();                                  // every wave's stores issued and acknowledged
        if (threadIdx.x == 0) {
            unsigned xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned long long *ts = reinterpret_cast<unsigned long long *>(w.TS);
            ts[tile * 4 + 2] = se_t0;
            ts[tile * 4 + 3] = (__builtin_amdgcn_s_memrealtime() & 0xffffffffffffull) | ((unsigned long long)(xcc & 0xf) << 60) |
                               ((unsigned long long)((hw >> 8) & 0xfff) << 48);
        }
    }
#endif
}

template <bool GRAD, int SRC, int TSM = 0>
__global__ __launch_bounds__(256) void k_se(Dims d, Consts c, Work w) {
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (d.aff_nb > 0) {
        int tile;
        xcd_affine(blockIdx.x, d.ntc * d.nmt, d.aff_nb, bz, tile);
        bx = tile % d.ntc;
        by = tile / d.ntc;
    }
    se_tile<GRAD, SRC, TSM>(d, c, w, bx, by, bz);
}

// Stateless evaluation: the S->E tiles and, in the same launch, the fold of k_scan's per-day I->R
// partials (k_colreduce's blocks: they only feed k_finish) -- 1-D grid of n_se + ntc * nb blocks.
template <bool GRAD>
__global__ __launch_bounds__(256) void k_se_colreduce(Dims d, Consts c, Work w, int nb) {
    const int n_se = d.ntc * d.nmt * nb;
    int L = blockIdx.x;
    if (L >= n_se) {
        L -= n_se;
        colreduce_block(d, w, L % d.ntc, L / d.ntc);
        return;
    }
    int bx, by, bz;
    if (d.aff_nb > 0) {
        int tile;
        xcd_affine(L, d.ntc * d.nmt, d.aff_nb, bz, tile);
        bx = tile % d.ntc;
        by = tile / d.ntc;
    } else {
        bx = L % d.ntc;
        by = (L / d.ntc) % d.nmt;
        bz = L / (d.ntc * d.nmt);
    }
    se_tile<GRAD, 0, 0>(d, c, w, bx, by, bz);
}

// k_scan<0> and, as one more block per chain, the parameter tables (k_params: it depends on u only)
__global__ __launch_bounds__(SCAN_WAVES * WAVE) void k_scan_params(Dims d, Consts c, Work w, const double *__restrict__ events,
                                                                   const double *__restrict__ u_all) {
    if ((int)blockIdx.x == d.nrb_scan) {
        __shared__ double sh[4];
        __shared__ double seg[256];
        if (threadIdx.x >= 256) return;                     // param_tables is written for 256 threads
        param_tables(d, c, w, d.b0 + blockIdx.y, u_all + (size_t)(d.b0 + blockIdx.y) * d.P, seg, sh);
        return;
    }
    scan_rows<0>(d, c, w, events, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------
// The fused stateless evaluation (seir_log_prob_dev): three launches instead of four, and nothing on the
// critical path that the contraction does not need.
//   k_state_params  [state scan, PART 1 | parameter tables]          HBM-bound: events in, Xn / KS out
//   k_eval_tiles    [contraction tiles with the S->E term as their epilogue | row constants (scan PART 2) |
//                    fold of the scan's I->R partials]
//   k_finish        reduction (sums the row constants itself: with_rowconst)
// The contraction is bound by the matrix cores and leaves the vector ALUs idle; the row constants are bound by
// the vector ALUs and use no matrix core: both kinds of workgroup have 8 waves and at most 128 VGPRs, so one of
// each fits a CU (2 + 2 waves per SIMD) and the hardware interleaves them.  The S->E term is evaluated on the
// accumulators of the tile (D layout of v_mfma_f64_16x16x4_f64: lane l holds rows (l>>4)+4r, column l&15), so F
// makes no round trip through memory before it is used (it is still written once: seir_eval_prepared_dev reads it).
// The epilogue's operands (KS, Xn) are fetched behind the MFMAs of the last K chunk.
// Tile 64 rows x TN days (TN = 96 or 64), 8 waves = 4 row blocks of 16 x 2 column groups of TN/2.
// Partial sums: Lpart / Ppart per tile, Kpart [Mp/64][Tp], Rpart [Tp/TN][Mp] -- the Dims handed to this kernel
// and to k_finish carry nmt = Mp/64, ntc = Tp/TN.
// ---------------------------------------------------------------------------
// K chunk of the fused tile.  The panels are double-buffered in LDS and two chunks are in flight in registers, so a
// chunk's global loads are issued two chunks (~1.3 us of matrix work) before they are needed and there is one
// barrier per chunk; 16 keeps the dynamic LDS (which every workgroup of the launch is given, the row-constant
// ones too) at the ~54 KB of the epilogue's tile, so that two workgroups fit a CU.
constexpr int GSE_KC = 16;
template <int TN>
__host__ __device__ inline size_t gemm_se_lds_bytes() {
    const size_t panels = ((size_t)2 * GSE_KC * (GEMM_RS + gemm_rsb<TN>()) + 2) * sizeof(double);   // + the dummy slot
    const size_t epi = ((size_t)GEMM_TM * (TN + 2) + 4 * TN + 2 * GEMM_TM + 16) * sizeof(double);
    return panels > epi ? panels : epi;
}
template <bool GRAD, int TN>
__device__ __forceinline__ void gemm_se_tile(const Dims &d, const Consts &c, const Work &w, int bx, int by, int bz) {
    extern __shared__ double lds[];                 // 2 x (A [KC][RS] | B [KC][RSB]); the epilogue's tile and reduction buffers afterwards
    __shared__ double2 ltab[LDSTAB_N];
    __shared__ double vec_t[2 * TN], vec_m[2 * GEMM_TM];      // ea, W of the tile's days; eb, N of its rows
    constexpr int NT = 512, NCG = 2, RSB = gemm_rsb<TN>(), CW = TN / NCG, NJ = CW / 16;
    constexpr int PANEL = GSE_KC * (GEMM_RS + RSB);           // doubles per buffer
    debug_skew(d);
    const int b = d.b0 + bz, m0 = by * GEMM_TM, t0 = bx * TN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave / NCG, wc = wave % NCG;        // row block (16 rows), column group (CW columns)
    const double *Xb = w.Xn + (size_t)b * d.Mp * d.Tp;
    constexpr int EA = GSE_KC * (GEMM_TM / 2), EB = GSE_KC * (TN / 2);
    constexpr int NA = (EA + NT - 1) / NT, NB = (EB + NT - 1) / NT;
    static_assert(NA == 1 && NB <= 2, "staging is written out for one A and up to two B elements per thread");
    // Two chunks in flight, in named registers (as arrays behind a reference they ended up in scratch memory).  Every
    // chunk asked for lies inside the Mp allocated (zero-padded) rows: no bounds checks, no branches around the loads;
    // the partial second round of the B chunk reads a valid (clamped) element that store_chunk drops.
    using v2d = __attribute__((ext_vector_type(2))) double;     // native vector: HIP's double2 struct copies became memcpys through scratch
    struct Stage { v2d a, b0, b1; } s0, s1;
    const int ea_row = tid / (GEMM_TM / 2), ea_col = (tid % (GEMM_TM / 2)) * 2;
    const int eb0_row = tid / (TN / 2), eb0_col = (tid % (TN / 2)) * 2;
    const int e1 = min(tid + NT, EB - 1), eb1_row = e1 / (TN / 2), eb1_col = (e1 % (TN / 2)) * 2;
    const bool has_b1 = NB > 1 && tid + NT < EB;
    const double *pa = c.Cstar + (size_t)ea_row * d.Kp0 + m0 + ea_col;
    const double *pb0 = Xb + (size_t)eb0_row * d.Tp + t0 + eb0_col, *pb1 = Xb + (size_t)eb1_row * d.Tp + t0 + eb1_col;
#define GSE_LOAD(st, kb)                                                        \
    do {                                                                        \
        st.a = *(const v2d *)(pa + (size_t)(kb) * d.Kp0);                   \
        st.b0 = *(const v2d *)(pb0 + (size_t)(kb) * d.Tp);                  \
        if (NB > 1) st.b1 = *(const v2d *)(pb1 + (size_t)(kb) * d.Tp);      \
    } while (0)
#define GSE_STORE(st, buf)                                                      \
    do {                                                                        \
        double *A_ = lds + (buf) * PANEL, *B_ = A_ + GSE_KC * GEMM_RS;          \
        *(v2d *)(A_ + ea_row * GEMM_RS + ea_col) = st.a;                        \
        *(v2d *)(B_ + eb0_row * RSB + eb0_col) = st.b0;                         \
        if (NB > 1) *(v2d *)(has_b1 ? B_ + eb1_row * RSB + eb1_col : lds + 2 * PANEL) = st.b1;  \
    } while (0)
    d4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    const int ar = lane & 15, ak = lane >> 4;
    auto mfma_chunk = [&](int buf) {
        const double *A = lds + buf * PANEL, *Bm = A + GSE_KC * GEMM_RS;
#pragma unroll
        for (int kk = 0; kk < GSE_KC; kk += 4) {
            const double a0 = A[(kk + ak) * GEMM_RS + wr * 16 + ar];
            const double *bp = Bm + (kk + ak) * RSB + wc * CW + ar;
            double bf[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = bp[16 * j];
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bf[j], acc[j], 0, 0, 0);
        }
    };
    if (tid < LDSTAB_N) ltab[tid] = c.logtab[tid];
    // the tile's slices of the parameter tables: read from LDS by the epilogue instead of living in registers
    // through the matrix loop
    // (two disjoint thread ranges, each with its own loads: written as one `if / else if` over a common index the
    // compiler selected the base addresses through a two-entry array in scratch memory -- TN = 64)
    {
        const int it_ = tid - 256, im_ = tid - (256 + TN);
        if (it_ >= 0 && it_ < TN) {
            const double ea_ = w.ea[(size_t)b * d.Tp + t0 + it_], w_ = c.W[t0 + it_];
            vec_t[it_] = ea_;
            vec_t[TN + it_] = w_;
        }
        asm volatile("" ::: "memory");
        if (im_ >= 0 && im_ < GEMM_TM) {
            const double eb_ = w.eb[(size_t)b * d.Mp + m0 + im_], n_ = c.N[m0 + im_];
            vec_m[im_] = eb_;
            vec_m[GEMM_TM + im_] = n_;
        }
    }
    // K runs to Kp rounded up to a PAIR of chunks: Cstar and Xn are allocated and zero up to Mp = ceil64(M) rows,
    // so the extra rows add exact zeros
    const int nch = 2 * ((d.Kp + 2 * GSE_KC - 1) / (2 * GSE_KC));
    // (lds_barrier, not __syncthreads: the latter also waits for the global loads just issued)
    GSE_LOAD(s0, 0);
    GSE_LOAD(s1, GSE_KC);
    GSE_STORE(s0, 0);
    GSE_LOAD(s0, min(2, nch - 1) * GSE_KC);           // unconditional, like every load of the loop (see there)
    lds_barrier();
    int k = 0;
    for (; k + 2 < nch; k += 2) {
        // chunk k from buffer 0; chunk k + 1 goes to buffer 1 (free since the barrier that ended chunk k - 1)
        // (straight-line body -- no branch around a load or a store, the threads without a second B element store to
        // a dummy slot, the last iteration re-reads the last chunk -- so that the compiler's s_waitcnt counts stay exact
        // and a stage really is two chunks ahead)
        mfma_chunk(0);
        GSE_STORE(s1, 1);
        GSE_LOAD(s1, (k + 3) * GSE_KC);              // k + 3 <= nch - 1 inside this loop
        lds_barrier();
        // chunk k + 1 from buffer 1; chunk k + 2 goes to buffer 0
        mfma_chunk(1);
        GSE_STORE(s0, 0);
        GSE_LOAD(s0, min(k + 4, nch - 1) * GSE_KC);
        lds_barrier();
    }
    // the last pair of chunks, with the epilogue's first operands in flight behind it.  Cell (r, j) of this lane:
    // row m0 + wr*16 + ak + 4r, day t0 + wc*CW + j*16 + ar; the operands of column block 0 are fetched here,
    // block j + 1's while block j is evaluated
    mfma_chunk(0);
    GSE_STORE(s1, 1);
    lds_barrier();
    using v2i = __attribute__((ext_vector_type(2))) int;
    v2i ks[4];
    double xn[4];
    const int lr0 = wr * 16 + ak, lc0 = wc * CW + ar;          // local row / column of cell (0, 0)
    const size_t q00 = ((size_t)b * d.Mp + m0 + lr0) * d.Tp + t0 + lc0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ks[r] = *(const v2i *)(w.KS + q00 + (size_t)(4 * r) * d.Tp);
        xn[r] = w.Xn[q00 + (size_t)(4 * r) * d.Tp];
    }
    const double psi = w.scal[(size_t)b * NSCAL + SC_PSI];
    mfma_chunk(1);
    lds_barrier();                                   // the panels are dead: their LDS carries the tile and the reductions
    // The thread index of the epilogue's reductions, formed again from the wave's number (a scalar since before the matrix
    // loop) and the lane's position in the wave: kept alive through the loop it was the register pair the allocator parked
    // in scratch memory (a private segment for one store and one load per launch).
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int tid_e = wave_s * WAVE + lane_e;
    // The accumulators go to LDS and the cells are evaluated by a rolled loop over the column blocks: unrolled, the
    // twelve cells of a lane_e (each with its series / libm branch) need far more than the 128 registers that let a
    // row-constant workgroup share the CU.
    constexpr int FS = TN + 2;
    double *Ft = lds;                                // [64][FS]
    double *ep_col = lds + GEMM_TM * FS;             // [4 row blocks][TN]
    double *ep_sc = ep_col + 4 * TN;                 // [8 waves][2]
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ft[(lr0 + 4 * r) * FS + lc0 + 16 * j] = acc[j][r];
    // each lane_e reads back exactly what it wrote: no barrier
    double ll = 0.0, gpsi = 0.0;
    int nbad = 0;
    // Hot pass: every cell through the small-rate series, no branch in the cell code (cells interleave freely and the
    // libm branch's registers are not live here); a cell whose rate is outside the series' range contributes nothing
    // and is counted.  The d/d eta of a cell replaces its F in the LDS tile (F has been written out by then).
#pragma unroll 1
    for (int j = 0; j < NJ; ++j) {
        v2i ksn[4];
        double xnn[4];
        const bool more = j + 1 < NJ;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t q = q00 + (size_t)(4 * r) * d.Tp + 16 * (more ? j + 1 : j);
            ksn[r] = *(const v2i *)(w.KS + q);
            xnn[r] = w.Xn[q];
        }
        const double ea_t = vec_t[lc0 + 16 * j], Wt = vec_t[TN + lc0 + 16 * j];
        const double psiW = psi * Wt;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int li = (lr0 + 4 * r) * FS + lc0 + 16 * j;
            const double F = Ft[li];
            w.F[q00 + (size_t)(4 * r) * d.Tp + 16 * j] = F;
            const double I = rint(xn[r] * vec_m[GEMM_TM + lr0 + 4 * r]), kse = (double)ks[r].x, snk = (double)ks[r].y;
            const double ee = ea_t * vec_m[lr0 + 4 * r];
            const double lam0 = ee * (I + psiW * F);
            const double rr = (lam0 + d.rate_floor) * d.dt;
            const bool ok = rr >= L1ME_SERIES_MIN && rr <= L1ME_SERIES_MAX;      // false for NaN
            double L, inv;
            l1me_inv_series(ok ? rr : L1ME_SERIES_MAX, L, inv, ltab);
            const bool has = kse != 0.0;
            const double cll = (has ? kse * L : 0.0) - snk * rr;
            ll += ok ? cll : 0.0;
            if (GRAD) {
                const double gl = d.dt * ((has ? kse * inv : 0.0) - snk);
                Ft[li] = ok ? gl * lam0 : 0.0;
                gpsi += ok ? gl * ee * Wt * F : 0.0;
            }
            nbad += ok ? 0 : 1;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { ks[r] = ksn[r]; xn[r] = xnn[r]; }
    }
    // Cold pass, only in a wave_s that counted such a cell (large hazards, or the NaN of a negative rate): the cells
    // again, one at a time, those outside the range through l1me_inv's libm branch.
    if (__builtin_amdgcn_ballot_w64(nbad != 0) != 0) {
        // (the cell index formed again from the local row: held for this rare block, the 64-bit index of the hot pass was
        // the register pair parked in scratch memory in the 64-day instance)
        int lr0c = lr0, lc0c = lc0;
        asm volatile("" : "+v"(lr0c), "+v"(lc0c));
        const size_t q00c = ((size_t)b * d.Mp + m0 + lr0c) * d.Tp + t0 + lc0c;
#pragma unroll 1
        for (int cell = 0; cell < 4 * NJ; ++cell) {
            const int j = cell >> 2, r = cell & 3;
            const size_t q = q00c + (size_t)(4 * r) * d.Tp + 16 * j;
            const double F = w.F[q];                              // this thread's own store above
            const v2i k2 = *(const v2i *)(w.KS + q);
            const double Wt = vec_t[TN + lc0c + 16 * j];
            const double I = rint(w.Xn[q] * vec_m[GEMM_TM + lr0c + 4 * r]), kse = (double)k2.x, snk = (double)k2.y;
            const double ee = vec_t[lc0c + 16 * j] * vec_m[lr0c + 4 * r];
            const double lam0 = ee * (I + psi * Wt * F);
            const double rr = (lam0 + d.rate_floor) * d.dt;
            if (!(rr >= L1ME_SERIES_MIN && rr <= L1ME_SERIES_MAX)) {
                double L, inv;
                l1me_inv(rr, L, inv, ltab);
                const bool has = kse != 0.0;
                ll += (has ? kse * L : 0.0) - snk * rr;
                if (GRAD) {
                    const double gl = d.dt * ((has ? kse * inv : 0.0) - snk);
                    Ft[(lr0c + 4 * r) * FS + lc0c + 16 * j] = gl * lam0;
                    gpsi += gl * ee * Wt * F;
                }
            }
        }
    }
    ll = wave_sum(ll);
    if (GRAD) gpsi = wave_sum(gpsi);
    if (lane_e == 0) { ep_sc[wave_s * 2] = ll; ep_sc[wave_s * 2 + 1] = gpsi; }
    __syncthreads();
    const size_t tile = (size_t)b * d.nmt * d.ntc + (size_t)by * d.ntc + bx;
    if (tid_e == 0) {
        double a = 0.0, g = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a += ep_sc[k * 2]; g += ep_sc[k * 2 + 1]; }
        w.Lpart[tile] = a;
        if (GRAD) w.Ppart[tile] = g;
    }
    if (GRAD) {
        // column sums (d/d eta over the tile's 64 rows, per day) and row sums (over its TN days, per row) of the LDS tile
        {
            // all 512 threads: 8 lanes per row, each adds TN/8 entries; then the 8 lanes combine by shuffles
            const int i = tid_e >> 3, s8 = tid_e & 7;
            const double *src = Ft + i * FS + s8;
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < TN / 8; ++k) v += src[8 * k];
            v += __shfl_xor(v, 1, WAVE);
            v += __shfl_xor(v, 2, WAVE);
            v += __shfl_xor(v, 4, WAVE);
            if (s8 == 0) w.Rpart[((size_t)b * d.ntc + bx) * d.Mp + m0 + i] = v;
        }
        {
            // 4 row groups of 16 per column: TN x 4 threads (TN = 96: 384, TN = 64: 256), combined through LDS
            const int t = tid_e % TN, g4 = tid_e / TN;
            if (g4 < 4) {
                const double *src = Ft + (g4 * 16) * FS + t;
                double v0 = 0.0, v1 = 0.0;
#pragma unroll
                for (int k = 0; k < 16; k += 2) { v0 += src[k * FS]; v1 += src[(k + 1) * FS]; }
                ep_col[g4 * TN + t] = v0 + v1;
            }
            __syncthreads();
            if (tid_e < TN)
                w.Kpart[((size_t)b * d.nmt + by) * d.Tp + t0 + tid_e] =
                    (ep_col[tid_e] + ep_col[TN + tid_e]) + (ep_col[2 * TN + tid_e] + ep_col[3 * TN + tid_e]);
        }
    }
}

#undef GSE_LOAD
#undef GSE_STORE

// One launch for everything between the state and the final reduction: 1-D grid of
// [contraction + S->E tiles | row-constant blocks | I->R fold blocks]; the tiles have the low ids and are
// placed first, the vector-ALU work fills in beside them.
template <bool GRAD, int TN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_eval_tiles(Dims d, Consts c, Work w, const double *__restrict__ events, int nb) {
    const int per = d.ntc * d.nmt, n_g = per * nb, n_c = d.nrb_scan * nb;
    int L = blockIdx.x;
    if (L < n_g) {
        int bx, by, bz;
        if (d.aff_nb > 0) {
            int tile;
            xcd_affine(L, per, d.aff_nb, bz, tile);
            bx = tile % d.ntc;
            by = tile / d.ntc;
        } else {
            bx = L % d.ntc;
            by = (L / d.ntc) % d.nmt;
            bz = L / per;
        }
        gemm_se_tile<GRAD, TN>(d, c, w, bx, by, bz);
        return;
    }
    L -= n_g;
    if (L < n_c) {
        scan_rows<0, 2>(d, c, w, events, L % d.nrb_scan, L / d.nrb_scan);
        return;
    }
    L -= n_c;
    if (threadIdx.x >= 256) return;                         // colreduce_block is written for 256 threads
    const int ncb = d.Tp / WAVE;
    colreduce_block(d, w, L % ncb, L / ncb, /*with_const=*/false);
}
template <int TN>
inline size_t eval_tiles_lds_bytes() {
    const size_t panels = gemm_se_lds_bytes<TN>(), lft = (size_t)SCAN_LFT * sizeof(double);
    return panels > lft ? panels : lft;
}

// the state part of the scan and, as one more block per chain, the parameter tables
__global__ __launch_bounds__(SCAN_WAVES * WAVE) void k_state_params(Dims d, Consts c, Work w, const double *__restrict__ events,
                                                                    const double *__restrict__ u_all) {
    if ((int)blockIdx.x == d.nrb_scan) {
        __shared__ double sh[4];
        __shared__ double seg[256];
        if (threadIdx.x >= 256) return;                     // param_tables is written for 256 threads
        param_tables(d, c, w, d.b0 + blockIdx.y, u_all + (size_t)(d.b0 + blockIdx.y) * d.P, seg, sh);
        return;
    }
    scan_rows<0, 1>(d, c, w, events, blockIdx.x, blockIdx.y);
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
                                      double *lds_col, double *seg, double *sh, const double2 *ltab) {
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
            double L, inv;
            l1me_inv_wide(r, L, inv, ltab);              // I->R rates are ~0.25-0.5 per day: the 8-term series, not libm
            acc += (kir != 0.0 ? kir * L : 0.0) - dir * r;
            if (GRAD) {
                const double gr = d.dt * ((kir != 0.0 ? kir * inv : 0.0) - dir);
                gg0 += gr * rate;
                gg1 += gr * rate * c.wd[t];
                const double *kp = w.Kpart + (size_t)b * d.nmt * d.Tp + t;
                double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
                int ty = 0;
                for (; ty + 3 < d.nmt; ty += 4) {       // independent loads in flight
                    c0 += kp[(size_t)ty * d.Tp]; c1 += kp[(size_t)(ty + 1) * d.Tp];
                    c2 += kp[(size_t)(ty + 2) * d.Tp]; c3 += kp[(size_t)(ty + 3) * d.Tp];
                }
                for (; ty < d.nmt; ++ty) c0 += kp[(size_t)ty * d.Tp];
                col = (c0 + c1) + (c2 + c3);
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
    double total;
    double run = block_incl_suffix_scan_256(csum, seg, total) - csum;   // sum over later segments
    for (int t = t_hi - 1; t >= t_lo; --t) {
        run += lds_col[t];
        if (t >= 1) g[6 + t - 1] = run - at[t - 1] / (0.005 * 0.005);
    }
    double gsig = 0.0, gbeta = 0.0;
    for (int m = tid; m < d.M; m += 256) {
        const double *rp = w.Rpart + (size_t)b * d.ntc * d.Mp + m;
        double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
        int tx = 0;
        for (; tx + 3 < d.ntc; tx += 4) {
            r0 += rp[(size_t)tx * d.Mp]; r1 += rp[(size_t)(tx + 1) * d.Mp];
            r2 += rp[(size_t)(tx + 2) * d.Mp]; r3 += rp[(size_t)(tx + 3) * d.Mp];
        }
        for (; tx < d.ntc; ++tx) r0 += rp[(size_t)tx * d.Mp];
        const double R = (r0 + r1) + (r2 + r3);
        gsig += sp[m] * R;
        gbeta += c.la[m] * R;
        g[6 + d.T - 1 + m] = sig * R - w.Qs[(size_t)b * d.Mp + m];
    }
    double gpsi = 0.0;
    for (int i = tid; i < ntile; i += 256) gpsi += w.Ppart[(size_t)b * ntile + i];
    // the five scalar sums in one pass through LDS (seg is free again: the suffix scan is done)
    {
        double v5[5] = {gsig, gbeta, gpsi, gg0, gg1};
#pragma unroll
        for (int k = 0; k < 5; ++k) v5[k] = wave_sum(v5[k]);
        __syncthreads();
        if ((tid & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) seg[(tid >> 6) * 5 + k] = v5[k];
        }
        __syncthreads();
        gsig = (seg[0] + seg[5]) + (seg[10] + seg[15]);
        gbeta = (seg[1] + seg[6]) + (seg[11] + seg[16]);
        gpsi = (seg[2] + seg[7]) + (seg[12] + seg[17]);
        gg0 = (seg[3] + seg[8]) + (seg[13] + seg[18]);
        gg1 = (seg[4] + seg[9]) + (seg[14] + seg[19]);
    }
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
// with_rowconst (fused evaluation): the row constants were written in the previous launch and are summed here
// (and left in Work::constsum, as k_colreduce does in the other forms)
template <bool GRAD>
__device__ __forceinline__ void finish_chain(const Dims &d, const Consts &c, const Work &w, const double *__restrict__ u_all,
                                             double *__restrict__ logp, double *__restrict__ grad, int with_rowconst, int b) {
    extern __shared__ double lds_col[];             // [Tp]
    __shared__ double sh[4];
    __shared__ double seg[256];
    __shared__ double2 ltab[LDSTAB_N];
    double rcs = 0.0;
    if (with_rowconst)
        for (int m = threadIdx.x; m < d.M; m += 256) rcs += w.rowconst[(size_t)b * d.Mp + m];
    log_table_to_lds(ltab, c.logtab);
    const double lp = reduce_chain<GRAD>(d, c, w, b, u_all + (size_t)b * d.P,
                                         GRAD ? grad + (size_t)b * d.P : nullptr, lds_col, seg, sh, ltab);
    if (with_rowconst) {
        __syncthreads();
        rcs = block_sum_256(rcs, sh);
        if (threadIdx.x == 0) w.constsum[b] = rcs;
    }
    if (threadIdx.x == 0) {
        const double *sc = w.scal + (size_t)b * NSCAL;
        logp[b] = lp + (with_rowconst ? rcs : w.constsum[b]) + sc[SC_PRIOR] + sc[SC_JAC];
    }
}
template <bool GRAD>
__global__ __launch_bounds__(256) void k_finish(Dims d, Consts c, Work w, const double *__restrict__ u_all,
                                               double *__restrict__ logp, double *__restrict__ grad, int with_rowconst) {
    finish_chain<GRAD>(d, c, w, u_all, logp, grad, with_rowconst, d.b0 + blockIdx.x);
}

// ---------------------------------------------------------------------------
// k_eval_all: the whole stateless evaluation in ONE launch (chain b on XCD b mod 8 -- the XCD-local hand-off of
// sampler_kernels.h, k_se_chunk: producers' stores are in the XCD's L2 before they count themselves in, consumers
// have higher block ids, wait on the chain's counter and read nothing they or a neighbour on their CU could have read
// before it was written).  Block id mod 8 = chain in every segment (all segment sizes are multiples of 8):
//   [parameter block | tiles: their share of the state scan, then the contraction with the S->E epilogue |
//    row-constant blocks | I->R fold | finish]
//   parameter block, the tiles' state parts -> counter A -> tiles (X, KS, ea, eb), fold (per-block I->R partials)
//   tiles, row constants, fold -> counter B -> finish
// Against the three-launch form it saves two launch ramps and boundaries and the idle time between them; same
// arithmetic, same bits.  cnt: [chains][2 TAIL-like lines]: A at cnt[chain * 32], B at cnt[chain * 32 + 16].
// Any number of chains in the layout of the next multiple of 8 (Dims::aff_nb, nlive as in k_se_chunk); the host uses
// it while every tile workgroup of the launch can be resident (a tile waits for its chain's state while it holds a slot).
// ---------------------------------------------------------------------------
constexpr int EVC_STRIDE = 32;          // 64-bit words per chain: counters A and B in lines of their own
// (`first`: the workgroup's thread 0, found from the wave's number -- a scalar since the kernel's first instruction -- and the
// lane's position instead of threadIdx.x, which would otherwise be kept in a vector register through the tile's matrix loop
// just for this; there it was the value the allocator parked in scratch memory)
__device__ __forceinline__ void evc_arrive(unsigned long long *p_, int wave_s) {
    __syncthreads();                                   // vmcnt(0): this block's stores are in the XCD's L2
    const bool first = wave_s == 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u;
    if (first) __hip_atomic_fetch_add(p_, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void evc_wait(const unsigned long long *p_, unsigned long long target, int *err) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { atomicAdd(err, 1); break; }     // ~0.1 s: reported by the host, no hang
        }
    }
    __syncthreads();
}
template <bool GRAD, int TN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_eval_all(Dims d, Consts c, Work w, const double *__restrict__ events, const double *__restrict__ u_all,
                double *__restrict__ logp, double *__restrict__ grad, unsigned long long *cnt,
                unsigned long long targetA, unsigned long long targetB, int *err, int do_finish) {
#ifdef EVAL_STAMPS
#define ESTAMP(i) do { if (threadIdx.x == 0 && chain == 0 && tile_or0 == 0) cnt[NB * EVC_STRIDE + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ESTAMP(i) do {} while (0)
#endif
    const int NB = d.aff_nb;                                // chains of the layout: a multiple of 8
    const int wave_s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int per = d.ntc * d.nmt, ncb = d.Tp / WAVE;
    const int nP = NB, nT = per * NB, nC = d.nrb_scan * NB, nR = ncb * NB;
    int L = blockIdx.x;
    const int chain = L % NB;                               // every segment is a multiple of NB blocks long
    if (d.nlive > 0 && chain >= d.nlive) return;            // a chain of the layout that does not exist
    unsigned long long *cA = cnt + (size_t)chain * EVC_STRIDE, *cB = cA + EVC_STRIDE / 2;
    if (L < nP) {                                           // the chain's parameter tables
        __shared__ double sh[4];
        __shared__ double seg[256];
        if (threadIdx.x >= 256) return;                     // param_tables is written for 256 threads
        param_tables(d, c, w, d.b0 + chain, u_all + (size_t)(d.b0 + chain) * d.P, seg, sh);
        evc_arrive(cA, wave_s);
        return;
    }
    L -= nP;
    if (L < nT) {
        // a tile workgroup first scans its share of the chain's row blocks (the state part: X, KS, I->R partials) -- one
        // workgroup per CU from the first cycle, and no second tile can land on a CU that already carries one, which
        // separate state blocks ahead of the tiles made happen (the matrix work of such a CU doubles) -- then waits
        // for the whole chain's state and evaluates its tile
        const int tile = L / NB;
        const int tile_or0 = tile;
        ESTAMP(0);
        const int rbt = (d.nrb_scan + per - 1) / per;
        if (rbt == 2 && tile * 2 + 1 < d.nrb_scan && d.Tp <= SCAN_CB * WAVE) {
            scan_rows_state2(d, c, w, events, tile * 2, tile * 2 + 1, chain);      // both blocks' events in flight at once
        } else {
            for (int k = 0; k < rbt; ++k) {
                const int rb = tile * rbt + k;
                if (rb < d.nrb_scan) scan_rows<0, 1>(d, c, w, events, rb, chain);
                __syncthreads();                            // the next row block reuses the scan's LDS
            }
        }
        ESTAMP(1);
        evc_arrive(cA, wave_s);
        evc_wait(cA, targetA, err);
        ESTAMP(2);
        gemm_se_tile<GRAD, TN>(d, c, w, tile % d.ntc, tile / d.ntc, chain);
        ESTAMP(3);
        evc_arrive(cB, wave_s);
        return;
    }
    L -= nT;
    if (L < nC) {
        scan_rows<0, 2>(d, c, w, events, L / NB, chain);
        evc_arrive(cB, wave_s);
        return;
    }
    L -= nC;
    if (L < nR) {
        evc_wait(cA, targetA, err);
        if (threadIdx.x >= 256) return;                     // colreduce_block is written for 256 threads
        colreduce_block(d, w, L / NB, chain, /*with_const=*/false);
        evc_arrive(cB, wave_s);
        return;
    }
    if (!do_finish) return;
    const int tile_or0 = 0;
    ESTAMP(4);
    evc_wait(cB, targetB, err);
    ESTAMP(5);
    if (threadIdx.x >= 256) return;
    finish_chain<GRAD>(d, c, w, u_all, logp, grad, 1, d.b0 + chain);
    ESTAMP(6);
}
#undef ESTAMP
template <int TN>
inline size_t eval_all_lds_bytes(const Dims &d) {
    const size_t a = eval_tiles_lds_bytes<TN>(), b2 = (size_t)SCAN_WAVES * d.Tp * 2 * sizeof(double);
    return a > b2 ? a : b2;
}

}  // namespace seir
