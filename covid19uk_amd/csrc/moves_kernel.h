// Event-update proposal + finalisation kernel (k_move_pa2) of the Metropolis-within-Gibbs sweep.
//
// One MH event update = [finalize the previous proposal | draw the next one] (this kernel, grid
// (nrb_d, B)) followed by k_move_delta (log-likelihood ratio over the touched cells).  The proposal
// is a latency chain executed by block 0 of each chain, so it is written to minimise dependent
// global round trips: the m metapopulations of a proposal are processed together (their rows staged
// in LDS, one scan per row), uniforms are drawn by one lane each and broadcast through LDS, the
// per-metapopulation scalar work runs on one lane of different waves in parallel, barriers order
// LDS traffic only.  Semantics, RNG slots and the arithmetic of every term are those of the first
// implementation (retired in round 2; it lives on as oracle/mcmc_oracle.py), against which it was validated.
//
// Tried and dropped (r01): running the whole MultiScan phase as ONE persistent launch, one
// workgroup per chain.  It removes 40 launches per sweep but a single CU cannot carry the E->I
// band (M x <=84 days x m cells, 60-120 us per update against 11 us spread over 48 workgroups)
// and the fused kernel spilled; measured 1.22-1.34 ms per sweep against 1.07 ms for this split form.
// (Round 3's k_move_pairs, at the end of this file, is the persistent form that works: 27 workgroups per chain --
// three roles and the band -- resident for the sweep, a per-chain step barrier in place of the launch boundary.)
#pragma once
#include "sampler_kernels.h"

namespace seir {

#ifdef SEIR_STAMPS
#ifndef SEIR_STAMP_SLOT
#define SEIR_STAMP_SLOT 1
#endif
#define MSTAMP(i) do { if (threadIdx.x == 0 && stamp_on) ((unsigned long long *)(stamp_hs + 16))[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MSTAMP(i) do {} while (0)
#endif

// developer probe (tools/dev/pair_timeline.py): plain stores of the clock by thread 0 of chain 0's workgroups, per step of
// the pair launch: Chains::leap_st[(slot * 12 + step) * 16 + i]
#ifdef PAIR_STAMPS
#define QSTAMP(slot_, step_, i_) do { if (threadIdx.x == 0 && (b_stamp) == 0 && (step_) < 12 && (slot_) < 27) \
    ch.leap_st[((size_t)(slot_) * 12 + (step_)) * 16 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QSTAMP(slot_, step_, i_) do {} while (0)
#endif

constexpr int MVB = 512;              // threads
constexpr int MVW = MVB / WAVE;       // waves
constexpr int MVU = 4;                // cells in flight per thread in the band loops

struct MvShared {
    Move mv;
    int acc;
    int nsel;                         // rows chosen by the proposal (set by mv_propose)
    int sel[MMAX];
    int ired[MVW * 4];
    double dred[MVW * 2];
    double u[2 * MMAX][2];            // uniforms of draw slots 0..2*MMAX-1
    double logq_part[MMAX];
    int pend_valid[MMAX];
};

// min over the 64 lanes (result uniform): row_shr 1,2,4,8, row_bcast 15/31; min is idempotent, so
// the overlapping Hillis-Steele windows are harmless.  Lanes without a source receive INT_MAX.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_step(int v) {
    return min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ int wave_min_int(int v) {
    v = dpp_min_step<0x111, 0xf>(v);
    v = dpp_min_step<0x112, 0xf>(v);
    v = dpp_min_step<0x114, 0xf>(v);
    v = dpp_min_step<0x118, 0xf>(v);
    v = dpp_min_step<0x142, 0xa>(v);
    v = dpp_min_step<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ void mv_sum2(double &a, double &b2, double *sh) {
    a = wave_sum(a); b2 = wave_sum(b2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds_barrier();
    if (lane == 0) { sh[wave * 2] = a; sh[wave * 2 + 1] = b2; }
    lds_barrier();
    double x = 0.0, y = 0.0;
#pragma unroll
    for (int j = 0; j < MVW; ++j) { x += sh[j * 2]; y += sh[j * 2 + 1]; }
    a = x; b2 = y;
}

// arrays the proposal works from
struct MvLds {
    const int *rt;         // [M] row totals of the target transition's events (LDS copy, patched)
    int *rg;               // [M] events of the target transition inside the occult range (LDS)
#ifdef SEIR_STAMPS
    double *stamp_hs; bool stamp_on;
#endif
};

// log of a small positive integer-valued double (counts, bounds): table log; log 1 = 0 exactly
__device__ __forceinline__ double mv_log(double x, const double2 *ltab) { return x == 1.0 ? 0.0 : fast_log(x, ltab); }

// Uniforms and descriptor header of a proposal: they depend on (seed, chain, sweep, scan, slot)
// only, so the kernel draws them at entry, overlapped with its first round of loads; the first
// barrier after this call publishes them.
__device__ inline void mv_draw(const SamplerCfg &s, const Chains &ch, int b, MoveSpec spec, MvShared &sm, int T) {
    Move &mv = sm.mv;
    const int tid = threadIdx.x;
    if (tid >= 2 * MMAX && tid != 64) return;
    const RngKey key = rng_key(s, ch, b);
    const uint32_t stream = RS_MOVE_BASE + (uint32_t)(spec.scan * 4 + spec.slot);
    // one lane per draw slot (slot 15 = the accept uniform), broadcast through LDS
    if (tid < 2 * MMAX) rng_uniform2(key, stream, (uint32_t)tid, sm.u[tid][0], sm.u[tid][1]);
    if (tid == 64) {
        mv.valid = (s.disable_mask >> (1 + spec.slot)) & 1 ? 0 : 1;    // disabled sub-kernel: drawn, never accepted
        mv.n = 0; mv.tgt = spec.tgt; mv.kind = spec.kind; mv.slot = spec.slot;
        mv.logq = 0.0; mv.any_dI = 0; mv.LO = T; mv.HI = -1;
        for (int j = 0; j < MMAX; ++j) { mv.tm[j] = mv.tt[j] = mv.tdt[j] = mv.tx[j] = 0; mv.b[j] = -1; sm.pend_valid[j] = -1; }
        double ua, ub;
        rng_uniform2(key, stream, 15u, ua, ub);
        mv.logu = cold_log(ua);
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-level form of the proposal (T <= 64 NCH days): the same draws and the same arithmetic as the block
// form below, organised so that nothing in it waits at a workgroup barrier.  What the block form does with
// block scans through LDS (count the rows / days that hold events, find the r-th of them, minimum of a
// compartment over a window of days) one wave does with ballots, population counts and DPP minima over
// 64 lanes: a row of T days is NCH registers per lane, the M row totals are read from the LDS copy in
// 64-row pieces.  Wave j draws sub-move j (its row's three planes are fetched by that wave alone, one
// round trip); the waves meet once, where the block form's last barrier is.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wv_popc(unsigned long long m) { return __builtin_popcountll(m); }
// position (0..63) of the k-th set bit of `mask`, k < popcount(mask); every lane of the wave must be here
__device__ __forceinline__ int wv_kth_bit(unsigned long long mask, int k, int lane) {
    const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    const bool me = ((mask >> lane) & 1ull) != 0ull && below == k;
    return (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(me));
}
// rows of `tot` [0, M) that hold events, as bit masks of 64 rows each: all the LDS reads are issued together, the
// rest (count, position of the p-th such row) is scalar work on the masks
constexpr int WV_MCH = 6;                                    // 64-row pieces held as masks (M <= 384); more rows: loops
struct HotRows {
    unsigned long long mask[WV_MCH];
    int H;
};
__device__ __forceinline__ HotRows wv_hot_rows(const int *tot, int M, int lane) {
    HotRows h;
    h.H = 0;
    if (M <= WV_MCH * WAVE) {
        int v[WV_MCH];
#pragma unroll
        for (int c = 0; c < WV_MCH; ++c) { const int m = c * WAVE + lane; v[c] = m < M ? tot[m] : 0; }
#pragma unroll
        for (int c = 0; c < WV_MCH; ++c) { h.mask[c] = __builtin_amdgcn_ballot_w64(v[c] > 0); h.H += wv_popc(h.mask[c]); }
    } else {
#pragma unroll
        for (int c = 0; c < WV_MCH; ++c) h.mask[c] = 0ull;
        for (int m0 = 0; m0 < M; m0 += WAVE) {
            const int m = m0 + lane;
            h.H += wv_popc(__builtin_amdgcn_ballot_w64(m < M && tot[m] > 0));
        }
    }
    return h;
}
// row index of the p-th row that holds events, p < h.H
__device__ __forceinline__ int wv_hot_row(const HotRows &h, const int *tot, int M, int p, int lane) {
    int row = 0;
    bool found = false;
    if (M <= WV_MCH * WAVE) {
#pragma unroll
        for (int c = 0; c < WV_MCH; ++c) {
            const int cnt = wv_popc(h.mask[c]);
            if (!found && p < cnt) { row = c * WAVE + wv_kth_bit(h.mask[c], p, lane); found = true; }
            if (!found) p -= cnt;
        }
    } else {
        for (int m0 = 0; m0 < M; m0 += WAVE) {             // uniform trip count: the ballots see every lane
            const int m = m0 + lane;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(m < M && tot[m] > 0);
            const int cnt = wv_popc(mask);
            if (!found && p < cnt) { row = m0 + wv_kth_bit(mask, p, lane); found = true; }
            if (!found) p -= cnt;
        }
    }
    return row;
}
// value of the per-lane array v[c] (day c*64 + lane) at day t (uniform)
template <int NCH>
__device__ __forceinline__ int wv_at(const int (&v)[NCH], int t) {
    int r = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if ((t >> 6) == c) r = __builtin_amdgcn_readlane(v[c], t & 63);
    return r;
}

template <int NCH>
__device__ inline void mv_propose_wave(const Dims &d, const Work &w, const SamplerCfg &s, const Chains &ch, int b,
                                       MoveSpec spec, MvShared &sm, const MvLds &L, const double2 *ltab,
                                       bool rows_only) {
    Move &mv = sm.mv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, M = d.M, T = d.T;
#ifdef SEIR_STAMPS
    double *stamp_hs = L.stamp_hs; const bool stamp_on = L.stamp_on;
#endif
    MSTAMP(4);
    const int tgt = spec.tgt;
    lds_barrier();                                           // publishes sm.u / the header (mv_draw) and the totals in LDS
    if (spec.kind == 0) {
        // ---- UncalibratedEventTimesUpdate: every wave finds the rows (cheap, no exchange), wave j draws sub-move j
        const int *rt = L.rt;
        const HotRows hr = wv_hot_rows(rt, M, lane);
        const int H = hr.H;
        const int nsel = min(min(s.mmax, MMAX), H);
        // positions of the nsel distinct hot rows: draw j is an index among the H - j rows not yet chosen.  Written with
        // compile-time indices only (a run-time index would put these few integers in scratch memory, and every
        // access of this latency chain would be a memory round trip)
        int pos[MMAX], chosen[MMAX], rowj[MMAX];
#pragma unroll
        for (int j = 0; j < MMAX; ++j) { pos[j] = 0; rowj[j] = 0; chosen[j] = 0x7fffffff; }
#pragma unroll
        for (int j = 0; j < MMAX; ++j) {
            if (j < nsel) {
                int p = rng_index(sm.u[2 * j][0], H - j);
#pragma unroll
                for (int a = 0; a < MMAX; ++a)
                    if (a < j && p >= chosen[a]) ++p;            // chosen[0..j-1] ascending
                pos[j] = p;
                chosen[j] = p;                                   // insert: bubble down
#pragma unroll
                for (int a = MMAX - 1; a >= 1; --a)
                    if (a <= j && chosen[a - 1] > chosen[a]) { const int tmp = chosen[a]; chosen[a] = chosen[a - 1]; chosen[a - 1] = tmp; }
            }
        }
#pragma unroll
        for (int j = 0; j < MMAX; ++j)
            if (j < nsel) rowj[j] = wv_hot_row(hr, rt, M, pos[j], lane);
        if (rows_only) {
            if (tid == 0) {
                sm.nsel = nsel;
#pragma unroll
                for (int j = 0; j < MMAX; ++j)
                    if (j < nsel) sm.sel[j] = rowj[j];
            }
            lds_barrier();
            return;
        }
        MSTAMP(5);
        if (wave < nsel) {
            const int j = wave;
            int m = 0;
#pragma unroll
            for (int jj = 0; jj < MMAX; ++jj)
                if (jj == j) m = rowj[jj];
            const size_t rowoff = ((size_t)b * d.Mp + m) * d.Tp;
            int rk[NCH], rsrc[NCH], rdst[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int t = c * WAVE + lane;
                const bool in = t < T;                               // (T <= Tp <= 64 NCH)
                rk[c] = in ? w.K[tgt][rowoff + t] : 0;
                rsrc[c] = in ? w.St[tgt][rowoff + t] : 0;
                rdst[c] = in ? w.St[tgt + 1][rowoff + t] : 0;
            }
            MSTAMP(6);
            // day: the floor(u D)-th day with events
            int D = 0;
            unsigned long long hot[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) { hot[c] = __builtin_amdgcn_ballot_w64(rk[c] > 0); D += wv_popc(hot[c]); }
            int r = rng_index(sm.u[2 * j][1], D), t = 0;
            {
                bool found = false;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int cnt = wv_popc(hot[c]);
                    if (!found && r < cnt) { t = c * WAVE + wv_kth_bit(hot[c], r, lane); found = true; }
                    if (!found) r -= cnt;
                }
            }
            MSTAMP(7);
            const int v = rng_index(sm.u[2 * j + 1][0], 2 * s.dmax);
            const int delta = v < s.dmax ? v - s.dmax : v - s.dmax + 1;
            const int t2 = t + delta;
            if (t2 < 0 || t2 >= T) {                                 // out of range: the whole update is rejected
                if (lane == 0) {
                    sm.pend_valid[j] = 0;
                    mv.tm[j] = m; mv.tt[j] = t; mv.tdt[j] = delta; mv.tx[j] = 0;
                    sm.logq_part[j] = 0.0;
                }
            } else {
                const int lo = min(t, t2), hi = max(t, t2);
                const bool later = delta > 0;
                // bounds: min over (lo, hi] of the compartment that loses x, and of the one that gains it
                int mdec = 0x7fffffff, minc = 0x7fffffff;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int tau = c * WAVE + lane;
                    if (tau > lo && tau <= hi) {
                        mdec = min(mdec, later ? rdst[c] : rsrc[c]);
                        minc = min(minc, later ? rsrc[c] : rdst[c]);
                    }
                }
                mdec = wave_min_int(mdec);
                minc = wave_min_int(minc);
                MSTAMP(8);
                const int kt = wv_at<NCH>(rk, t), kt2 = wv_at<NCH>(rk, t2);
                if (lane == 0) {
                    const bool dec_unbounded = !later && tgt == 0;   // S: prev_event_id None -> no bound
                    const bool inc_unbounded = later && tgt == 0;
                    const int min_dec = dec_unbounded ? 0x7fffffff : mdec;
                    const int min_inc = minc;
                    const int xmax = max(0, min(min(s.nmax, kt), min_dec));
                    const int x = rng_index(sm.u[2 * j + 1][1], xmax + 1);
                    const int Dn = D - ((x > 0 && x == kt) ? 1 : 0) + ((x > 0 && kt2 == 0) ? 1 : 0);
                    const long long binc = inc_unbounded ? 0x7fffffffLL : (long long)min_inc + x;
                    const int xmax_r = (int)max(0LL, min((long long)min(s.nmax, kt2 + x), binc));
                    // a null sub-move (x == 0) is its own reverse: no correction (see oracle/mcmc_oracle.py,
                    // event_time_move, and tests/test_invariance*.py)
                    sm.logq_part[j] = x > 0 ? (-mv_log((double)Dn, ltab) - mv_log((double)(xmax_r + 1), ltab)) -
                                                  (-mv_log((double)D, ltab) - mv_log((double)(xmax + 1), ltab))
                                            : 0.0;
                    sm.pend_valid[j] = 1;
                    mv.m[j] = m; mv.a[j] = t; mv.b[j] = t2; mv.dka[j] = -x; mv.dkb[j] = x;
                    mv.lo[j] = lo; mv.hi[j] = hi;
                    mv.dsrc[j] = later ? x : -x;
                    mv.tm[j] = m; mv.tt[j] = t; mv.tdt[j] = delta; mv.tx[j] = x;
                }
            }
        }
        if (tid == 0) {
            sm.nsel = nsel;
#pragma unroll
            for (int j = 0; j < MMAX; ++j)
                if (j < nsel) sm.sel[j] = rowj[j];
        }
        lds_barrier();
        MSTAMP(9);
        if (tid == 0) {
            // compact the in-range updates (order preserved) and combine the correction
            int n = 0;
            double lq = 0.0;
            for (int j = 0; j < nsel; ++j) {
                if (sm.pend_valid[j] == 0) { mv.valid = 0; continue; }
                lq += sm.logq_part[j];
                if (n != j) {
                    mv.m[n] = mv.m[j]; mv.a[n] = mv.a[j]; mv.b[n] = mv.b[j]; mv.dka[n] = mv.dka[j];
                    mv.dkb[n] = mv.dkb[j]; mv.lo[n] = mv.lo[j]; mv.hi[n] = mv.hi[j]; mv.dsrc[n] = mv.dsrc[j];
                }
                mv.LO = min(mv.LO, mv.lo[n]); mv.HI = max(mv.HI, mv.hi[n]);
                if (tgt == 1 && mv.dsrc[n] != 0) mv.any_dI = 1;
                ++n;
            }
            mv.n = n;
            mv.logq = lq;
        }
    } else {
        // ---- UncalibratedOccultUpdate: one row, drawn by wave 0
        const int R = s.tr_hi - s.tr_lo;
        const int *rg = L.rg;
        int msel = 0;
        if (wave == 0 || rows_only) {
            const HotRows hr = wv_hot_rows(rg, M, lane);
            const int Hd = hr.H;
            const double u_br = sm.u[0][0], u_m = sm.u[0][1], u_t = sm.u[1][0], u_x = sm.u[1][1];
            const bool is_del = (u_br < 0.5) && Hd > 0;
            msel = is_del ? wv_hot_row(hr, rg, M, rng_index(u_m, Hd), lane) : rng_index(u_m, M);
            if (!rows_only) {
                MSTAMP(5);
                const int m = msel;
                const size_t rowoff = ((size_t)b * d.Mp + m) * d.Tp;
                int rk[NCH], rsrc[NCH], rdst[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = c * WAVE + lane;
                    const bool in = t < T;
                    rk[c] = in ? w.K[tgt][rowoff + t] : 0;
                    rsrc[c] = in ? w.St[tgt][rowoff + t] : 0x7fffffff;      // neutral in the minima below
                    rdst[c] = in ? w.St[tgt + 1][rowoff + t] : 0x7fffffff;
                }
                // the closed end of the series (state after the last day) is part of every window (t, T]
                const int end_src = comp_start(d, w, rowoff, tgt, T), end_dst = comp_start(d, w, rowoff, tgt + 1, T);
                MSTAMP(6);
                // hot days of the row inside the range; the day of a delete is the floor(u Dm)-th of them
                int Dm = 0;
                unsigned long long hot[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = c * WAVE + lane;
                    hot[c] = __builtin_amdgcn_ballot_w64(t >= s.tr_lo && t < s.tr_hi && rk[c] > 0);
                    Dm += wv_popc(hot[c]);
                }
                int t = s.tr_lo + rng_index(u_t, R);
                if (is_del) {
                    int r = rng_index(u_t, Dm);
                    bool found = false;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const int cnt = wv_popc(hot[c]);
                        if (!found && r < cnt) { t = c * WAVE + wv_kth_bit(hot[c], r, lane); found = true; }
                        if (!found) r -= cnt;
                    }
                }
                MSTAMP(7);
                int m0 = 0x7fffffff, m1 = 0x7fffffff;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int tau = c * WAVE + lane;
                    if (tau > t && tau < T) { m0 = min(m0, rsrc[c]); m1 = min(m1, rdst[c]); }
                }
                m0 = min(wave_min_int(m0), end_src);
                m1 = min(wave_min_int(m1), end_dst);
                MSTAMP(8);
                const int kt = wv_at<NCH>(rk, t);
                if (lane == 0) {
                    const int min_src = tgt == 0 ? 0x7fffffff : m0, min_dst = m1;
                    const int rt_m = rg[m];
                    const double lM = mv_log((double)M, ltab), lR = mv_log((double)R, ltab), l2 = 0.6931471805599453;
                    int x;
                    if (!is_del) {
                        const int xmax = max(0, min(s.occult_nmax, min_src));
                        x = rng_index(u_x, xmax + 1);
                        const double qf = (Hd > 0 ? -l2 : 0.0) - lM - lR - mv_log((double)(xmax + 1), ltab);
                        const int Hd2 = Hd + ((rt_m == 0 && x > 0) ? 1 : 0);
                        const int Dm2 = Dm + ((kt == 0 && x > 0) ? 1 : 0);
                        const long long bd = (long long)min_dst + x;
                        const int xmax_r = (int)max(0LL, min((long long)min(s.occult_nmax, kt + x), bd));
                        const double qr = (Hd2 > 0 && kt + x > 0)
                                              ? -l2 - mv_log((double)Hd2, ltab) - mv_log((double)Dm2, ltab) - mv_log((double)(xmax_r + 1), ltab)
                                              : -INFINITY;
                        mv.logq = qr - qf;
                        mv.dka[0] = x; mv.dsrc[0] = -x;
                    } else {
                        const int xmax = max(0, min(min(s.occult_nmax, kt), min_dst));
                        x = rng_index(u_x, xmax + 1);
                        const double qf = -l2 - mv_log((double)Hd, ltab) - mv_log((double)Dm, ltab) - mv_log((double)(xmax + 1), ltab);
                        const int Hd2 = Hd - ((x > 0 && rt_m == x) ? 1 : 0);
                        const long long bs = tgt == 0 ? 0x7fffffffLL : (long long)min_src + x;
                        const int xmax_r = (int)max(0LL, min((long long)s.occult_nmax, bs));
                        const double qr = (Hd2 > 0 ? -l2 : 0.0) - lM - lR - mv_log((double)(xmax_r + 1), ltab);
                        mv.logq = qr - qf;
                        mv.dka[0] = -x; mv.dsrc[0] = x;
                    }
                    mv.n = 1;
                    mv.m[0] = m; mv.a[0] = t; mv.b[0] = -1; mv.dkb[0] = 0;
                    mv.lo[0] = t; mv.hi[0] = T - 1;
                    mv.LO = t; mv.HI = T - 1;
                    mv.any_dI = (tgt == 1 && x != 0) ? 1 : 0;
                    mv.tm[0] = m; mv.tt[0] = t; mv.tdt[0] = is_del ? -1 : 1; mv.tx[0] = x;
                }
            }
        }
        if (tid == 0) { sm.nsel = 1; sm.sel[0] = msel; }
    }
    lds_barrier();
}

// rows_only: stop after the rows are chosen (sm.nsel, sm.sel[j] = row): what a proposal's rows are
// depends on the row totals and the uniforms only.
// NCH (compile time, chosen by the host from ceil64(T)): 64-day chunks a row is held in -- 6 (T <= 384: UK-380, NI-11),
// 12 (T <= 768: SYN-2048 x 730) or 16 (T <= 1024, the sampler's limit); the kernels are instantiated per value, so that
// an instance carries one size's code.  (Rounds 1-2 had a block-level form for T > 384 -- scans through LDS, ten
// workgroup barriers; the wave form replaced it at every size in round 3.)
template <int NCH>
__device__ __forceinline__ void mv_propose(const Dims &d, const Work &w, const SamplerCfg &s, const Chains &ch, int b,
                                           MoveSpec spec, MvShared &sm, const MvLds &L, const double2 *ltab,
                                           bool rows_only = false) {
    mv_propose_wave<NCH>(d, w, s, ch, b, spec, sm, L, ltab, rows_only);
}

// events of the target transition inside the occult range, for every row, into LDS (M beyond the prefetch width)
__device__ __forceinline__ void range_totals_to_lds(const Dims &d, const Work &w, const SamplerCfg &s, int b, int tgt,
                                                    int *rg) {
    for (int m = threadIdx.x; m < d.M; m += MVB) rg[m] = w.rngtot[((size_t)b * 2 + tgt) * d.Mp + m];
    __syncthreads();
}

// k_move_pa2: (1) finalize the pending proposal -- MetropolisHastings accept test
// (mcmc_kernel_factory.py:72,99), F band update by every block, row-local state update and trace
// by block 0 -- then (2) block 0 draws the next proposal.  next.kind < 0: finalize only.
// grid (nrb_d, B), MVB threads.  The pending descriptor is read from buffer pbuf, the next one
// written to pbuf^1 (late blocks must not see the new one).
// next.kind == -2: finalize only and advance the chain's sweep counter (closing launch of a sweep).
template <int NCH>
__global__ __launch_bounds__(MVB) void k_move_pa2(Dims d, Consts c, Work w, SamplerCfg s, Chains ch, MoveSpec next,
                                                  int have_prev, int pbuf) {
    extern __shared__ __attribute__((aligned(16))) int dyn_i[];                     // block 0: rg [M] | rt [M]
    __shared__ MvShared sm;
    __shared__ Move pend;
    __shared__ double2 ltab[LOGTAB_N];
    __shared__ int s_acc;
    __shared__ double s_dth, s_dcn;
    debug_skew(d);
    int bx = blockIdx.x, by = blockIdx.y;
    if (d.aff_nb > 0) xcd_affine(blockIdx.x, s.nrb_d, d.aff_nb, by, bx);
    const int b = d.b0 + by, tid = threadIdx.x;
    const int M = d.M, T = d.T;
    // ---- everything block 0 will need that does not depend on the accept decision is fetched
    // now, in the same round trip as the pending descriptor and the partial sums; what the
    // pending update changes (row totals, range totals) is patched after the decision.
    const bool proposer = bx == 0 && next.kind >= 0;
#ifdef SEIR_STAMPS
    double *stamp_hs = ch.hs + (size_t)b * NHS;
    const bool stamp_on = proposer && b == 0 && next.slot == SEIR_STAMP_SLOT && next.scan == 0;
#endif
    MSTAMP(0);
    constexpr int PRE_RT = 4;                                  // rows per thread (M <= 2048)
    const int R = s.tr_hi - s.tr_lo;
    const bool pre_rt = proposer && M <= PRE_RT * MVB;
    // kind 0: the row totals of the target plane; kind 1: each row's events inside the occult range
    int rt_pre[PRE_RT];
#pragma unroll
    for (int k = 0; k < PRE_RT; ++k) {
        const int m = tid + k * MVB;
        rt_pre[k] = 0;
        if (pre_rt && m < M) {
            if (next.kind == 0) {
                rt_pre[k] = w.rowtot[((size_t)b * 2 + next.tgt) * d.Mp + m];
            } else {
                rt_pre[k] = w.rngtot[((size_t)b * 2 + next.tgt) * d.Mp + m];
            }
        }
    }
    if (proposer) {
        if (tid >= 128 && tid < 128 + LOGTAB_N) ltab[tid - 128] = c.logtab[tid - 128];
        mv_draw(s, ch, b, next, sm, T);
    }
    double hs_th = 0.0, hs_cn = 0.0;
    unsigned tr_slot = 0xffffffffu;
    if (bx == 0 && have_prev && tid == 0) {
        const double *hs = ch.hs + (size_t)b * NHS;
        hs_th = hs[HS_LP_THETA]; hs_cn = hs[HS_LP_CONST];
        tr_slot = ch.sweep[b] - ch.slot0[0];
    }
    if (have_prev) {
        if (tid == 0) pend = ch.mv[(size_t)pbuf * s.B + b];
        double dth = 0.0, dcn = 0.0;
        for (int i = tid; i < s.nrb_d; i += MVB) {
            dth += ch.Dpart[((size_t)b * s.nrb_d + i) * 2];
            dcn += ch.Dpart[((size_t)b * s.nrb_d + i) * 2 + 1];
        }
        mv_sum2(dth, dcn, sm.dred);
        MSTAMP(1);
        if (tid == 0) {
            const double ratio = dth + dcn + pend.logq;
            s_acc = (pend.valid && pend.logu < ratio) ? 1 : 0;   // NaN -> reject
            s_dth = dth; s_dcn = dcn;
        }
        __syncthreads();
        const Move &mv = pend;
        if (s_acc && mv.any_dI) {
            // F[j][t] += sum_i Cstar[j][m_i] dI_i / N_{m_i}  on each update's day window
            const int rows_per_blk = (M + s.nrb_d - 1) / s.nrb_d;
            const int r_lo = bx * rows_per_blk, r_hi = min(M, r_lo + rows_per_blk);
            const int wave = tid >> 6, lane = tid & 63;
            for (int j = r_lo + wave; j < r_hi; j += MVW) {
                double coef[MMAX];
#pragma unroll
                for (int i = 0; i < MMAX; ++i)
                    coef[i] = i < mv.n ? c.Cstar[(size_t)mv.m[i] * d.Kp0 + j] * c.invN[mv.m[i]] * (double)(-mv.dsrc[i])
                                       : 0.0;
                double *Fr = w.F + ((size_t)b * d.Mp + j) * d.Tp;
                for (int t = mv.LO + lane; t <= mv.HI; t += WAVE) {
                    double dF = 0.0;
#pragma unroll
                    for (int i = 0; i < MMAX; ++i)
                        if (i < mv.n && t > mv.lo[i] && t <= mv.hi[i]) dF += coef[i];
                    if (dF != 0.0) Fr[t] += dF;
                }
            }
        }
        if (bx == 0) {
            if (s_acc) {
                for (int i = 0; i < mv.n; ++i) {
                    const size_t rowoff = ((size_t)b * d.Mp + mv.m[i]) * d.Tp;
                    const int src = mv.tgt, dst = mv.tgt + 1;
                    for (int t = mv.lo[i] + 1 + tid; t <= mv.hi[i]; t += MVB) {
                        w.St[src][rowoff + t] += mv.dsrc[i];
                        w.St[dst][rowoff + t] -= mv.dsrc[i];
                        if (mv.tgt == 1) w.Dir[(size_t)b * d.Tp + t] -= (double)mv.dsrc[i];
                    }
                    if (tid == 0) {
                        w.K[mv.tgt][rowoff + mv.a[i]] += mv.dka[i];
                        if (mv.b[i] >= 0) w.K[mv.tgt][rowoff + mv.b[i]] += mv.dkb[i];
                        w.rowtot[((size_t)b * 2 + mv.tgt) * d.Mp + mv.m[i]] += mv.dka[i] + mv.dkb[i];
                        const int ra = (mv.a[i] >= s.tr_lo && mv.a[i] < s.tr_hi) ? mv.dka[i] : 0;
                        const int rb = (mv.b[i] >= s.tr_lo && mv.b[i] < s.tr_hi) ? mv.dkb[i] : 0;
                        if (ra + rb != 0) w.rngtot[((size_t)b * 2 + mv.tgt) * d.Mp + mv.m[i]] += ra + rb;
                    }
                    __syncthreads();     // two updates may touch the same Dir[t]
                }
            }
            if (tid == 0) {
                double *hs = ch.hs + (size_t)b * NHS;
                if (s_acc) {
                    hs_th += s_dth; hs_cn += s_dcn;
                    hs[HS_LP_THETA] = hs_th; hs[HS_LP_CONST] = hs_cn;
                }
                if (tr_slot < (unsigned)s.cap) {
                    double *tr = ch.tr_mv + (((size_t)tr_slot * s.B + b) * 4 + mv.slot) * NMVTR;
                    tr[0] = (double)s_acc;
                    tr[1] = hs_th + hs_cn;
                    for (int j = 0; j < MMAX; ++j) {
                        tr[2 + j] = mv.tm[j]; tr[2 + MMAX + j] = mv.tt[j];
                        tr[2 + 2 * MMAX + j] = mv.tdt[j]; tr[2 + 3 * MMAX + j] = mv.tx[j];
                    }
                }
            }
            lds_barrier();               // LDS only: tid 0's trace stores need not drain before the proposal
        }
    }
    MSTAMP(2);
    // nothing in this launch reads the counter after block 0's trace write above
    if (bx == 0 && next.kind == -2 && tid == 0) ch.sweep[b] += 1;
    if (proposer) {
        MvLds L{};
        int *rtl = dyn_i + M;
        L.rt = rtl;
        L.rg = dyn_i;
        const bool patch = have_prev && s_acc && pend.tgt == next.tgt;
        if (next.kind == 0) {
            // row totals of the target plane: prefetched values (+ what the accepted update moved)
            if (pre_rt) {
#pragma unroll
                for (int k = 0; k < PRE_RT; ++k) {
                    const int m = tid + k * MVB;
                    if (m < M) rtl[m] = rt_pre[k];
                }
                lds_barrier();
                if (patch && tid == 0)
                    for (int i = 0; i < pend.n; ++i) rtl[pend.m[i]] += pend.dka[i] + pend.dkb[i];
            } else {
                __syncthreads();                              // block 0's own row-total stores above
                for (int m = tid; m < M; m += MVB) rtl[m] = w.rowtot[((size_t)b * 2 + next.tgt) * d.Mp + m];
            }
            lds_barrier();
        } else {
            if (pre_rt) {
#pragma unroll
                for (int k = 0; k < PRE_RT; ++k) {
                    const int m = tid + k * MVB;
                    if (m < M) L.rg[m] = rt_pre[k];
                }
                lds_barrier();
                if (patch && tid == 0)
                    for (int i = 0; i < pend.n; ++i) {
                        if (pend.a[i] >= s.tr_lo && pend.a[i] < s.tr_hi) L.rg[pend.m[i]] += pend.dka[i];
                        if (pend.b[i] >= s.tr_lo && pend.b[i] < s.tr_hi) L.rg[pend.m[i]] += pend.dkb[i];
                    }
                lds_barrier();
            } else {
                __syncthreads();                              // block 0's own K stores above
                range_totals_to_lds(d, w, s, b, next.tgt, L.rg);
            }
        }
        MSTAMP(3);
#ifdef SEIR_STAMPS
        L.stamp_hs = stamp_hs; L.stamp_on = stamp_on;
#endif
        mv_propose<NCH>(d, w, s, ch, b, next, sm, L, ltab);
        MSTAMP(10);
        if (tid == 0) ch.mv[(size_t)(pbuf ^ 1) * s.B + b] = sm.mv;
        MSTAMP(11);
    }
}

// ---------------------------------------------------------------------------------------------
// k_move_pair: one launch per PAIR of event updates (S->E-type, then E->I-type).
//
// An S->E update (event-time move or occult) changes only the rows it touches: no other row's
// force of infection moves, so its whole Metropolis-Hastings step -- propose, log-ratio over the
// updated rows, accept test, state update, trace -- is local to the workgroup that proposes it.
// Only the E->I updates need the chip (F moves on a band of days for every row).  The launch
// therefore does, in order:
//   (1) finalize the pending E->I-type proposal: accept test from k_move_delta's partial sums, rows,
//       trace; its F band is left to the next k_move_delta / k_apply_fpend (Chains::fpend)
//   (2) block 0: the complete S->E-type update `se`
//   (3) block 0: draw the E->I-type proposal `next` for the following k_move_delta
// which takes a scan from 8 launches to 4.  se.kind < 0: no S->E update; next.kind == -2: closing
// launch of the sweep (advance the counter).  Random streams, proposal arithmetic and the order
// of the four updates are those of k_move_pa2 / the oracle.
// ---------------------------------------------------------------------------------------------
constexpr int PRE_RT = 4;                                      // prefetched rows per thread (M <= 2048)

// row totals (kind 0) or per-row events inside the occult range (kind 1) of plane spec.tgt
__device__ __forceinline__ void mv_prefetch_rows(const Dims &d, const Work &w, const SamplerCfg &s, int b, MoveSpec spec,
                                                 bool on, int (&pre)[PRE_RT]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < PRE_RT; ++k) {
        const int m = tid + k * MVB;
        pre[k] = 0;
        if (on && m < d.M) {
            if (spec.kind == 0) {
                pre[k] = w.rowtot[((size_t)b * 2 + spec.tgt) * d.Mp + m];
            } else {
                pre[k] = w.rngtot[((size_t)b * 2 + spec.tgt) * d.Mp + m];
            }
        }
    }
}

// publish the prefetched values in LDS, corrected for an update of the same plane that was
// accepted after the prefetch (`fix` != nullptr); falls back to fresh loads when M is too large
__device__ __forceinline__ void mv_rows_to_lds(const Dims &d, const Work &w, const SamplerCfg &s, int b, MoveSpec spec,
                                               bool pre_ok, const int (&pre)[PRE_RT], const Move *fix, MvLds &L,
                                               int *rtl) {
    const int tid = threadIdx.x, M = d.M;
    int *dst = spec.kind == 0 ? rtl : L.rg;
    if (pre_ok) {
#pragma unroll
        for (int k = 0; k < PRE_RT; ++k) {
            const int m = tid + k * MVB;
            if (m < M) dst[m] = pre[k];
        }
        lds_barrier();
        if (fix && tid == 0)
            for (int i = 0; i < fix->n; ++i) {
                if (spec.kind == 0) {
                    dst[fix->m[i]] += fix->dka[i] + fix->dkb[i];
                } else {
                    if (fix->a[i] >= s.tr_lo && fix->a[i] < s.tr_hi) dst[fix->m[i]] += fix->dka[i];
                    if (fix->b[i] >= s.tr_lo && fix->b[i] < s.tr_hi) dst[fix->m[i]] += fix->dkb[i];
                }
            }
        lds_barrier();
    } else {
        __syncthreads();                                      // this block's own stores to the planes
        if (spec.kind == 0) {
            for (int m = tid; m < M; m += MVB) dst[m] = w.rowtot[((size_t)b * 2 + spec.tgt) * d.Mp + m];
            lds_barrier();
        } else {
            range_totals_to_lds(d, w, s, b, spec.tgt, L.rg);
        }
    }
}

// state update of an accepted proposal by the calling block (rows, events, row totals, I->R
// exposure); ends with the stores drained
// trans (optional, LDS, thread 0 writes): bit 0 set if a row total of the plane went from zero to non-zero or back,
// bit 1 the same for a total inside the occult range -- what decides whether a proposal pre-drawn from the old totals
// would still pick the same rows
__device__ __forceinline__ void mv_apply_rows(const Dims &d, const Work &w, const SamplerCfg &s, int b, const Move &mv,
                                              int *trans = nullptr) {
    const int tid = threadIdx.x;
    for (int i = 0; i < mv.n; ++i) {
        const size_t rowoff = ((size_t)b * d.Mp + mv.m[i]) * d.Tp;
        const int src = mv.tgt, dst = mv.tgt + 1;
        for (int t = mv.lo[i] + 1 + tid; t <= mv.hi[i]; t += MVB) {
            w.St[src][rowoff + t] += mv.dsrc[i];
            w.St[dst][rowoff + t] -= mv.dsrc[i];
            if (mv.tgt == 1) w.Dir[(size_t)b * d.Tp + t] -= (double)mv.dsrc[i];
        }
        if (tid == 0) {
            w.K[mv.tgt][rowoff + mv.a[i]] += mv.dka[i];
            if (mv.b[i] >= 0) w.K[mv.tgt][rowoff + mv.b[i]] += mv.dkb[i];
            int *prt = w.rowtot + ((size_t)b * 2 + mv.tgt) * d.Mp + mv.m[i];
            const int dk = mv.dka[i] + mv.dkb[i];
            const int ra = (mv.a[i] >= s.tr_lo && mv.a[i] < s.tr_hi) ? mv.dka[i] : 0;
            const int rb = (mv.b[i] >= s.tr_lo && mv.b[i] < s.tr_hi) ? mv.dkb[i] : 0;
            int *prg = w.rngtot + ((size_t)b * 2 + mv.tgt) * d.Mp + mv.m[i];
            if (trans != nullptr) {
                const int rt0 = *prt, rg0 = (ra + rb != 0) ? *prg : 1;
                int tb = 0;
                if ((rt0 > 0) != (rt0 + dk > 0)) tb |= 1;
                if (ra + rb != 0 && (rg0 > 0) != (rg0 + ra + rb > 0)) tb |= 2;
                *trans |= tb;
                *prt = rt0 + dk;
                if (ra + rb != 0) *prg = rg0 + ra + rb;
            } else {
                *prt += dk;
                if (ra + rb != 0) *prg += ra + rb;
            }
        }
        __syncthreads();     // two updates may touch the same cells
    }
}

// trace row of one update (thread 0)
__device__ __forceinline__ void mv_trace(const SamplerCfg &s, const Chains &ch, int b, const Move &mv, int acc,
                                         unsigned tr_slot, double lp) {
    if (tr_slot < (unsigned)s.cap) {
        double *tr = ch.tr_mv + (((size_t)tr_slot * s.B + b) * 4 + mv.slot) * NMVTR;
        tr[0] = (double)acc;
        tr[1] = lp;
        for (int j = 0; j < MMAX; ++j) {
            tr[2 + j] = mv.tm[j]; tr[2 + MMAX + j] = mv.tt[j];
            tr[2 + 2 * MMAX + j] = mv.tdt[j]; tr[2 + 3 * MMAX + j] = mv.tx[j];
        }
    }
}

// log-ratio of a drawn proposal over the rows it updates -> od[0..1] = {theta part, constant part}
__device__ __forceinline__ void mv_own_rows_to_down(const Dims &d, const Consts &c, const Work &w, int b, const Move &mv,
                                                    double psi, const double2 *ltab, const Move *fp, double *red,
                                                    double *od) {
    double dth = 0.0, dcn = 0.0;
    if (mv.valid && mv.n > 0) own_rows_delta<MVB>(d, c, w, b, mv, psi, 0, d.M, ltab, dth, dcn, fp);
    mv_sum2(dth, dcn, red);
    if (threadIdx.x == 0) {
        od[0] = dth;
        od[1] = dcn;
    }
}

// What the authoritative workgroup leaves for the next launch about the pre-drawn S->E-type proposal
// (Chains::prev, double-buffered by launch parity): whether the proposal may be used at all, and the rows
// the pending E->I-type update changed while the pre-draw was reading the planes.
struct PairNote {
    int ok;                           // 1: nothing in this launch invalidated the pre-drawn proposal outright
    int n, rows[MMAX];                // rows of the E->I-type update accepted in this launch
};
static_assert(sizeof(PairNote) == (2 + MMAX) * sizeof(int), "k_move_pair writes the note as 2 + MMAX ints");

// grid (R B) x MVB threads, R = 1..3 roles:
//   role 0, the authoritative workgroup of the chain: (1) finalize the pending E->I-type proposal,
//     (2) the S->E-type update `se` -- from the proposal and own-rows log-ratio PRE-DRAWN by role 2 of the
//     previous launch when nothing invalidated them, otherwise drawn here -- (3) certify the speculative
//     E->I proposal of role 1: its rows depend on the row totals and the uniforms only, so role 0
//     recomputes just those; the proposal is valid unless one of its rows was changed in (1) or (2) (then
//     role 0 draws it again from the final state into Chains::mvfix and sets Chains::mvsel).
//   role 1: draws the E->I-type proposal `next` from the state at entry, concurrently with role 0's work.
//     It has no side effect besides Chains::mv[pbuf^1] and Chains::Down.
//   role 2: draws the S->E-type proposal `se_next` of the NEXT launch from the state at entry, and its
//     log-ratio over the rows it updates (Chains::mvs / Chains::DownS[pbuf^1]).  An S->E-type proposal reads
//     only its own rows and the S->E plane's row (or occult-range) totals, so it is still the proposal the
//     next launch would draw unless (a) this launch's own S->E-type update is accepted AND turns a row total of
//     the kind the proposal reads on or off (which rows hold events is all a proposal takes from the totals,
//     besides the total of the row it picks), (b) one of its rows is a row of an update accepted in between
//     (this launch's S->E-type one, the pending E->I-type one of this launch or the one finalized at the start
//     of the next), or (c) this workgroup was late.
//     (a)-(c) are all decided by role 0 without looking at role 2's output; the next launch then draws the
//     update itself, exactly as before.  If the E->I-type update finalized at the start of the next launch
//     is accepted, F moved under the proposal's rows and only the log-ratio is recomputed.
//   What the speculative roles read is of two kinds: rows of the planes -- if role 0 writes one of them it
//     is a row of a conflict, which role 0 detects on its own -- and the planes' row/range totals, fetched
//     at entry: role 0 does not write anything before roles 1 and 2 hold them (one-word handshakes,
//     Chains::hand / hand2; if a role is not there in time role 0 goes on and discards its output).
// ---------------------------------------------------------------------------------------------
// Band workgroups of k_move_pair (nband > 0): k_move_delta's work inside the pair launch.  The launch then carries
// the whole MH step of the E->I-type proposal except its accept test, and the ten k_move_delta launches of a sweep
// (7.7 us each, 2.5 of them launch ramp and boundary) are gone.  A band workgroup (8 waves, rows bx*rpb ..) waits
// for the authoritative role's done-token (the proposal is certified, Chains::mvsel says which descriptor stands,
// Chains::fpend is this launch's), evaluates the band part of the log-ratio over its rows with the F band of the
// update accepted in this launch added ON THE FLY (as the roles do), writes its partial sums, then waits for the
// speculative roles' tokens -- nobody reads F any more -- and applies that F band to its rows.
// Hand-off as in k_se_chunk: XCD-local (tokens in the chain's own cache line, stores acknowledged by the shared L2
// before a token is written, everything another workgroup of this launch may have written is read past the L1), no
// agent-scope fence; the host launches band workgroups only where the XCC_ID probe allows it.  Band workgroups have
// the highest block ids: they are placed after every role, so a waiting one never holds a slot a role needs.
// ---------------------------------------------------------------------------------------------
template <typename TT>
__device__ __forceinline__ TT ld_l2(const TT *p_) { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void move_copy_l2(Move *dst, const Move *src, int t0) {
    const int i = (int)threadIdx.x - t0;
    if (i >= 0 && i < MOVE_DW) reinterpret_cast<int *>(dst)[i] = ld_l2(reinterpret_cast<const int *>(src) + i);
}
__device__ __forceinline__ void wait_token(const unsigned *p_, unsigned token, unsigned *late) {
    int spins = 0;
    while (ld_l2(p_) != token) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        // once ANY wait of the chain has timed out (its workgroups cannot all have been placed: something else holds part of
        // the chip) every later wait of the chain gives up at its first look at the counter, every 256 polls, so that a burst
        // that cannot complete drains in about a second instead of a second per wait (as leap_wait); the host finds the
        // counter at the next read of the trace, fails loudly and can restore the last snapshot (seir_sampler_restore)
        if ((spins & 255) == 0 && ld_l2(late) != 0u) break;
        if (spins > (1 << 22)) { __hip_atomic_fetch_add(late, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // counted, no hang
    }
}
// Role 0's last token of a step also says which descriptor stands (Chains::mvsel): the top bit -- one round trip less for the
// band workgroups than the token and then the word (a token is sweep * 64 + step + 1: 31 bits hold 33 million sweeps)
__device__ __forceinline__ unsigned wait_token_flag(const unsigned *p_, unsigned token, unsigned *late) {
    int spins = 0;
    unsigned v;
    while (((v = ld_l2(p_)) & 0x7fffffffu) != token) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if ((spins & 255) == 0 && ld_l2(late) != 0u) break;
        if (spins > (1 << 22)) { __hip_atomic_fetch_add(late, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    return v >> 31;
}
// A proposal descriptor as hand-off words (sampler_kernels.h): 60 dwords in 30 words {dword, token, dword, token}.  The band
// workgroups used to wait for role 1's token (after its stores were acknowledged) and then copy the descriptor -- two round
// trips behind each other; now lanes 0..29 of one wave look at their word until it shows the launch's token.
constexpr int MOVE_LLW = (MOVE_DW + 1) / 2;
static_assert(MOVE_LLW <= 32, "Chains::llmv holds 32 words per descriptor");
__device__ __forceinline__ void move_store_ll(uint4 *dst, const Move *src_lds, int lane, unsigned token) {
    if (lane < MOVE_LLW) {
        const int *sd = reinterpret_cast<const int *>(src_lds);
        uint4 x;
        x.x = (unsigned)sd[2 * lane]; x.y = token;
        x.z = 2 * lane + 1 < MOVE_DW ? (unsigned)sd[2 * lane + 1] : 0u; x.w = token;
        dst[lane] = x;
    }
}
__device__ __forceinline__ void move_wait_ll(Move *dst_lds, const uint4 *src, int lane, unsigned token, unsigned *late) {
    // (one wave; every lane looks at a word -- the lanes beyond the descriptor at its last one)
    const uint4 *pp[1] = {src + min(lane, MOVE_LLW - 1)};
    u32x4 x[1];
    int spins = 0;
    for (;;) {
        ll_load<1>(pp, x);
        if (__builtin_amdgcn_ballot_w64(!ll_ok(x[0], token)) == 0ull) break;
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if ((spins & 255) == 0 && ld_l2(late) != 0u) break;
        if (spins > (1 << 22)) { if (lane == 0) __hip_atomic_fetch_add(late, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    if (lane < MOVE_LLW) {
        int *dd = reinterpret_cast<int *>(dst_lds);
        dd[2 * lane] = (int)x[0].x;
        if (2 * lane + 1 < MOVE_DW) dd[2 * lane + 1] = (int)x[0].z;
    }
}
// SOLO (k_move_pairs): the workgroup has its CU to itself and its L1 was emptied when the step began, so what it loads of
// the planes AFTER the token that declares them final cannot be an older copy: plain loads, which a wave issues back to
// back, where the launch-per-pair form (other workgroups of the chain may share the CU and its L1) reads past the L1
template <bool SOLO, typename TT>
__device__ __forceinline__ TT ld_band(const TT *p_) { return SOLO ? *p_ : ld_l2(p_); }
// NRB: rows per wave of a band workgroup -- 2 (16 rows per workgroup: 24 of them per chain at UK-380, one chain per XCD) or 4 (32
// rows: 12 per chain, so that (3 + 12) x 16 workgroups -- sixteen chains, two per XCD -- still hold one CU each)
template <bool SOLO, int NRB>
__device__ __forceinline__ void pair_band_block(const Dims &d, const Consts &c, const Work &w, const SamplerCfg &s,
                                                const Chains &ch, int b, int bx, int nband, unsigned token, bool has_r1,
                                                bool has_r2, int buf, int st_slot = 0, int st_step = 0) {
    const int b_stamp = b - d.b0;
    (void)b_stamp;
    __shared__ Move mvA, mvB, fp;
    __shared__ int mv_sel;
    __shared__ double sh_th[MVW], sh_cn[MVW];
    __shared__ double2 ltab[LDSTAB_N];
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                          // (see pair_step)
    const int wave = tid >> 6, lane = tid & 63;
    if (tid < LDSTAB_N) ltab[tid] = c.logtab[tid];
    const double psi = w.scal[(size_t)b * NSCAL + SC_PSI];      // constant during the event updates
    const unsigned *done = ch.done + (size_t)b * 2 * TAIL_STRIDE;
    const int rpb = (d.M + nband - 1) / nband;
    const int r_lo = bx * rpb, r_hi = min(d.M, r_lo + rpb);
    // ---- the update accepted in this launch (token 4, early in the authoritative role): its descriptor, and what its F
    // band needs for this workgroup's rows -- coefficients and the F values themselves (nobody writes F but this code)
#ifndef PAIR_BAND_BACKOFF
#define PAIR_BAND_BACKOFF 0
#endif
    if (tid == 0) {
        if (PAIR_BAND_BACKOFF > 0) __builtin_amdgcn_s_sleep(PAIR_BAND_BACKOFF);   // (role 0 decides the pending update ~3.5 us into a step)
        wait_token(done + 4, token, ch.late + ch.late_fatal + b);
    }
    QSTAMP(st_slot, st_step, 1);
    __syncthreads();
    move_copy_l2(&fp, ch.fpend + b, 128);
    __syncthreads();
    const bool has_fp = fp.valid == 1;
    constexpr int NPF = 2;                             // 64-day pieces of a hull (dmax <= 127); rows per wave: NRB (rpb <= 8 NRB)
    // (prefetched for two rows per wave; with four the sixteen coefficients and eight F values per lane held across the
    // evaluation put the persistent launch into scratch: there they are formed where they are used)
    constexpr bool PREF = NRB == 2;
    constexpr int NRP = PREF ? NRB : 1;
    double cfb[NRP][MMAX], Fpre[NRP][NPF];
#pragma unroll
    for (int r = 0; r < NRP; ++r) {
        const int j = r_lo + wave + r * MVW;
        const bool onr = PREF && has_fp && j < r_hi;
#pragma unroll
        for (int i = 0; i < MMAX; ++i)
            cfb[r][i] = (onr && i < fp.n) ? c.Cstar[(size_t)fp.m[i] * d.Kp0 + j] * c.invN[fp.m[i]] * (double)(-fp.dsrc[i]) : 0.0;
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int t = fp.LO + q * WAVE + lane;
            Fpre[r][q] = (onr && t <= fp.HI) ? ld_band<SOLO>(w.F + ((size_t)b * d.Mp + j) * d.Tp + t) : 0.0;
        }
    }
    const bool pre_ok = PREF && (!has_fp || fp.HI - fp.LO < NPF * WAVE);   // a longer hull (dmax > 127): the loop at the end reloads
    // ---- the proposal to evaluate.  The planes are final once the authoritative role has applied its own updates (token
    // 5), a little before it has certified the speculative proposal; the band is evaluated for that proposal at once
    // and, in the rare launch in which a row conflict made the authoritative role draw it again (Chains::mvsel, known
    // with the last token), evaluated again for the re-drawn one.
    if (tid == 0) wait_token(done + 5, token, ch.late + ch.late_fatal + b);
    if (has_r1) {
        // the speculative role publishes its descriptor as hand-off words well before it is done (token 1): wave 1 looks for
        // them while thread 0 waits for the planes
        if (wave == 1) move_wait_ll(&mvA, ch.llmv + ((size_t)buf * s.B + b) * 32, lane, token, ch.late + ch.late_fatal + b);
        QSTAMP(st_slot, st_step, 2);
        __syncthreads();
    } else {
        QSTAMP(st_slot, st_step, 2);
        __syncthreads();
        move_copy_l2(&mvA, ch.mv + (size_t)buf * s.B + b, 0);
        __syncthreads();
    }
    auto evaluate = [&](const Move &mv) -> double {
        double dth = 0.0;
        if (mv.valid && mv.n > 0 && mv.any_dI) {
            // a wave's rows two by two, side by side: the coefficients of both, then per 64-day piece of the hull
            // the loads of both before any arithmetic -- half the dependent round trips of one row after the other
            const double *ea = w.ea + (size_t)b * d.Tp;
            constexpr int NR = 2;
    #pragma unroll 1
            for (int r0 = 0; r0 < NRB; r0 += NR) {
            int jr[NR];
            bool on[NR];
            double eb[NR], coef[NR][MMAX], cfp[NR][MMAX];
    #pragma unroll
            for (int r = 0; r < NR; ++r) {
                jr[r] = r_lo + wave + (r0 + r) * MVW;
                bool mine = false;
    #pragma unroll
                for (int i = 0; i < MMAX; ++i) mine |= (i < mv.n && mv.m[i] == jr[r]);
                on[r] = jr[r] < r_hi && !mine;           // wave-uniform: the updated rows' part is the drawing role's (Chains::Down)
                const int j = on[r] ? jr[r] : r_lo;
                eb[r] = w.eb[(size_t)b * d.Mp + j];
    #pragma unroll
                for (int i = 0; i < MMAX; ++i) {
                    coef[r][i] = (on[r] && i < mv.n) ? c.Cstar[(size_t)mv.m[i] * d.Kp0 + j] * c.invN[mv.m[i]] * (double)(-mv.dsrc[i]) : 0.0;
                    // the accepted update's coefficients: prefetched above for the same rows (two rows per wave), or formed here
                    if (PREF) cfp[r][i] = on[r] ? cfb[PREF ? r : 0][i] : 0.0;
                    else cfp[r][i] = (has_fp && on[r] && i < fp.n) ? c.Cstar[(size_t)fp.m[i] * d.Kp0 + j] * c.invN[fp.m[i]] * (double)(-fp.dsrc[i]) : 0.0;
                }
            }
            for (int t0 = mv.LO; t0 <= mv.HI; t0 += WAVE) {
                const int t = t0 + lane;
                double dF[NR], S[NR], I[NR], kse[NR], F[NR];
                bool act[NR];
    #pragma unroll
                for (int r = 0; r < NR; ++r) {
                    dF[r] = 0.0;
    #pragma unroll
                    for (int i = 0; i < MMAX; ++i)
                        if (i < mv.n && t > mv.lo[i] && t <= mv.hi[i]) dF[r] += coef[r][i];
                    act[r] = on[r] && t <= mv.HI && dF[r] != 0.0;
                    const size_t q = ((size_t)b * d.Mp + (on[r] ? jr[r] : r_lo)) * d.Tp + (act[r] ? t : mv.LO);
                    S[r] = (double)ld_band<SOLO>(w.St[0] + q); I[r] = (double)ld_band<SOLO>(w.St[2] + q);
                    kse[r] = (double)ld_band<SOLO>(w.K[0] + q); F[r] = ld_band<SOLO>(w.F + q);
                }
    #pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (!act[r]) continue;
                    if (has_fp) {                          // summed first, as apply_f_band does
                        double dFp = 0.0;
    #pragma unroll
                        for (int i = 0; i < MMAX; ++i)
                            if (i < fp.n && t > fp.lo[i] && t <= fp.hi[i]) dFp += cfp[r][i];
                        if (dFp != 0.0) F[r] += dFp;
                    }
                    dth += band_delta(S[r], I[r], kse[r], F[r], dF[r], ea[t] * eb[r], psi * c.W[t], d.rate_floor * d.dt, d.dt, ltab);
                }
            }
            }
        }
        return dth;
    };
    double dth = evaluate(mvA);
    QSTAMP(st_slot, st_step, 3);
    if (tid == 0) mv_sel = (int)wait_token_flag(done + 0, token, ch.late + ch.late_fatal + b);   // (Chains::mvsel rides on the token)
    __syncthreads();
    if (mv_sel) {                                          // uniform, rare
        move_copy_l2(&mvB, ch.mvfix + (size_t)buf * s.B + b, 64);
        __syncthreads();
        dth = evaluate(mvB);
    }
    dth = wave_sum(dth);
    if (lane == 0) sh_th[wave] = dth;
    __syncthreads();
    if (tid == 0) {
        double *out = ch.Dpart + ((size_t)b * nband + bx) * 2;
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < MVW; ++k) a += sh_th[k];
        out[0] = a;
        out[1] = 0.0;
        QSTAMP(st_slot, st_step, 4);
        // the F band: once nobody reads F any more
        if (has_fp) {
            if (has_r1) wait_token(done + 1, token, ch.late + ch.late_fatal + b);
            if (has_r2) wait_token(done + 2, token, ch.late + ch.late_fatal + b);
        }
        QSTAMP(st_slot, st_step, 5);
    }
    if (!has_fp) return;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NRB; ++r) {
        const int j = r_lo + wave + r * MVW;
        if (j >= r_hi) continue;
        double *Fr = w.F + ((size_t)b * d.Mp + j) * d.Tp;
        double cfr[MMAX];                                  // this row's coefficients: the prefetched ones, or formed now
#pragma unroll
        for (int i = 0; i < MMAX; ++i) {
            if (PREF) cfr[i] = cfb[PREF ? r : 0][i];
            else cfr[i] = i < fp.n ? c.Cstar[(size_t)fp.m[i] * d.Kp0 + j] * c.invN[fp.m[i]] * (double)(-fp.dsrc[i]) : 0.0;
        }
        if (pre_ok) {
#pragma unroll
            for (int q = 0; q < NPF; ++q) {
                const int t = fp.LO + q * WAVE + lane;
                double dFp = 0.0;
#pragma unroll
                for (int i = 0; i < MMAX; ++i)
                    if (i < fp.n && t > fp.lo[i] && t <= fp.hi[i]) dFp += cfr[i];
                if (t <= fp.HI && dFp != 0.0) Fr[t] = Fpre[PREF ? r : 0][q] + dFp;
            }
        } else {
            for (int t = fp.LO + lane; t <= fp.HI; t += WAVE) {
                double dFp = 0.0;
#pragma unroll
                for (int i = 0; i < MMAX; ++i)
                    if (i < fp.n && t > fp.lo[i] && t <= fp.hi[i]) dFp += cfr[i];
                if (dFp != 0.0) Fr[t] = ld_band<SOLO>(Fr + t) + dFp;
            }
        }
    }
    (void)sh_cn;
}

// The closing step of k_move_pairs for a band workgroup: what k_record (or k_apply_fpend) did as a launch of its own at
// the end of a sweep -- the F band of the sweep's last accepted E->I-type update applied to this workgroup's rows, and
// (record != 0) those rows' events written to the trace in the reference's [M][T][3] order.  Role 0 raises token 4 when
// the pending update is decided (Chains::fpend) and token 5 when its rows are in the planes; the workgroup's L1 was
// emptied when the step began and it has loaded nothing of the planes since: plain loads.
// sweep0: the chain's sweep counter as the launch found it (role 0 advances it in this very step).
__device__ __forceinline__ void pair_band_finish(const Dims &d, const Consts &c, const Work &w, const SamplerCfg &s,
                                                 const Chains &ch, int b, int bx, int nband, unsigned token, unsigned sweep0,
                                                 int record) {
    __shared__ Move fp;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                          // (see pair_step)
    const int wave = tid >> 6, lane = tid & 63;
    const unsigned *done = ch.done + (size_t)b * 2 * TAIL_STRIDE;
    const int rpb = (d.M + nband - 1) / nband;
    const int r_lo = bx * rpb, r_hi = min(d.M, r_lo + rpb);
    if (tid == 0) wait_token(done + 4, token, ch.late + ch.late_fatal + b);
    __syncthreads();
    move_copy_l2(&fp, ch.fpend + b, 128);
    if (tid == 0) wait_token(done + 5, token, ch.late + ch.late_fatal + b);
    __syncthreads();
    const unsigned slot = sweep0 - ch.slot0[0];
    for (int m = r_lo + wave; m < r_hi; m += MVW) {
        if (fp.valid == 1) {
            double coef[MMAX];
#pragma unroll
            for (int i = 0; i < MMAX; ++i)
                coef[i] = i < fp.n ? c.Cstar[(size_t)fp.m[i] * d.Kp0 + m] * c.invN[fp.m[i]] * (double)(-fp.dsrc[i]) : 0.0;
            double *Fr = w.F + ((size_t)b * d.Mp + m) * d.Tp;
            for (int t = fp.LO + lane; t <= fp.HI; t += WAVE) {
                double dF = 0.0;
#pragma unroll
                for (int i = 0; i < MMAX; ++i)
                    if (i < fp.n && t > fp.lo[i] && t <= fp.hi[i]) dF += coef[i];
                if (dF != 0.0) Fr[t] += dF;
            }
        }
        if (!record || slot >= (unsigned)s.cap) continue;
        // (k_record's copy: one wave per row, lanes over days, four 64-day pieces of the three planes in flight)
        const size_t o0 = (((size_t)slot * s.B + b) * d.M + m) * d.T * 3;
        int *out = (int *)ch.tr_events + o0;
        unsigned short *out16 = (unsigned short *)ch.tr_events + o0;
        const size_t q0 = ((size_t)b * d.Mp + m) * d.Tp;
        for (int t0 = 0; t0 < d.T; t0 += 4 * WAVE) {
            int k0[4], k1[4], k2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                     // pads exist up to Tp: unconditional loads
                const int t = t0 + j * WAVE + lane;
                const bool in = t < d.Tp;
                k0[j] = in ? w.K[0][q0 + t] : 0;
                k1[j] = in ? w.K[1][q0 + t] : 0;
                k2[j] = in ? w.K[2][q0 + t] : 0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + j * WAVE + lane;
                if (t < d.T) {
                    if (s.ev16) {                                   // uniform
                        if ((unsigned)(k0[j] | k1[j] | k2[j]) > 0xffffu) ch.ev_overflow[0] = 1u;     // reported by the read
                        out16[t * 3 + 0] = (unsigned short)k0[j]; out16[t * 3 + 1] = (unsigned short)k1[j];
                        out16[t * 3 + 2] = (unsigned short)k2[j];
                    } else {
                        out[t * 3 + 0] = k0[j]; out[t * 3 + 1] = k1[j]; out[t * 3 + 2] = k2[j];
                    }
                }
            }
        }
    }
}

// One pair of updates by the workgroup in `slot` of chain b: the body of k_move_pair (one launch per pair) and of one
// step of k_move_pairs (every pair of a sweep in one launch).  nroles: role slots of the grid (the speculative roles in
// the low slots, role 0 in the last), nband: band workgroups per chain that take part in THIS pair (0: none).
template <int NCH, bool SOLO>
__device__ __forceinline__ void pair_step(const Dims &d, const Consts &c, const Work &w, const SamplerCfg &s, const Chains &ch,
                                          MoveSpec se, MoveSpec next, MoveSpec se_next, int have_prev, int have_pre, int pbuf,
                                          int lidx, int dbg, int nband, int nroles, int slot, int b, int fin = 0,
                                          unsigned sweep0 = 0u) {
    extern __shared__ __attribute__((aligned(16))) int dyn_i[];                     // rg [M] | rt [M]
    __shared__ MvShared sm_se, sm_nx;
    __shared__ Move pendA, pendB;
    __shared__ PairNote note;
    __shared__ double pre_down[2], s_down[4], s_dsum[2];
    __shared__ double2 ltab[LDSTAB_N];
    __shared__ int s_sel, s_acc_se, s_conf, s_late, s_late2, s_use_pre, s_trans;
    // (opaque to the compiler: inside k_move_pairs' loop nothing derived from the thread or the chain is to be computed
    // once ahead of the loop and kept in registers across every step)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    asm volatile("" : "+s"(b));
    const int b_stamp = b - d.b0;
    (void)b_stamp;
    QSTAMP(slot, lidx, 0);
    if (slot >= nroles) {                                  // band workgroups (the highest block ids)
        if (nband == 0) return;
        if (fin) {                                         // the closing step of k_move_pairs: they finish the sweep
            pair_band_finish(d, c, w, s, ch, b, slot - nroles, nband, sweep0 * 64u + (unsigned)lidx + 1u, sweep0, fin & 2);
            return;
        }
        const unsigned tok = ch.sweep[b] * 64u + (unsigned)lidx + 1u;
        const bool r1 = next.kind >= 0 && nroles >= 2, r2 = se_next.kind >= 0 && nroles == 3;
        if ((d.M + nband - 1) / nband > 2 * MVW)              // (uniform) more than 16 rows per band workgroup: four per wave
            pair_band_block<SOLO, 4>(d, c, w, s, ch, b, slot - nroles, nband, tok, r1, r2, pbuf ^ 1, slot, lidx);
        else
            pair_band_block<SOLO, 2>(d, c, w, s, ch, b, slot - nroles, nband, tok, r1, r2, pbuf ^ 1, slot, lidx);
        return;
    }
    const int role = slot == nroles - 1 ? 0 : slot + 1;
    const int M = d.M, T = d.T;
    const bool do_se = role == 0 && se.kind >= 0, do_nx = next.kind >= 0, do_pre = se_next.kind >= 0 && nroles == 3;
    if (role == 1 && !do_nx) return;
    if (role == 2 && !do_pre) return;
#ifdef SEIR_STAMPS
    double *stamp_hs = ch.hs + (size_t)b * NHS;
    const bool stamp_on = do_se && b == 0 && se.slot == (SEIR_STAMP_SLOT & 2) && se.scan == 0;
#define PSTAMP(i) do { if (threadIdx.x == 0 && stamp_on) ((unsigned long long *)(stamp_hs + 16))[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#ifndef SEIR_STAMP_ROLE
#define SEIR_STAMP_ROLE 1
#endif
    // a speculative role's phases (slots 8..11): role SEIR_STAMP_ROLE of chain 0 in the launch whose S->E slot matches
    const bool rstamp_on = role == SEIR_STAMP_ROLE && b == 0 && se.slot == (SEIR_STAMP_SLOT & 2) && se.scan == 0;
#ifdef SEIR_STAMP_PROPOSE      // phases inside the speculative role's mv_propose (slots 4..9); its own phases move to 12..15
#define RSTAMP(i) do { if (threadIdx.x == 0 && rstamp_on) ((unsigned long long *)(stamp_hs + 16))[(i) + 4] = __builtin_amdgcn_s_memrealtime(); } while (0)
#undef PSTAMP
#define PSTAMP(i) do {} while (0)
#else
#define RSTAMP(i) do { if (threadIdx.x == 0 && rstamp_on) ((unsigned long long *)(stamp_hs + 16))[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#else
#define PSTAMP(i) do {} while (0)
#define RSTAMP(i) do {} while (0)
#endif
    PSTAMP(0);
    RSTAMP(8);
    // ---- entry: everything that does not depend on a decision made in this launch
    const bool pre_ok = M <= PRE_RT * MVB;
    const MoveSpec &mine = role == 2 ? se_next : next;          // the proposal a speculative role draws
    int pre_se[PRE_RT], pre_nx[PRE_RT];
    mv_prefetch_rows(d, w, s, b, se, do_se && pre_ok, pre_se);
    mv_prefetch_rows(d, w, s, b, mine, (role == 2 || do_nx) && pre_ok, pre_nx);
    const unsigned token = ch.sweep[b] * 64u + (unsigned)lidx + 1u;       // unique per (sweep, launch): lidx < 63
    // ... all of it issued before the first wait: the pending descriptors, k_move_delta's partial sums, the
    // own-rows parts, the pre-drawn proposal and its note -- one round trip for the whole entry, into registers (a thread
    // holds at most one word of the descriptors and one of the doubles): the uniforms of this step's proposals (Philox and
    // a logarithm by a few lanes, ~1 us) are drawn while the loads are in flight, and only then do the values go to LDS
    double dth0 = 0.0, dcn0 = 0.0;
    int ld_i = 0;
    double ld_d = 0.0;
    if (have_prev) {
        if (tid >= 64 && tid < 64 + MOVE_DW) ld_i = reinterpret_cast<const int *>(ch.mv + (size_t)pbuf * s.B + b)[tid - 64];
        if (tid >= 128 && tid < 128 + MOVE_DW) ld_i = reinterpret_cast<const int *>(ch.mvfix + (size_t)pbuf * s.B + b)[tid - 128];
        if (tid == 192) ld_i = ch.mvsel[(size_t)pbuf * s.B + b];
        if (tid >= 196 && tid < 200) ld_d = ch.Down[(((size_t)pbuf * 2 + ((tid - 196) >> 1)) * s.B + b) * 2 + ((tid - 196) & 1)];
        if (tid < s.nrb_d) {
            dth0 = ch.Dpart[((size_t)b * s.nrb_d + tid) * 2];
            dcn0 = ch.Dpart[((size_t)b * s.nrb_d + tid) * 2 + 1];
        }
        for (int i = tid + MVB; i < s.nrb_d; i += MVB) {
            dth0 += ch.Dpart[((size_t)b * s.nrb_d + i) * 2];
            dcn0 += ch.Dpart[((size_t)b * s.nrb_d + i) * 2 + 1];
        }
    }
    const bool pre_avail = do_se && have_pre;
    if (pre_avail) {
        if (tid >= 256 && tid < 256 + MOVE_DW) ld_i = reinterpret_cast<const int *>(ch.mvs + (size_t)pbuf * s.B + b)[tid - 256];
        if (tid >= 320 && tid < 320 + 2 + MMAX) ld_i = reinterpret_cast<const int *>(ch.prev + (size_t)pbuf * s.B + b)[tid - 320];
        if (tid == 328 || tid == 329) ld_d = ch.DownS[((size_t)pbuf * s.B + b) * 2 + (tid - 328)];
    }
    static_assert(MOVE_DW <= 64 && 2 + MMAX <= 8, "the entry loads of pair_step: disjoint thread ranges");
    double hs_th = 0.0, hs_cn = 0.0, psi = 0.0;
    unsigned tr_slot = 0xffffffffu;
    double2 ld_t = make_double2(0.0, 0.0);
    if (tid < LDSTAB_N) ld_t = c.logtab[tid];
    if (role == 0 && tid == 0) {
        const double *hs = ch.hs + (size_t)b * NHS;
        hs_th = hs[HS_LP_THETA]; hs_cn = hs[HS_LP_CONST];
        tr_slot = ch.sweep[b] - ch.slot0[0];
    }
    psi = w.scal[(size_t)b * NSCAL + SC_PSI];
    QSTAMP(slot, lidx, 6);
    if (do_se && !pre_avail) mv_draw(s, ch, b, se, sm_se, T);
    if (role != 0 || do_nx) mv_draw(s, ch, b, mine, sm_nx, T);
    QSTAMP(slot, lidx, 14);
    if (have_prev) {
        if (tid >= 64 && tid < 64 + MOVE_DW) reinterpret_cast<int *>(&pendA)[tid - 64] = ld_i;
        if (tid >= 128 && tid < 128 + MOVE_DW) reinterpret_cast<int *>(&pendB)[tid - 128] = ld_i;
        if (tid == 192) s_sel = ld_i;
        if (tid >= 196 && tid < 200) s_down[tid - 196] = ld_d;
    }
    if (pre_avail) {
        if (tid >= 256 && tid < 256 + MOVE_DW) reinterpret_cast<int *>(&sm_se.mv)[tid - 256] = ld_i;
        if (tid >= 320 && tid < 320 + 2 + MMAX) reinterpret_cast<int *>(&note)[tid - 320] = ld_i;
        if (tid == 328 || tid == 329) pre_down[tid - 328] = ld_d;
    }
    if (tid < LDSTAB_N) ltab[tid] = ld_t;
    if (have_prev && s.nrb_d <= WAVE && tid < WAVE) {
        const double a = wave_sum(dth0), a2 = wave_sum(dcn0);
        if (tid == 0) { s_dsum[0] = 0.0 + a; s_dsum[1] = 0.0 + a2; }
    }
    if (role != 0) {
        // the totals are in registers: tell role 0 it may start writing
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QSTAMP(slot, lidx, 12);
        __syncthreads();
        QSTAMP(slot, lidx, 13);
        // test hooks (seir_sampler_desc::debug_pair): 1 = role 1 posts its token late, 2 = never; 4 / 8: role 2
        const int late_bit = role == 1 ? 1 : 4, absent_bit = role == 1 ? 2 : 8;
        if (dbg & late_bit)
            for (int i = 0; i < 100; ++i) __builtin_amdgcn_s_sleep(127);   // ~0.35 ms: well inside role 0's bounded wait
        if (tid == 0 && !(dbg & absent_bit))
            __hip_atomic_store((role == 1 ? ch.hand : ch.hand2) + b, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (loads done: above)
    }
    MvLds L{};
    int *rtl = dyn_i + M;
    L.rt = rtl;
    L.rg = dyn_i;
    // ---- (1) the pending E->I-type proposal: every role needs the decision, role 0 acts on it
    bool pend_acc = false;
    const Move *pendp = nullptr;
    if (have_prev) {
        __syncthreads();                                   // the descriptors fetched at entry are in LDS
        QSTAMP(slot, lidx, 15);
        const Move &pend = s_sel ? pendB : pendA;          // speculative one, or re-drawn after a row conflict
        pendp = &pend;
        if (s.nrb_d <= WAVE) {
            // the partial sums are all in wave 0: its sums went to LDS ahead of the barrier above (the value mv_sum2 gives:
            // the other waves' parts are zeros)
            dth0 = s_dsum[0]; dcn0 = s_dsum[1];
        } else {
            mv_sum2(dth0, dcn0, sm_nx.dred);
        }
        dth0 += s_down[s_sel ? 2 : 0];                     // the updated rows' part
        dcn0 += s_down[s_sel ? 3 : 1];
        const double ratio = dth0 + dcn0 + pend.logq;
        pend_acc = pend.valid && pend.logu < ratio;        // NaN -> reject
    }
    if (role != 0) {
        // ------------------------------------------------------------ speculative proposal (role 1: E->I-type
        // of this pair, role 2: S->E-type of the next pair)
        const Move *fix = (pend_acc && pendp->tgt == mine.tgt) ? pendp : nullptr;
        RSTAMP(9);
#if defined(SEIR_STAMPS) && defined(SEIR_STAMP_PROPOSE)
        L.stamp_hs = stamp_hs; L.stamp_on = rstamp_on;
#endif
        QSTAMP(slot, lidx, 1);
        mv_rows_to_lds(d, w, s, b, mine, pre_ok, pre_nx, fix, L, rtl);
        mv_propose<NCH>(d, w, s, ch, b, mine, sm_nx, L, ltab);
        QSTAMP(slot, lidx, 2);
        RSTAMP(10);
        Move *out = (role == 1 ? ch.mv : ch.mvs) + (size_t)(pbuf ^ 1) * s.B + b;
        move_copy(out, &sm_nx.mv, MVB - WAVE);             // by the last wave: nobody's loads queue behind the store
        // band workgroups start from this descriptor while the log-ratio over the updated rows is still being evaluated: as
        // hand-off words, which need neither the acknowledgement of the stores nor a token
        if (nband > 0 && role == 1 && tid >= MVB - WAVE)
            move_store_ll(ch.llmv + ((size_t)(pbuf ^ 1) * s.B + b) * 32, &sm_nx.mv, tid - (MVB - WAVE), token);
        // ... and its log-ratio over the rows it updates (for role 1, k_move_delta then does the band only); the F
        // band of an accepted pending update is not in F yet and is added on the fly
        const Move *fpp = (pend_acc && pendp->any_dI) ? pendp : nullptr;
        double *od = role == 1 ? ch.Down + (((size_t)(pbuf ^ 1) * 2 + 0) * s.B + b) * 2
                               : ch.DownS + ((size_t)(pbuf ^ 1) * s.B + b) * 2;
        mv_own_rows_to_down(d, c, w, b, sm_nx.mv, psi, ltab, fpp, sm_nx.dred, od);
        QSTAMP(slot, lidx, 3);
        RSTAMP(11);
        if (nband > 0) {                                   // band workgroups: this role reads F no more, its output is in L2
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(ch.done + (size_t)b * 2 * TAIL_STRIDE + role, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (!have_prev && tid == 64) {                         // first launch of a sweep: nothing pending
        ch.fpend[b].valid = 0;
        if (nband > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(ch.done + (size_t)b * 2 * TAIL_STRIDE + 4, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // before the first store a speculative role could mistake for the state at entry: has it fetched its totals?
    // (bounded wait; normally the roles are long past that point when the accept test above is done)
    bool late = false, late2 = false;
    auto wait_roles = [&]() {
        if (tid == 0) {
            int spins = 0;
            if (do_nx) {
                while (__hip_atomic_load(ch.hand + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != token && spins < 4000) {
                    __builtin_amdgcn_s_sleep(2);
                    ++spins;
                }
            }
            s_late = spins >= 4000 ? 1 : 0;
            int spins2 = 0;
            if (do_pre) {
                while (__hip_atomic_load(ch.hand2 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != token && spins2 < 4000) {
                    __builtin_amdgcn_s_sleep(2);
                    ++spins2;
                }
            }
            s_late2 = spins2 >= 4000 ? 1 : 0;
            if (spins >= 4000 || spins2 >= 4000) ch.late[b] += 1;   // visible through seir_sampler_pair_timeouts
            // relaxed polls, one acquire once the tokens are seen: this workgroup's stores below are ordered after it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        lds_barrier();
        late = s_late != 0;
        late2 = s_late2 != 0;
    };
    bool waited = !do_nx && !do_pre;                       // no speculative role in this launch: nothing to wait for
    if (have_prev) {
        const Move &mv = *pendp;
        if (pend_acc && !waited) { wait_roles(); waited = true; }
        // The F band of an accepted E->I update is NOT written here: k_move_delta (or k_record / k_apply_fpend
        // at the end of the sweep) does it with the whole chip; the S->E update below adds the pending band to
        // the F values it reads.
        if (pend_acc && mv.any_dI) move_copy(ch.fpend + b, &mv, MVB - WAVE);     // mv.valid == 1: it was accepted
        else if (tid == MVB - WAVE) ch.fpend[b].valid = 0;
        if (nband > 0 && tid >= MVB - WAVE) {
            // band workgroups prefetch what the F band needs (token 4) long before the proposal they evaluate is certified
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == MVB - WAVE)
                __hip_atomic_store(ch.done + (size_t)b * 2 * TAIL_STRIDE + 4, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pend_acc) mv_apply_rows(d, w, s, b, mv);
        if (tid == 0) {
            if (pend_acc) {
                double *hs = ch.hs + (size_t)b * NHS;
                hs_th += dth0; hs_cn += dcn0;
                hs[HS_LP_THETA] = hs_th; hs[HS_LP_CONST] = hs_cn;
            }
            mv_trace(s, ch, b, mv, pend_acc ? 1 : 0, tr_slot, hs_th + hs_cn);
        }
        lds_barrier();
    }
    PSTAMP(1);
    QSTAMP(slot, lidx, 1);
    if (fin && nband > 0) {
        // closing step of k_move_pairs: the band workgroups finish the sweep (pair_band_finish) once the planes are final
        __syncthreads();                                   // mv_apply_rows ends drained; the trace stores need not be
        if (tid == 0)
            __hip_atomic_store(ch.done + (size_t)b * 2 * TAIL_STRIDE + 5, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- (2) the whole S->E-type update
    bool se_acc = false;
    if (do_se) {
        const Move *fpp = (pend_acc && pendp->any_dI) ? pendp : nullptr;
        // the pre-drawn proposal stands unless the previous launch ruled it out or one of its rows was changed
        // by the E->I-type update accepted there (note.rows) or by the one just finalized
        if (tid == 0) {
            int use = pre_avail && note.ok;
            if (use) {
                const Move &mv = sm_se.mv;
                for (int i = 0; i < mv.n; ++i) {
                    for (int j = 0; j < note.n; ++j) use &= mv.m[i] != note.rows[j];
                    if (pend_acc)
                        for (int j = 0; j < pendp->n; ++j) use &= mv.m[i] != pendp->m[j];
                }
                // an out-of-range sub-move leaves no row in mv.m: its rows cannot be checked -- it is rejected
                // whatever the state is, so the proposal stands (mv.valid == 0)
            }
            s_use_pre = use;
        }
        lds_barrier();
        const bool use_pre = s_use_pre != 0;
        double dth = 0.0, dcn = 0.0;
        if (!use_pre) {
            if (pre_avail) {                                // not drawn at entry: draw the uniforms and the header now
                lds_barrier();
                mv_draw(s, ch, b, se, sm_se, T);
            }
            // the pending update was of the other plane (tgt 1): nothing to correct in plane 0's totals
            mv_rows_to_lds(d, w, s, b, se, pre_ok, pre_se, nullptr, L, rtl);
            PSTAMP(2);
            mv_propose<NCH>(d, w, s, ch, b, se, sm_se, L, ltab);
            PSTAMP(3);
        }
        const Move &mv = sm_se.mv;
        if (use_pre && fpp == nullptr) {
            dth = pre_down[0]; dcn = pre_down[1];           // F under the proposal's rows is what role 2 saw
        } else {
            if (mv.valid && mv.n > 0) own_rows_delta<MVB>(d, c, w, b, mv, psi, 0, M, ltab, dth, dcn, fpp);
            PSTAMP(4);
            mv_sum2(dth, dcn, sm_se.dred);
        }
        if (tid == 0) {
            const double ratio = dth + dcn + mv.logq;
            s_acc_se = (mv.valid && mv.logu < ratio) ? 1 : 0;    // NaN -> reject
        }
        lds_barrier();
        se_acc = s_acc_se != 0;
        if (se_acc && !waited) { wait_roles(); waited = true; }
        if (tid == 0) s_trans = 0;
        if (se_acc) mv_apply_rows(d, w, s, b, mv, &s_trans);       // (its first barrier orders the store above)
        if (tid == 0) {
            if (se_acc) {
                double *hs = ch.hs + (size_t)b * NHS;
                hs_th += dth; hs_cn += dcn;
                hs[HS_LP_THETA] = hs_th; hs[HS_LP_CONST] = hs_cn;
            }
            mv_trace(s, ch, b, mv, s_acc_se, tr_slot, hs_th + hs_cn);
        }
        __syncthreads();             // a re-drawn proposal in (3) must see the state written above
        // band workgroups: the planes are final (token 5); which descriptor stands they learn from the last token
        if (nband > 0 && tid == 0)
            __hip_atomic_store(ch.done + (size_t)b * 2 * TAIL_STRIDE + 5, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PSTAMP(5);
        QSTAMP(slot, lidx, 2);
    }
    // ---- (3) certify role 1's proposal; closing launch: advance the sweep counter
    if (next.kind == -2 && tid == 0) ch.sweep[b] += 1;
    if (do_nx) {
        // nothing was accepted in (1) and (2): the state role 1 read is the final one, no row can conflict
        const bool changed = se_acc || pend_acc;             // uniform
        if (changed) {
            // plane 1 totals were prefetched before (1): correct them if the pending (plane 1) update was accepted
            const Move *fix = (pend_acc && pendp->tgt == next.tgt) ? pendp : nullptr;
            mv_rows_to_lds(d, w, s, b, next, pre_ok, pre_nx, fix, L, rtl);
            PSTAMP(6);
            mv_propose<NCH>(d, w, s, ch, b, next, sm_nx, L, ltab, /*rows_only=*/true);
        }
        if (tid == 0) {
            int conf = late ? 1 : 0;                         // role 1 was not there in time: do not trust it
            if (changed)
                for (int j = 0; j < sm_nx.nsel; ++j) {
                    const int row = sm_nx.sel[j];
                    if (se_acc)
                        for (int i = 0; i < sm_se.mv.n; ++i) conf |= sm_se.mv.m[i] == row;
                    if (pend_acc)
                        for (int i = 0; i < pendp->n; ++i) conf |= pendp->m[i] == row;
                }
            s_conf = conf;
            ch.mvsel[(size_t)(pbuf ^ 1) * s.B + b] = conf;   // double-buffered like Chains::mv: a late role 1 still reads the old one
        }
        lds_barrier();
        if (s_conf) {                                        // rare: draw it again from the final state
            mv_propose<NCH>(d, w, s, ch, b, next, sm_nx, L, ltab);
            move_copy(ch.mvfix + (size_t)(pbuf ^ 1) * s.B + b, &sm_nx.mv, MVB - WAVE);
            const Move *fpp = (pend_acc && pendp->any_dI) ? pendp : nullptr;
            mv_own_rows_to_down(d, c, w, b, sm_nx.mv, psi, ltab, fpp, sm_nx.dred,
                                ch.Down + (((size_t)(pbuf ^ 1) * 2 + 1) * s.B + b) * 2);
        }
        PSTAMP(7);
        QSTAMP(slot, lidx, 3);
    }
    // ---- (4) the note for the next launch about the proposal role 2 is pre-drawing
    if (do_pre && tid == 0) {
        // (field by field: a struct assembled in a local and assigned is copied through scratch memory)
        int *nt = reinterpret_cast<int *>(ch.prev + (size_t)(pbuf ^ 1) * s.B + b);
        // if the wait never happened nothing was written in this launch, and a late role 2 read the final state.
        // An S->E-type update accepted in this launch does not by itself rule the pre-drawn proposal out: it was drawn
        // from the totals at entry, and what a proposal takes from the totals is WHICH rows hold events (and the
        // total of the row it picks) -- unless one of the updated rows went from none to some or back in the totals
        // the pre-drawn kind reads (row totals for an event-time move, occult-range totals for an occult), it picks
        // the same rows, and unless it picked an updated row (checked by the next launch against the rows noted
        // here, like the rows of an accepted E->I-type update) it finds them as it saw them.
        const int n_pend = pend_acc ? pendp->n : 0, n_se = se_acc ? sm_se.mv.n : 0;
        const bool se_keeps = !se_acc || (((s_trans & (se_next.kind == 0 ? 1 : 2)) == 0) && n_pend + n_se <= MMAX);
        nt[0] = (se_keeps && !(waited && late2)) ? 1 : 0;                       // PairNote::ok
        nt[1] = n_pend + (se_keeps ? n_se : 0);                                 // PairNote::n
#pragma unroll
        for (int j = 0; j < MMAX; ++j) {
            int row = -1;
            if (j < n_pend) row = pendp->m[j];
            else if (se_keeps && j - n_pend < n_se) row = sm_se.mv.m[j - n_pend];
            nt[2 + j] = row;                                                    // PairNote::rows
        }
    }
    if (nband > 0) {                                       // band workgroups may go: descriptors, mvsel and fpend are in L2
        __syncthreads();
        if (tid == 0)                                      // (which descriptor stands -- Chains::mvsel -- in the token's top bit)
            __hip_atomic_store(ch.done + (size_t)b * 2 * TAIL_STRIDE + 0, token | ((do_nx && s_conf) ? 0x80000000u : 0u), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NCH>
__global__ __launch_bounds__(MVB) void k_move_pair(Dims d, Consts c, Work w, SamplerCfg s, Chains ch, MoveSpec se,
                                                   MoveSpec next, MoveSpec se_next, int have_prev, int have_pre,
                                                   int pbuf, int nbk, int lidx, int dbg, int nband) {
    debug_skew(d);
    // block id = slot * nbk + chain with the speculative roles in the low slots: they are dispatched first,
    // so an authoritative workgroup never holds a CU waiting for a partner that has not been placed yet,
    // whatever the number of chains; the closing launch has role 0 only
    const int nroles = (int)gridDim.x / nbk - nband, slot = (int)blockIdx.x / nbk;
    const int b = d.b0 + (int)blockIdx.x - slot * nbk;
    if (d.nlive > 0 && (int)blockIdx.x - slot * nbk >= d.nlive) return;             // a chain of the layout that does not exist
    pair_step<NCH, false>(d, c, w, s, ch, se, next, se_next, have_prev, have_pre, pbuf, lidx, dbg, nband, nroles, slot, b);
    const int b_stamp = b - d.b0;
    (void)b_stamp;
    QSTAMP(slot, lidx, 8);
}

// ---------------------------------------------------------------------------------------------
// k_move_pairs: EVERY pair of a sweep, and the closing step, in ONE launch -- the grid of k_move_pair with band
// workgroups (3 roles + nband band workgroups per chain, every workgroup resident, a chain's workgroups on one XCD),
// each workgroup walking through the steps k_move_pair is launched for.  What a launch boundary gave is restated per
// chain: a workgroup that has finished step i drains its stores (they are acknowledged by the XCD's L2), counts in on
// the chain's counter (Chains::pbar) and waits until it shows them all; then it drops its CU's L1 and scalar
// cache, so that the plain loads of step i + 1 see what any workgroup of the chain wrote in step i, exactly as the
// loads of a new launch would.  Inside a step nothing changes: the hand-offs of k_move_pair (tokens unique per sweep
// and step, reads past the L1 of everything written in the same step).  What it saves is what a launch costs a
// latency chain like this one: the dispatch ramp, the kernel-argument and first-touch misses, a cold instruction
// cache for a path of a few thousand instructions executed once, and the drain at the end.
// Chains are independent: there is no barrier across chains.
// ---------------------------------------------------------------------------------------------
constexpr int PBAR_STRIDE = 64;                               // 32-bit words per chain: the counter in a line of its own (and a spare line:
                                                              // a flag there, raised by the last arrival and polled instead of the
                                                              // counter, changed nothing -- 0.3344 against 0.3342 ms per sweep)
// `target`: what the chain's counter shows once every workgroup of the chain has finished the step -- the counter runs on
// over the launches (the host knows how many steps have been counted: seir_sampler::pbar_count), so the last one in
// has nothing to reset or to raise, and the others poll the counter itself
__device__ __forceinline__ void pair_chain_barrier(const Chains &ch, int b, unsigned target, int b_stamp = 0, int st_slot = 0,
                                                   int st_step = 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores of the step are in L2
    __syncthreads();
    QSTAMP(st_slot, st_step, 10);
    if (threadIdx.x == 0) {
        unsigned *cnt = ch.pbar + (size_t)b * PBAR_STRIDE;
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Drop this CU's L1, so that plain loads of the next step see what other workgroups wrote in this one -- at once,
        // not when the count is complete: the launch holds ONE workgroup per CU (its LDS request, k_move_pairs_lds_bytes),
        // and from here to the end of the wait this one loads nothing but the counter, past the L1, so the cache stays empty.
        // ONE wave per workgroup, and not all at the same moment: an agent-scope invalidate (buffer_inv sc1) is also a
        // request to the XCD's L2, where the requests of a whole grid queue up -- with every wave issuing one after the wait
        // (8 x 27 x 8 of them per step) the next step's first loads came back 10 us late, with one per workgroup after the
        // wait 1.3 us (tools/dev/pair_timeline.py); the narrower `buffer_inv sc0` leaves the L1 as it is
        // (tools/probes/l1inv_probe.hip).  The instruction completes like a load: waited for before the barrier below
        // lets the other waves go.  (The scalar cache needs nothing: what is read through it is kernel arguments.)
        // (Issued AHEAD of the arrival instead, so that the last workgroup in has the two round trips side by side: slower,
        // 0.3063 against 0.3029 ms per sweep -- the atomic queues behind the invalidate.)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (old + 1u != target) {
            int spins = 0;
            unsigned *late = ch.late + ch.late_fatal + b;
#ifndef PBAR_SLEEP
#define PBAR_SLEEP 1
#endif
            while ((int)(ld_l2(cnt) - target) < 0) {
                __builtin_amdgcn_s_sleep(PBAR_SLEEP);
                ++spins;
                if ((spins & 255) == 0 && ld_l2(late) != 0u) break;                          // (see wait_token)
                if (spins > (1 << 22)) { __hip_atomic_fetch_add(late, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // counted, no hang
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    QSTAMP(st_slot, st_step, 11);
    __syncthreads();
}

template <int NCH>
__global__ __launch_bounds__(MVB) void k_move_pairs(Dims d, Consts c, Work w, SamplerCfg s, Chains ch, int npairs, int pre_on,
                                                    int nbk, int dbg, int nband, unsigned pbase, int fin) {
    debug_skew(d);
    const int nroles = (int)gridDim.x / nbk - nband, slot = (int)blockIdx.x / nbk;   // 3 roles
    const int b = d.b0 + (int)blockIdx.x - slot * nbk;
    if (d.nlive > 0 && (int)blockIdx.x - slot * nbk >= d.nlive) return;             // a chain of the layout that does not exist
    const MoveSpec none{-1, 0, 0, 0};
    // fin: in the closing step the band workgroups apply the F band of the last accepted E->I-type update (1) and write
    // the sweep's events to the trace (3): k_apply_fpend's / k_record's work, without their launch
    const unsigned sweep0 = ch.sweep[b];
    for (int pair = 0; pair <= npairs; ++pair) {
        // as enqueue_sweep launches k_move_pair: scan = pair / 2, the first half of a scan is the event-time moves, the second
        // the occults; step npairs is the closing one: role 0 finalizes the last E->I-type proposal and advances the sweep counter
        const bool closing = pair == npairs;
        const int scan = pair >> 1, half = pair & 1, nh = half ^ 1, nscan = scan + half;
        const bool pre = pre_on && pair + 1 < npairs;
        const MoveSpec se = closing ? none : MoveSpec{half, 0, 2 * half, scan};
        const MoveSpec nx = closing ? MoveSpec{-2, 0, 0, 0} : MoveSpec{half, 1, 2 * half + 1, scan};
        const MoveSpec se_next = pre ? MoveSpec{nh, 0, 2 * nh, nscan} : none;
        pair_step<NCH, true>(d, c, w, s, ch, se, nx, se_next, pair > 0 ? 1 : 0, (pair > 0 && pre_on && !closing) ? 1 : 0, pair & 1,
                       closing ? 62 : pair, closing ? 0 : dbg, (closing && !fin) ? 0 : nband, nroles, slot, b, closing ? fin : 0, sweep0);
        if (closing) break;
        const int b_stamp = b - d.b0;
        (void)b_stamp;
        QSTAMP(slot, pair, 8);
        pair_chain_barrier(ch, b, pbase + (unsigned)((pair + 1) * (nroles + nband)), b_stamp, slot, pair);
        QSTAMP(slot, pair, 9);
    }
}

inline size_t k_move_pa2_lds_bytes(const Dims &d) {
    return sizeof(int) * ((size_t)2 * d.M);
}
// k_move_pairs asks for more than half a CU's LDS (160 KB): one workgroup per CU, which its step barrier relies on
inline size_t k_move_pairs_lds_bytes(const Dims &d) {
    const size_t need = k_move_pa2_lds_bytes(d), half = 82 * 1024;
    return need > half ? need : half;
}

}  // namespace seir
