// Device-resident Metropolis-within-Gibbs sweep for B chains (one posterior draw
// per chain per sweep), re-stating on MI355X what the reference builds from
//   GibbsKernel[ HMC(u | events),  MultiScan(num_event_time_updates,
//       Gibbs[ MH(EventTimesUpdate S->E), MH(EventTimesUpdate E->I),
//              MH(OccultUpdate S->E),     MH(OccultUpdate E->I) ]) ]
// (covid19uk/inference/inference.py:86-101,219-228;
//  covid19uk/inference/mcmc_kernel_factory.py:14-168).
//
// Chain state lives in HBM: int32 event planes K[3] and state planes St[3]
// ([B][Mp][Tp], T contiguous), the cached mobility contraction F = Cstar.I/N,
// the per-day I->R statistics, and the HMC position/momentum vectors.  The
// reference re-evaluates the full target for every MH proposal
// (mcmc_kernel_factory.py:72-83,99-110); here a proposal's log-ratio is
// evaluated over the cells it changes (k_move_delta) and F is updated by a
// rank-1 column update on acceptance.  tests/test_sampler_gpu.py checks the
// running log-prob against a full re-evaluation.
//
// The proposal distributions re-state gemlib's UncalibratedEventTimesUpdate /
// UncalibratedOccultUpdate as documented in DESIGN.md ("MCMC kernels"); the
// same definitions are implemented by the CPU oracle oracle/mcmc_oracle.py with
// the same Philox stream, so traces are comparable draw by draw.
#pragma once
#include "logprob_kernels.h"
#include "philox.h"

namespace seir {

constexpr int MMAX = 4;           // upper bound on config["m"]
constexpr int NHS = 32;           // per-chain HMC scalar block
enum {
    HS_EPS = 0,       // step size used by the current/next HMC step
    HS_LP_THETA,      // prior + jac + S->E + I->R terms at (q, events)
    HS_LP_CONST,      // binomial coefficients + E->I term (parameter free)
    HS_LP0,           // HS_LP_THETA at the start of the trajectory
    HS_K0,            // kinetic energy at the start
    HS_DA_ERR, HS_DA_STEP, HS_DA_LOGAVG, HS_DA_MU,
    HS_RV_N,          // running-variance sample count
    HS_ACC,           // last HMC accept flag
    HS_LOGACC         // last log accept ratio
};

struct SamplerCfg {
    int B, dmax, nmax, mmax, occult_nmax, n_scans, tr_lo, tr_hi, L;
    uint32_t k0, k1;
    int chain0;
    int adapt_step, adapt_mass, n_adapt;
    double target_accept;
    int cap;          // trace slots
    int nrb_d;        // row blocks of k_move_delta / k_move_pa2
    int ev16;         // 1: samples/seir recorded as uint16 (half the burst buffer and half the bytes over PCIe)
    int disable_mask; // bit 0 HMC, bits 1..4 the four event sub-kernels: proposal drawn, always rejected
};

struct Move {
    int valid, n, tgt, kind;          // kind 0 = event-time move, 1 = occult
    int m[MMAX], a[MMAX], b[MMAX];    // row; day the events leave / occult day; arrival day (-1 none)
    int dka[MMAX], dkb[MMAX];         // change of K[tgt] at a, b
    int lo[MMAX], hi[MMAX];           // state days (lo, hi] change
    int dsrc[MMAX];                   // change of the source compartment there; dest gets -dsrc
    int LO, HI;                       // hull of touched days [LO, HI]
    int any_dI;
    int slot;                         // which of the 4 sub-kernels (trace column)
    int tm[MMAX], tt[MMAX], tdt[MMAX], tx[MMAX];
    double logq, logu;
};

// A descriptor is 60 dwords: copied by one dword per lane of the wave that starts at thread t0 -- one
// coalesced wave instruction instead of 60 dependent ones from a single lane (whose later loads would
// also queue behind them: vmcnt counts in issue order)
constexpr int MOVE_DW = (int)(sizeof(Move) / 4);
static_assert(sizeof(Move) % 4 == 0 && MOVE_DW <= 64, "Move must fit one dword per lane of a wave");
__device__ __forceinline__ void move_copy(Move *dst, const Move *src, int t0) {
    const int i = (int)threadIdx.x - t0;
    if (i >= 0 && i < MOVE_DW) reinterpret_cast<int *>(dst)[i] = reinterpret_cast<const int *>(src)[i];
}

constexpr int TAIL_STRIDE = 16;       // 64-bit words between two chains' ticket counters
constexpr int ROLE_SLOTS = 64;        // chunk roles of a chain that leave parts of the trajectory's ends (k0part, finpart): T-chunks +
                                      // M-chunks, one lane each when a role adds them up (SYN-2048 has 12 + 32)
constexpr int NMVTR = 2 + 4 * MMAX;   // is_accepted, target_log_prob, m[], t[], delta_t[], x_star[]

struct PairNote;
struct Chains {
    double *q, *p, *q0, *grad, *var, *rv_mean, *rv_m2;   // [B][Pp]
    double *hs;                                          // [B][NHS]
    Move *mv;                                            // [2][B] double-buffered proposal descriptors
    Move *fpend;                                         // [B] accepted E->I-type update whose F band is still to be applied (valid = 1)
    Move *mvfix;                                         // [2][B] E->I-type proposal re-drawn after a row conflict (k_move_pair)
    unsigned *hand;                                      // [B] k_move_pair: token of the launch whose role 1 has its totals
    unsigned *late;                                      // [2][B] time-outs of in-launch waits.  [0][b]: k_move_pair launches whose role 0 gave up
                                                         // waiting for a speculative role -- benign, it draws the proposal itself and the
                                                         // traces are the same; [1][b] (offset late_fatal = B): waits that cannot be recovered
                                                         // from (band tokens, k_se_chunk's tile flag): the workgroup went on without its data
    int late_fatal;
                                                         //     (+ k_se_chunk: chunk roles that gave up waiting for the chain's tiles)
    double *finpart;                                     // [B][ROLE_SLOTS][4] k_leap with the trajectory's end folded in: per role, its parts of the
                                                         //     end point's kinetic energy and log-probability (and, by T-chunk 0, the
                                                         //     running-variance count), for the accept test every role then makes
    unsigned *pbar;                                      // [B][PBAR_STRIDE] k_move_pairs: arrivals of the chain's workgroups at the end of a
                                                         //     step, over all launches (a counter per 128-byte line)
    unsigned *done;                                      // [B][2 TAIL_STRIDE] k_move_pair with band workgroups: token of the launch whose
                                                         //     role r has finished, at [b][r] (a chain's three tokens in its own line)
    unsigned long long *tail;                            // [B][TAIL_STRIDE] k_se_chunk: tiles of the chain that have arrived, over
                                                         //     all launches -- one counter per 128-byte line: eight chains' counters in
                                                         //     one line made every ticket a cross-XCD transaction (+14 us per launch);
                                                         //     then [B][TAIL_FLAG_STRIDE]: the count at which the chain's last tile raised
                                                         //     the flag the roles poll
    unsigned long long *leap;                            // [B][LEAP_CH] k_leap's counters and flags (see there), a chain's in its own 8 KB
    double *k0part;                                      // [B][ROLE_SLOTS] k_leap with the trajectory's first step folded in: the roles' parts of
                                                         //         the start point's kinetic energy
    double *irl0;                                        // [B] ... and the I->R term of its log-probability
    unsigned long long *leap_st;                         // [B][16][8] developer timeline of k_leap (LEAP_STAMPS builds only)
    // k_leap's hand-offs as self-validating words (below: "Hand-off words"): what a tile leaves for the roles, per step parity
    uint4 *llK;                                          // [B][2][Mp/16][Tp]   column sums (Work::Kpart)
    uint4 *llR;                                          // [B][2][ntc][Mp]     row sums (Work::Rpart)
    uint4 *llP;                                          // [B][2][ntc Mp/16]   psi parts (Work::Ppart)
    uint4 *llTS;                                         // [B][2][ntc Mp/16][4] tile scalars (Work::TS)
    uint4 *llT;                                          // [B][Tp + 2 Mp + 8]  ... and what the roles leave for the tiles: exp(a_t) | exp(b_m)/N_m |
                                                         //     spatial effects | psi (Work::ea, eb, sp, scal[SC_PSI])
    uint4 *llmv;                                         // [2][B][32] k_move_pair(s) with band workgroups: role 1's proposal descriptor (Chains::mv)
                                                         //     as hand-off words, two dwords apiece, numbered by the launch's token
    unsigned *hand2;                                     // [B] the same token for role 2 (pre-drawn S->E-type proposal)
    Move *mvs;                                           // [2][B] S->E-type proposal pre-drawn for the next launch (k_move_pair, role 2)
    double *DownS;                                       // [2][B][2] its own-rows log-ratio {theta, const}
    struct PairNote *prev;                               // [2][B] role 0's note about the proposal being pre-drawn
    int *mvsel;                                          // [2][B] 1: the pending descriptor is mvfix, 0: mv[buf]
    double *Dpart;                                       // [B][nrb_d][2]
    double *Down;                                        // [2][2][B][2] paired form: own-rows log-ratio of the speculative /
                                                         // re-drawn E->I proposal (parity, which, chain, {theta, const})
    unsigned *sweep;                                     // [B] sweeps done (device resident: graph replays advance it)
    unsigned *slot0;                                     // [1] sweep index of trace slot 0
    // traces
    double *tr_theta;                                    // [cap][B][P]
    void *tr_events;                                     // [cap][B][M][T][3] int32, or uint16 when SamplerCfg::ev16
    unsigned *ev_overflow;                               // [1] set when a count did not fit the 16-bit trace
    double *tr_hmc;                                      // [cap][B][3]  is_accepted, target_log_prob, step_size
    double *tr_mv;                                       // [cap][B][4][NMVTR]
};

// ---------------------------------------------------------------------------------------------
// Hand-off words (k_leap).  A value handed from one workgroup of a launch to another through the XCD's L2 used to cost the
// producer its stores, the wait for their acknowledgement (s_waitcnt vmcnt(0), ~0.3 us), a returning atomic on the chain's
// counter (~0.4 us) and -- for the last one in -- a flag store; the consumer a poll of the flag (a round trip, ~0.5 us) and only
// then the loads of the values (another).  Here every value carries its own flag: a double travels as 16 bytes
// {low word, seq, high word, seq} -- two aligned 8-byte halves, each written whole by the memory system, each with the step
// number -- so the producer only issues its stores and the consumer's first look at the DATA is also its wait: it loads past
// the L1 and looks again until both halves of everything it asked for show the step.  seq = the step's number over all
// launches of the sampler with the top bit set: never the zero of a reset buffer, never the number of the step before.
// No release / acquire anywhere: all of a chain's workgroups share one L2 (checked at creation, as for every hand-off here).
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned ll_seq(unsigned long long step) { return (unsigned)step | 0x80000000u; }
__device__ __forceinline__ void ll_store(uint4 *p, double v, unsigned seq) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    uint4 x;
    x.x = (unsigned)u; x.y = seq; x.z = (unsigned)(u >> 32); x.w = seq;
    *p = x;                                                 // one global_store_dwordx4
}
__device__ __forceinline__ bool ll_ok(const u32x4 &x, unsigned seq) { return x.y == seq && x.w == seq; }
__device__ __forceinline__ double ll_value(const u32x4 &x) {
    return __longlong_as_double((long long)(((unsigned long long)x.z << 32) | (unsigned long long)x.x));
}
// N 16-byte loads past the L1 and the wait for them, as ONE asm block: the compiler does not count these loads, so nothing
// may touch the destination registers between the issue and the wait
template <int N> __device__ __forceinline__ void ll_load(const uint4 *const (&p)[N], u32x4 (&x)[N]) {
    static_assert(N >= 1 && N <= 6, "groups of up to six");
    if constexpr (N == 1)
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x[0]) : "v"(p[0]) : "memory");
    else if constexpr (N == 2)
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(x[0]), "=&v"(x[1]) : "v"(p[0]), "v"(p[1]) : "memory");
    else if constexpr (N == 3)
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\t"
                     "s_waitcnt vmcnt(0)" : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]) : "v"(p[0]), "v"(p[1]), "v"(p[2]) : "memory");
    else if constexpr (N == 4)
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                     "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
    else if constexpr (N == 5)
        asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                     "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4])
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %7, off sc1\n\tglobal_load_dwordx4 %2, %8, off sc1\n\t"
                     "global_load_dwordx4 %3, %9, off sc1\n\tglobal_load_dwordx4 %4, %10, off sc1\n\tglobal_load_dwordx4 %5, %11, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5])
                     : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]) : "memory");
}
// The consumer's wait: load, look, again -- until every lane of the wave has the step's words in all N places.  Bounded like
// every wait of k_leap (leap_wait): a time-out is counted in the chain's fatal counter, and once that is non-zero every wait
// of the chain gives up at its next look at it; the values are then whatever was there -- the host discards the burst.
// (Looking again at only the places that were late, one load at a time, was slower -- 140.7 us per launch against 134.7: with
// several tiles late, each place's round trip came behind the last one's.)
#ifndef LEAP_ROLE_PRIO
#define LEAP_ROLE_PRIO 3            // k_leap's role waves (see there)
#endif
#ifdef LL_POLL_DROP_PRIO
#define LL_RPRIO LEAP_ROLE_PRIO
#else
#define LL_RPRIO -1
#endif
// RPRIO >= 0 (a role's waves): the wave's priority is dropped while it looks again and again, and put back to RPRIO when the
// words are there
template <int N, int RPRIO = -1> __device__ __forceinline__ void ll_poll(const uint4 *const (&p)[N], unsigned seq, unsigned *late, double (&v)[N]) {
    u32x4 x[N];
    int spins = 0;
    for (;;) {
        ll_load<N>(p, x);
        bool ok = true;
#pragma unroll
        for (int j = 0; j < N; ++j) ok = ok && ll_ok(x[j], seq);
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
        if (RPRIO >= 0 && spins == 0) __builtin_amdgcn_s_setprio(0);
#ifndef LL_POLL_SLEEP
#define LL_POLL_SLEEP 2
#endif
        __builtin_amdgcn_s_sleep(LL_POLL_SLEEP);
        ++spins;
        if ((spins & 63) == 0 && __hip_atomic_load(late, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (spins > (1 << 19)) { if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(late, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    if (RPRIO >= 0 && spins > 0) __builtin_amdgcn_s_setprio(RPRIO);
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = ll_value(x[j]);
}

__device__ inline RngKey rng_key(const SamplerCfg &s, const Chains &ch, int b) {
    return RngKey{s.k0, s.k1, (uint32_t)(s.chain0 + b), ch.sweep[b]};
}

// ---------------------------------------------------------------------------
// events fp64 [B][M][T][3]  <->  int32 planes
// ---------------------------------------------------------------------------
__global__ void k_import_events(Dims d, Work w, const double *__restrict__ events, int B) {
    const size_t n = (size_t)B * d.M * d.T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % d.T);
        const int m = (int)((i / d.T) % d.M);
        const int b = (int)(i / ((size_t)d.T * d.M));
        const size_t q = ((size_t)b * d.Mp + m) * d.Tp + t;
#pragma unroll
        for (int x = 0; x < 3; ++x) w.K[x][q] = (int)events[i * 3 + x];
    }
}

__global__ void k_export_events(Dims d, Work w, double *__restrict__ events, int B) {
    const size_t n = (size_t)B * d.M * d.T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % d.T);
        const int m = (int)((i / d.T) % d.M);
        const int b = (int)(i / ((size_t)d.T * d.M));
        const size_t q = ((size_t)b * d.Mp + m) * d.Tp + t;
#pragma unroll
        for (int x = 0; x < 3; ++x) events[i * 3 + x] = (double)w.K[x][q];
    }
}

// ---------------------------------------------------------------------------
// HMC.  PreconditionedHamiltonianMonteCarlo with a diagonal mass matrix
// M = diag(1/var) (mcmc_kernel_factory.py:14-29; num_leapfrog_steps=16,
// inference.py:324-329), DualAveragingStepSizeAdaptation (:32-44) and
// DiagonalMassMatrixAdaptation (:47-60) as re-stated in DESIGN.md.
//
// One 512-thread workgroup per chain.  The kernel is a latency chain, so it is
// organised around few global-memory round trips: every thread owns at most HT
// days and HM rows -- i.e. the alpha_t / spatial_effect entries, their momenta,
// and the table entries ea[t], rir[t], eb[m] derived from them -- issues all its
// loads up front, and keeps gradient, momentum and position in registers:
//   phase 1  reduce k_se's partials  -> d/d alpha_t (suffix scan of the column sums),
//            d/d spatial (row sums), and six block-reduced scalars
//   phase 2  leapfrog kick/drift on the owned entries (STAGE 0 draws the momentum,
//            STAGE 2 finishes the trajectory: accept/reject, adaptation, trace)
//   phase 3  tables and priors for the new position (prefix scan for alpha, CAR matvec
//            from an LDS copy of spatial_effect)
// STAGE 0: first kernel of a trajectory; STAGE 1: interior leapfrog; STAGE 2: last.
// ---------------------------------------------------------------------------
constexpr int HB = 512;             // threads (8 waves: leaves 256 VGPRs per lane, no spills)
constexpr int HWV = HB / WAVE;      // waves
// HT / HM (template parameters): days / rows per thread, ceil(Tp/HB) in {1,2}, ceil(M/HB) in {1,2,4}
constexpr int NRED = 8;
#ifdef SEIR_STAMPS
#ifndef SEIR_STAMP_STAGE
#define SEIR_STAMP_STAGE 2
#endif
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && STAGE == SEIR_STAMP_STAGE) ((unsigned long long *)(hs + 16))[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP_DRAIN(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); STAMP(i); } while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMP_DRAIN(i) do {} while (0)
#endif

// standard normal for component i of the momentum (Box-Muller; components 2j, 2j+1 share a Philox call)
__device__ SEIR_COLD double momentum_normal(RngKey key, int i) {
    double u1, u2;
    rng_uniform2(key, RS_MOMENTUM, (uint32_t)(i >> 1), u1, u2);
    const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586 * u2;
    return (i & 1) ? rad * sin(ang) : rad * cos(ang);
}

// sum NV values over the block; results replicated in every thread.  sh: [HWV][NRED]
template <int NV>
__device__ __forceinline__ void block_sum_vec(double (&v)[NV], double *sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    lds_barrier();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) sh[wave * NRED + k] = v[k];
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < HWV; ++j) a += sh[j * NRED + k];
        v[k] = a;
    }
}

// exclusive prefix (in thread order) over the block; sh: [HWV]
__device__ __forceinline__ double block_excl_scan_hb(double v, double *sh, double &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double inc = wave_incl_scan(v, lane);
    lds_barrier();
    if (lane == 63) sh[wave] = inc;
    lds_barrier();
    double base = 0.0, tot = 0.0;
#pragma unroll
    for (int j = 0; j < HWV; ++j) { const double x = sh[j]; tot += x; if (j < wave) base += x; }
    total = tot;
    return base + inc - v;
}

// inclusive suffix (thread i gets sum_{j >= i}); sh: [HWV]
__device__ __forceinline__ double block_incl_suffix_hb(double v, double *sh, double &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double inc = wave_incl_suffix_scan(v, lane);
    lds_barrier();
    if (lane == 0) sh[wave] = inc;
    lds_barrier();
    double tail = 0.0, tot = 0.0;
#pragma unroll
    for (int j = 0; j < HWV; ++j) { const double x = sh[j]; tot += x; if (j > wave) tail += x; }
    total = tot;
    return inc + tail;
}

// gather_qs != 0: (Q s) of the current position is gathered from q instead of read from w.Qs (the
// chunked leapfrog steps before this launch do not maintain w.Qs).
template <int STAGE, int HT, int HM>
__global__ __launch_bounds__(HB) void k_hmc_step(Dims d, Consts c, Work w, SamplerCfg s, Chains ch, int gather_qs) {
    extern __shared__ double lds_sp[];                 // [Mp] spatial_effect of the new position
    __shared__ double red[HWV * NRED];
    __shared__ double scn[HWV];
    __shared__ double bc[12];                          // broadcast scalars
    __shared__ double bc2[2];
    __shared__ double zz[6];                           // stage 0: standard normals of the six global parameters
    __shared__ int s_accept;
    __shared__ double2 ltab[LDSTAB_N];
    const int b = d.b0 + blockIdx.x, tid = threadIdx.x;
    const int T = d.T, M = d.M;
    double *q = ch.q + (size_t)b * d.Pp, *p = ch.p + (size_t)b * d.Pp, *q0 = ch.q0 + (size_t)b * d.Pp;
    double *var = ch.var + (size_t)b * d.Pp;
    double *hs = ch.hs + (size_t)b * NHS;
    double *sc = w.scal + (size_t)b * NSCAL;
    const int oT = 6 - 1, oM = 6 + T - 1;              // alpha_t[t-1] at oT + t ; spatial[m] at oM + m

    STAMP(0);
    // ---------------- phase 0: all loads --------------------------------------
    if (tid < LOGTAB_N) ltab[tid] = c.logtab[tid];
    const double eps = hs[HS_EPS];
    const double psi = sc[SC_PSI], sig = sc[SC_SIG], beta = sc[SC_BETA], g0 = sc[SC_G0], g1 = sc[SC_G1],
                 a0 = sc[SC_A0], s0 = sc[SC_S0], s1 = sc[SC_S1], prior = sc[SC_PRIOR], jac = sc[SC_JAC];
    double kir[HT], dir[HT], rate[HT], wdt[HT], col[HT], qa[HT], pa[HT], va[HT];
    double Rm[HM], lam[HM], qs[HM], qm[HM], pm[HM], vm[HM], inN[HM];
    const int ntile = d.nmt * d.ntc;
    double lpart = 0.0, ppart = 0.0;
    for (int i = tid; i < ntile; i += HB) { lpart += w.Lpart[(size_t)b * ntile + i]; ppart += w.Ppart[(size_t)b * ntile + i]; }
    STAMP_DRAIN(10);
#pragma unroll
    for (int k = 0; k < HT; ++k) {
        const int t = tid + k * HB;
        kir[k] = dir[k] = rate[k] = wdt[k] = col[k] = qa[k] = pa[k] = 0.0; va[k] = 1.0;
        if (t < d.Tp) {
            kir[k] = w.Kir[(size_t)b * d.Tp + t];
            dir[k] = w.Dir[(size_t)b * d.Tp + t];
            rate[k] = w.rir[(size_t)b * d.Tp + t];
            wdt[k] = c.wd[t];
            if (t >= 1 && t < T) { qa[k] = q[oT + t]; pa[k] = STAGE == 0 ? 0.0 : p[oT + t]; va[k] = var[oT + t]; }
            const double *kp = w.Kpart + (size_t)b * d.nmt * d.Tp + t;
            double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
            if (d.nmt <= 32) {                           // one batch of loads, no loop-carried waits
                double v[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) v[j] = j < d.nmt ? kp[(size_t)j * d.Tp] : 0.0;
#pragma unroll
                for (int j = 0; j < 32; j += 4) { c0 += v[j]; c1 += v[j + 1]; c2 += v[j + 2]; c3 += v[j + 3]; }
            } else {
                for (int ty0 = 0; ty0 < d.nmt; ty0 += 16) {
                    double v[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = ty0 + j < d.nmt ? kp[(size_t)(ty0 + j) * d.Tp] : 0.0;
#pragma unroll
                    for (int j = 0; j < 16; j += 4) { c0 += v[j]; c1 += v[j + 1]; c2 += v[j + 2]; c3 += v[j + 3]; }
                }
            }
            col[k] = (c0 + c1) + (c2 + c3);
        }
    }
    STAMP_DRAIN(11);
#pragma unroll
    for (int k = 0; k < HM; ++k) {
        const int m = tid + k * HB;
        Rm[k] = lam[k] = qs[k] = qm[k] = pm[k] = inN[k] = 0.0; vm[k] = 1.0;
        if (m < M) {
            lam[k] = c.la[m];
            inN[k] = c.invN[m];
            qs[k] = w.Qs[(size_t)b * d.Mp + m];
            qm[k] = q[oM + m]; pm[k] = STAGE == 0 ? 0.0 : p[oM + m]; vm[k] = var[oM + m];
            const double *rp = w.Rpart + (size_t)b * d.ntc * d.Mp + m;
            double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
            {                                            // ntc = Tp/64 <= 16 (T <= 1024)
                double v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = j < d.ntc ? rp[(size_t)j * d.Mp] : 0.0;
#pragma unroll
                for (int j = 0; j < 16; j += 4) { r0 += v[j]; r1 += v[j + 1]; r2 += v[j + 2]; r3 += v[j + 3]; }
            }
            Rm[k] = (r0 + r1) + (r2 + r3);
        }
    }
    // constants of phase 3 (CAR precision rows in ELL form): fetched now, used after the leapfrog
    // (not at HM = 4, M > 1024: 96 more registers per lane there, and the instance spilled 113-161 of them)
    constexpr int QPRE = HM <= 2 ? 8 : 1;
    double qell_v[HM][QPRE];
    int qell_c[HM][QPRE];
    const bool ell_pre = HM <= 2 && c.qw > 0 && c.qw <= QPRE;
#pragma unroll
    for (int k = 0; k < HM; ++k) {
        const int m = tid + k * HB;
#pragma unroll
        for (int j = 0; j < QPRE; ++j) {
            const bool on = ell_pre && m < M && j < c.qw;
            qell_v[k][j] = on ? c.Qell_val[(size_t)j * d.Mp + m] : 0.0;
            qell_c[k][j] = on ? c.Qell_col[(size_t)j * d.Mp + m] : 0;
        }
    }
    double q6[6], p6[6], v6[6];
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { q6[i] = q[i]; p6[i] = STAGE == 0 ? 0.0 : p[i]; v6[i] = var[i]; }
    }
    STAMP_DRAIN(12);
    // stage 0: the momentum draws depend on nothing loaded above -- done here they overlap the load latency
    // (the six global parameters by six lanes of the last wave instead of one lane six times)
    double zt[HT], zm[HM];
    if (STAGE == 0) {
        const RngKey key = rng_key(s, ch, b);
#pragma unroll
        for (int k = 0; k < HT; ++k) { const int t = tid + k * HB; zt[k] = (t >= 1 && t < T) ? momentum_normal(key, oT + t) : 0.0; }
#pragma unroll
        for (int k = 0; k < HM; ++k) { const int m = tid + k * HB; zm[k] = m < M ? momentum_normal(key, oM + m) : 0.0; }
        if (tid >= HB - 8 && tid < HB - 2) zz[tid - (HB - 8)] = momentum_normal(key, tid - (HB - 8));
    }
    if (gather_qs & 1) {                               // uniform branch
#pragma unroll
        for (int k = 0; k < HM; ++k) { const int m = tid + k * HB; if (m < M) lds_sp[m] = qm[k]; }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < HM; ++k) {
            const int m = tid + k * HB;
            if (m < M) {
                double acc = 0.0;
                if (ell_pre) {
#pragma unroll
                    for (int j = 0; j < QPRE; ++j) acc += qell_v[k][j] * lds_sp[qell_c[k][j]];
                } else {
                    for (int e = c.Qrow[m]; e < c.Qrow[m + 1]; ++e) acc += c.Qval[e] * lds_sp[c.Qcol[e]];
                }
                qs[k] = acc;
            }
        }
    }
    lds_barrier();                                     // ltab (and lds_sp reads done before it is rewritten)
    STAMP(1);

    // ---------------- phase 1: gradient at the current position ---------------
    // prior_here (stage 2 after a chunked last inner step, gather_qs bit 1): nobody left the priors and the Jacobian of
    // the end point in the scalar block, so they are computed here -- two more sums in the same reduction
    constexpr int NRV = STAGE == 2 ? 8 : 6;
    const bool prior_here = STAGE == 2 && (gather_qs & 2) != 0;
    double rv[NRV] = {lpart, 0.0, 0.0, 0.0, 0.0, ppart};  // lik, gg0, gg1, gsig, gbeta, gpsi [, sum alpha_t^2, s' Q s]
    if (STAGE == 2 && prior_here) {
#pragma unroll
        for (int k = 0; k < HT; ++k) rv[NRV - 2] += qa[k] * qa[k];
#pragma unroll
        for (int k = 0; k < HM; ++k) rv[NRV - 1] += qm[k] * qs[k];
    }
#pragma unroll
    for (int k = 0; k < HT; ++k) {
        const int t = tid + k * HB;
        if (t < T) {
            const double r = rate[k] * d.dt;
            double L, inv;
            l1me_inv_wide(r, L, inv, ltab);
            rv[0] += (kir[k] != 0.0 ? kir[k] * L : 0.0) - dir[k] * r;
            const double gr = d.dt * ((kir[k] != 0.0 ? kir[k] * inv : 0.0) - dir[k]);
            rv[1] += gr * rate[k];
            rv[2] += gr * rate[k] * wdt[k];
        }
    }
#pragma unroll
    for (int k = 0; k < HM; ++k) {
        rv[3] += qm[k] * Rm[k];
        rv[4] += lam[k] * Rm[k];
    }
    STAMP(2);
    block_sum_vec<NRV>(rv, red);
    STAMP(3);
    double prior_v = prior, jac_v = jac;
    if (STAGE == 2 && prior_here) {
        // model_spec.py:140-198 and the bijector's Jacobian (inference.py:555-557), as phase 3 writes them
        const double e0 = 2.220446049250313e-16;
        double lp = d.prior_const;
        lp += -0.5 * a0 * a0 / 100.0 - 0.5 * beta * beta;
        lp += 2.0 * cold_log(psi) - 10.0 * psi;
        lp += -0.5 * rv[NRV - 2] / (0.005 * 0.005);
        lp += -sig * sig / 0.02;
        lp += -0.5 * rv[NRV - 1];
        lp += -0.5 * (g0 * g0 + g1 * g1) / 1.0e4;
        prior_v = lp;
        jac_v = (q[0] - (psi - e0)) + (q[1] - (sig - e0));      // log sigmoid(u) = u - softplus(u), softplus(u) = value - eps
    }
    const double lp_theta = rv[0] + prior_v + jac_v;
    // d/d alpha_t[t-1] = sum_{t' >= t} col[t']: suffix scan, chunks from the back
    double ga[HT], gtot = 0.0;
    {
        double tail = 0.0;
#pragma unroll
        for (int k = HT - 1; k >= 0; --k) {
            double tot;
            ga[k] = block_incl_suffix_hb(col[k], scn, tot) + tail;
            tail += tot;
        }
        gtot = tail;
    }
#pragma unroll
    for (int k = 0; k < HT; ++k) {
        const int t = tid + k * HB;
        ga[k] = (t >= 1 && t < T) ? ga[k] - qa[k] / (0.005 * 0.005) : 0.0;   // + prior gradient; t=0 owns no alpha_t
    }
    double gm[HM];
#pragma unroll
    for (int k = 0; k < HM; ++k) gm[k] = sig * Rm[k] - qs[k];
    double g6[6];
    g6[0] = (rv[5] + 2.0 / psi - 10.0) * s0 + (1.0 - s0);
    g6[1] = (rv[3] - sig / 0.01) * s1 + (1.0 - s1);
    g6[2] = rv[4] - beta;
    g6[3] = rv[1] - g0 / 1.0e4;
    g6[4] = rv[2] - g1 / 1.0e4;
    g6[5] = gtot - a0 / 100.0;

    STAMP(4);
    // ---------------- phase 2: leapfrog on the owned entries -------------------
    double kin = 0.0;                                   // kinetic energy contribution (STAGE 0: start, 2: end)
    if (STAGE == 0) {
#pragma unroll
        for (int k = 0; k < HT; ++k) {
            const int t = tid + k * HB;
            if (t >= 1 && t < T) {
                double pi = zt[k] / sqrt(va[k]);
                kin += 0.5 * va[k] * pi * pi;
                q0[oT + t] = qa[k];
                pi += 0.5 * eps * ga[k];
                pa[k] = pi;
                qa[k] += eps * va[k] * pi;
            }
        }
#pragma unroll
        for (int k = 0; k < HM; ++k) {
            const int m = tid + k * HB;
            if (m < M) {
                double pi = zm[k] / sqrt(vm[k]);
                kin += 0.5 * vm[k] * pi * pi;
                q0[oM + m] = qm[k];
                pi += 0.5 * eps * gm[k];
                pm[k] = pi;
                qm[k] += eps * vm[k] * pi;
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                double pi = zz[i] / sqrt(v6[i]);
                kin += 0.5 * v6[i] * pi * pi;
                q0[i] = q6[i];
                pi += 0.5 * eps * g6[i];
                p6[i] = pi;
                q6[i] += eps * v6[i] * pi;
            }
        }
    } else {
        const double kick = STAGE == 1 ? eps : 0.5 * eps;
#pragma unroll
        for (int k = 0; k < HT; ++k) {
            pa[k] += kick * ga[k];
            if (STAGE == 1) qa[k] += eps * va[k] * pa[k];
            else kin += 0.5 * va[k] * pa[k] * pa[k];
        }
#pragma unroll
        for (int k = 0; k < HM; ++k) {
            pm[k] += kick * gm[k];
            if (STAGE == 1) qm[k] += eps * vm[k] * pm[k];
            else kin += 0.5 * vm[k] * pm[k] * pm[k];
        }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                p6[i] += kick * g6[i];
                if (STAGE == 1) q6[i] += eps * v6[i] * p6[i];
                else kin += 0.5 * v6[i] * p6[i] * p6[i];
            }
        }
    }
    bool accepted = true;
    if (STAGE != 1) {
        double kv[1] = {kin};
        block_sum_vec<1>(kv, red);
        kin = kv[0];
    }
    if (STAGE == 0 && tid == 0) { hs[HS_LP0] = lp_theta; hs[HS_K0] = kin; }
    if (STAGE == 2) {
        if (tid == 0) {
            const RngKey key = rng_key(s, ch, b);
            double u1, u2;
            rng_uniform2(key, RS_HMC_ACCEPT, 0u, u1, u2);
            const double lp0 = hs[HS_LP0];
            const double log_ratio = (lp_theta - lp0) - (kin - hs[HS_K0]);
            const int acc = (cold_log(u1) < log_ratio && !(s.disable_mask & 1)) ? 1 : 0;   // NaN compares false -> reject
            s_accept = acc;
            hs[HS_ACC] = (double)acc;
            hs[HS_LOGACC] = log_ratio;
            const double lpt = acc ? lp_theta : lp0;
            hs[HS_LP_THETA] = lpt;
            double eps_traced = eps;
            if (s.adapt_step) {                       // dual averaging (Hoffman & Gelman alg. 5, TFP defaults)
                const double a = isfinite(log_ratio) ? fmin(1.0, cold_exp(log_ratio)) : 0.0;
                const double prev_step = hs[HS_DA_STEP];
                const double n = prev_step + 1.0;
                const double err = hs[HS_DA_ERR] + s.target_accept - a;
                const double log_step = hs[HS_DA_MU] - err * sqrt(n) / ((n + 10.0) * 0.05);
                const double eta = cold_exp(-0.75 * cold_log(n));
                const double log_avg = eta * log_step + (1.0 - eta) * hs[HS_DA_LOGAVG];
                hs[HS_DA_ERR] = err; hs[HS_DA_STEP] = n; hs[HS_DA_LOGAVG] = log_avg;
                if (prev_step <= (double)s.n_adapt) {
                    eps_traced = cold_exp(prev_step < (double)s.n_adapt ? log_step : log_avg);
                    hs[HS_EPS] = eps_traced;
                }
            }
            // trace_results_fn reads step_size from the kernel results AFTER DualAveragingStepSizeAdaptation
            // has written new_step_size back (inference.py:255-261): the traced value is the step size the
            // NEXT sweep will use, which is what run_mcmc averages over the last 25 draws (inference.py:439-441)
            const unsigned slot = ch.sweep[b] - ch.slot0[0];
            if (slot < (unsigned)s.cap) {
                double *tr = ch.tr_hmc + ((size_t)slot * s.B + b) * 3;
                tr[0] = (double)acc;
                tr[1] = lpt + hs[HS_LP_CONST];
                tr[2] = eps_traced;
            }
        }
        __syncthreads();
        accepted = s_accept != 0;
        if (!accepted) {                              // back to the start of the trajectory
#pragma unroll
            for (int k = 0; k < HT; ++k) { const int t = tid + k * HB; if (t >= 1 && t < T) qa[k] = q0[oT + t]; }
#pragma unroll
            for (int k = 0; k < HM; ++k) { const int m = tid + k * HB; if (m < M) qm[k] = q0[oM + m]; }
            if (tid == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) q6[i] = q0[i];
            }
        }
        if (s.adapt_mass) {                           // Welford update with the new state (ddof 0)
            const double n1 = hs[HS_RV_N] + 1.0;
            double *mean = ch.rv_mean + (size_t)b * d.Pp, *m2 = ch.rv_m2 + (size_t)b * d.Pp;
            auto upd = [&](int i, double x) {
                const double dlt = x - mean[i];
                const double mu = mean[i] + dlt / n1;
                const double ss = m2[i] + dlt * (x - mu);
                mean[i] = mu; m2[i] = ss;
                var[i] = ss / n1;
            };
#pragma unroll
            for (int k = 0; k < HT; ++k) { const int t = tid + k * HB; if (t >= 1 && t < T) upd(oT + t, qa[k]); }
#pragma unroll
            for (int k = 0; k < HM; ++k) { const int m = tid + k * HB; if (m < M) upd(oM + m, qm[k]); }
            if (tid == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) upd(i, q6[i]);
            }
            __syncthreads();
            if (tid == 0) hs[HS_RV_N] = n1;
        }
    }
    STAMP(5);
    // write back position / momentum
#pragma unroll
    // (stage 2 leaves q0 = q: the start point of the next trajectory, which a folded first step reads from q0)
    for (int k = 0; k < HT; ++k) {
        const int t = tid + k * HB;
        if (t >= 1 && t < T) { q[oT + t] = qa[k]; if (STAGE != 2) p[oT + t] = pa[k]; else q0[oT + t] = qa[k]; }
    }
#pragma unroll
    for (int k = 0; k < HM; ++k) {
        const int m = tid + k * HB;
        if (m < M) { q[oM + m] = qm[k]; if (STAGE != 2) p[oM + m] = pm[k]; else q0[oM + m] = qm[k]; lds_sp[m] = qm[k]; }
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { q[i] = q6[i]; if (STAGE != 2) p[i] = p6[i]; else q0[i] = q6[i]; }
        bc[2] = q6[2]; bc[3] = q6[3]; bc[4] = q6[4]; bc[5] = q6[5];
        bc[6] = q6[0]; bc[7] = q6[1];
    }
    lds_barrier();                                     // bc[6..7], lds_sp (LDS only: do not drain the stores)
    // the two softplus and what follows from them (log psi for the Gamma prior, the sigmoids of the
    // chain rule) are ~3 serial libm calls each: lane 0 of waves 0 and 1 take one parameter each
    if ((tid & 63) == 0 && (tid >> 6) < 2) {
        const int i = tid >> 6;
        const double e0 = 2.220446049250313e-16;
        const double u = bc[6 + i];
        const double sp = softplus(u);
        bc[i] = sp + e0;
        const double ls = u - sp;                      // log sigmoid(u) = u - softplus(u)
        bc[8 + i] = ls;
        bc[10 + i] = cold_exp(ls);
        if (i == 0) bc2[0] = cold_log(sp + e0);
    }
    lds_barrier();
    STAMP(6);

    // ---------------- phase 3: tables and priors at the new position -----------
    const double npsi = bc[0], nsig = bc[1], nbeta = bc[2], ng0 = bc[3], ng1 = bc[4], na0 = bc[5];
    double pr[2] = {0.0, 0.0};                         // sum alpha_t^2, s' Q s
    {
        double carry = na0;
#pragma unroll
        for (int k = 0; k < HT; ++k) {
            const int t = tid + k * HB;
            const double v = (t >= 1 && t < T) ? qa[k] : 0.0;
            double tot;
            const double acc = carry + block_excl_scan_hb(v, scn, tot) + v;   // alpha_0 + cumsum(alpha_t)[t-1]
            carry += tot;
            pr[0] += v * v;
            if (t < T) {
                w.ea[(size_t)b * d.Tp + t] = exp(acc);
                const double rnew = exp(ng0 + ng1 * wdt[k]);
                w.rir[(size_t)b * d.Tp + t] = rnew;
            }
            if (STAGE == 0 && d.chunked) {
                // hand-over to the chunked leapfrog steps: a_t, and per 64-day chunk (= one wave here)
                // the sums of alpha, v p and v alpha at the new position / momentum
                if (t < T) w.acur[(size_t)b * d.Tp + t] = acc;
                const double vv = (t >= 1 && t < T) ? va[k] : 0.0;
                const double ca = wave_sum(v), cvp = wave_sum(vv * pa[k]), cva = wave_sum(vv * v);
                // ... and the chunk's part of the I->R term's d/d gamma0, d/d gamma1 at the new rates (each chunked step
                // leaves the same for the next one: a step then adds ntc pairs instead of evaluating every day again)
                double cg0 = 0.0, cg1 = 0.0;
                if (t < T) {
                    const double rnew = exp(ng0 + ng1 * wdt[k]);
                    double L, inv;
                    l1me_inv_wide(rnew * d.dt, L, inv, ltab);
                    const double grr = d.dt * ((kir[k] != 0.0 ? kir[k] * inv : 0.0) - dir[k]);
                    cg0 = grr * rnew;
                    cg1 = grr * rnew * wdt[k];
                }
                cg0 = wave_sum(cg0); cg1 = wave_sum(cg1);
                const int chunk = (tid >> 6) + k * (HB / WAVE);
                if ((tid & 63) == 0 && chunk < d.ntc) {
                    double *ct = w.CT + (((size_t)b * 2 + 0) * CT_MAXC + chunk) * 4;
                    ct[0] = ca; ct[1] = cvp; ct[2] = cva;
                    double *cg = w.CG + (((size_t)b * 2 + 0) * CT_MAXC + chunk) * 2;
                    cg[0] = cg0; cg[1] = cg1;
                }
            }
        }
        if (STAGE == 0 && d.chunked) {
            // V(t) = sum_{s=1..t} var[alpha_t[s-1]]: constant over the trajectory
            double vcarry = 0.0;
#pragma unroll
            for (int k = 0; k < HT; ++k) {
                const int t = tid + k * HB;
                const double vv = (t >= 1 && t < T) ? va[k] : 0.0;
                double tot;
                const double inc = vcarry + block_excl_scan_hb(vv, scn, tot) + vv;
                vcarry += tot;
                if (t < d.Tp) w.Vt[(size_t)b * d.Tp + t] = inc;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < HM; ++k) {
        const int m = tid + k * HB;
        if (m < M) {
            w.eb[(size_t)b * d.Mp + m] = exp(nbeta * lam[k] + nsig * qm[k]) * inN[k];
            if (w.sp != nullptr) w.sp[((size_t)b * 2 + 0) * d.Mp + m] = qm[k];     // buffer 0: see k_hmc_chunk
            double acc = 0.0;
            if (ell_pre) {
#pragma unroll
                for (int j = 0; j < QPRE; ++j) acc += qell_v[k][j] * lds_sp[qell_c[k][j]];
            } else if (c.qw > 0) {
                for (int e0 = 0; e0 < c.qw; e0 += 8) {
                    double qv[8]; int qc[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool on = e0 + j < c.qw;
                        qv[j] = on ? c.Qell_val[(size_t)(e0 + j) * d.Mp + m] : 0.0;
                        qc[j] = on ? c.Qell_col[(size_t)(e0 + j) * d.Mp + m] : 0;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc += qv[j] * lds_sp[qc[j]];
                }
            } else {
                for (int e = c.Qrow[m]; e < c.Qrow[m + 1]; ++e) acc += c.Qval[e] * lds_sp[c.Qcol[e]];
            }
            w.Qs[(size_t)b * d.Mp + m] = acc;
            pr[1] += qm[k] * acc;
        }
    }
    STAMP(7);
    block_sum_vec<2>(pr, red);
    STAMP(8);
    if (tid == 0) {
        // model_spec.py:140-198; the parameter-free normalisers are folded into d.prior_const
        double lp = d.prior_const;
        lp += -0.5 * na0 * na0 / 100.0 - 0.5 * nbeta * nbeta;
        lp += 2.0 * bc2[0] - 10.0 * npsi;
        lp += -0.5 * pr[0] / (0.005 * 0.005);
        lp += -nsig * nsig / 0.02;
        lp += -0.5 * pr[1];
        lp += -0.5 * (ng0 * ng0 + ng1 * ng1) / 1.0e4;
        sc[SC_PSI] = npsi; sc[SC_SIG] = nsig; sc[SC_BETA] = nbeta; sc[SC_G0] = ng0; sc[SC_G1] = ng1;
        sc[SC_A0] = na0;
        sc[SC_S0] = bc[10]; sc[SC_S1] = bc[11];
        sc[SC_PRIOR] = lp;
        sc[SC_JAC] = bc[8] + bc[9];
        if (STAGE == 0 && d.chunked) {
            double *g = w.gst + ((size_t)b * 2 + 0) * GST_N;
#pragma unroll
            for (int i = 0; i < 6; ++i) { g[i] = q6[i]; g[6 + i] = p6[i]; }
            g[12] = npsi; g[13] = nsig; g[14] = bc[10]; g[15] = bc[11];
        }
    }
    STAMP(9);
    if (STAGE == 2) {
        // constrained draw -> trace (param_bijector.inverse(draws[0]), inference.py:375)
        const unsigned slot = ch.sweep[b] - ch.slot0[0];
        if (slot < (unsigned)s.cap) {
            double *tr = ch.tr_theta + ((size_t)slot * s.B + b) * d.P;
#pragma unroll
            for (int k = 0; k < HT; ++k) { const int t = tid + k * HB; if (t >= 1 && t < T) tr[oT + t] = qa[k]; }
#pragma unroll
            for (int k = 0; k < HM; ++k) { const int m = tid + k * HB; if (m < M) tr[oM + m] = qm[k]; }
            if (tid == 0) {
                tr[0] = npsi; tr[1] = nsig;
#pragma unroll
                for (int i = 2; i < 6; ++i) tr[i] = q6[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_hmc_chunk: one inner leapfrog step (full kick at the current position, drift to the next,
// tables of the next) split over independent single-wave workgroups instead of one workgroup
// per chain.  grid (ntc + Mp/64, chains) x 64 threads:
//   T-chunk c (blockIdx.x < ntc): days 64c..64c+63 -- alpha_t entries, a_t, exp(a_t), the I->R
//     rate table; chunk 0 also integrates alpha_0, gamma0, gamma1.
//   M-chunk c: rows 64c..64c+63 -- spatial effects, exp(beta l + sigma s)/N; chunk 0 also
//     integrates psi, sigma_space, beta_area (unconstrained u0, u1).
// What couples the chunks is carried by scalars, so no workgroup waits for another:
//   * k_se's tile scalars (Work::TS): per (row tile, day chunk) sum_t col, sum_t col V(t),
//     sum_m l_m row, sum_m s_m row.  With G(s) = sum_{tau >= s} col[tau] the alpha_t gradient is
//     G(s) - alpha_s / 0.005^2, and for a whole chunk c'
//         sum_{s in c'} v_s G(s) = [A(c') - V0(c') B(c')] + Vtot(c') sum_{c'' > c'} B(c'')
//     (A, B = the tile scalars summed over row tiles, V = prefix sums of the mass-matrix
//     variances, constant over a trajectory): a chunk gets the prefix of the NEW alpha over the
//     chunks before it without seeing their entries.
//   * per-chunk sums of alpha, v p, v alpha at the current position (Work::CT), written by each
//     T-chunk for the next step, double-buffered by the step's parity `par` like the spatial
//     effects (Work::sp, gathered for the CAR term) and the global parameters (Work::gst).
// Arithmetic per entry is that of k_hmc_step<1>; sums are associated differently (~1e-16).
// ---------------------------------------------------------------------------------------------
// NTC: the number of 64-day chunks at compile time (0 = any, loops stay rolled).  The kernel is one
// wave of straight-line code executed once, so its cost is its instruction count.
// COH: the partial sums read here (Kpart, TS, Rpart, Ppart) were written by OTHER workgroups of the SAME launch
// (k_se_chunk below) -- they are read past the L1 (agent-scope loads); see k_se_chunk for why that is enough.
// bx: chunk role (T-chunks first), b: chain; executed by one wave (threadIdx.x < 64).
// wait(): called once, after every load that does not depend on this launch's partial sums has been issued and
// before the first one that does (k_se_chunk: the spin on the chain's tile counter goes there, so those loads and
// the wait overlap); a no-op in k_hmc_chunk.
// role_gather (k_leap): the loads of a chunk role that wait for the step's tiles, shared by the FOUR waves of the role's
// workgroup.  A load past the L1 (what a role must use for anything another workgroup of the launch wrote) is not pipelined:
// one wave gets them back ~47 ns apart whatever it has in flight (timeline: 24 column-sum rows 1.40 us, 12 rows 0.84, 6 rows
// 0.56), and a T-chunk has 36 of them, an M-chunk of UK-380 51 -- most of the 2.9 us a role took.  The three waves that used
// to retire at once now fetch a quarter each and leave what wave 0 needs in LDS, already summed where the order of k_hmc_chunk's
// additions allows it: bit-identical to the other forms.
//   T-chunk: wave w accumulates the column sums of row tiles w, w+4, w+8 ... (k_hmc_chunk's accumulator c_w) and fetches the
//            tile scalars of day chunks w, w+4, w+8.
//   M-chunk: wave 1 the row partials of the chunk's own rows, wave 2 the per-tile psi partials (and row scalars), and for the
//            small-M form (the M-chunks sum the row partials of ALL rows themselves) rows kk of a lane by waves 1,2,3,1,2,3,1,2.
template <int NC> struct RoleGather {                           // offsets (doubles) into the role's LDS block
    static constexpr int C = 0, BS = 4 * WAVE, AS = BS + NC * WAVE;                              // T-chunk
    static constexpr int X = 0, PS = NC * WAVE, RL = PS + WAVE, RS = RL + WAVE, ACC = RS + WAVE, SX = ACC + 8 * WAVE,
                         QS = SX + 8 * WAVE;                                                     // M-chunk
    static constexpr int SIZE = (4 + 2 * NC) * WAVE > (NC + 3 + 17) * WAVE ? (4 + 2 * NC) * WAVE : (NC + 3 + 17) * WAVE;
};
__device__ __forceinline__ int role_gather_row_wave(int kk) { return 1 + kk % 3; }     // rows of a lane by waves 1, 2, 3, 1, 2, 3, 1, 2
// (the role's own wave takes none of them: it is the last to arrive -- with two of the six row blocks on it the launch took
// 138.5 us against 135.1)
__device__ __forceinline__ int role_gather_acc_wave(int kk) { return role_gather_row_wave(kk); }
// ... and, before the wait for the tiles (but after the previous step's roles are done), the rows' spatial effects at the
// current position for the same rows: 8 more loads past the L1 that wave 0 no longer issues on its way to the wait
// ... and the CAR term (Q s)_m of the chunk's own rows at the current position, by the helper wave with the fewest rows to
// gather (wave 3): two dependent round trips (the row's columns, then the spatial effects there) that the role's own wave made
// on its way to the wait -- it reached it 1.3 us after the tiles were in.  Same operands in the same order: the same bits.
__device__ __forceinline__ bool role_qs_by_helper(const Consts &c) { return c.qw > 0 && c.qw <= 8; }
template <int NTC>
__device__ __forceinline__ void role_pregather(const Dims &d, const Consts &c, const double *spr, int bx, int wv, int lane, double *g) {
    constexpr int NC = NTC > 0 ? NTC : CT_MAXC;
    using G = RoleGather<NC>;
    const int ntc = NTC > 0 ? NTC : d.ntc;
    if (bx < ntc) return;
    if (wv == 3 && role_qs_by_helper(c)) {
        const int m = (bx - ntc) * WAVE + lane;
        const bool own = m < d.M;
        const int mc = own ? m : 0;
        double qv[8]; int qc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool on = j < c.qw;
            const size_t qi = (size_t)(on ? j : 0) * d.Mp + mc;
            const double qv_ = c.Qell_val[qi];
            const int qc_ = c.Qell_col[qi];
            qv[j] = on ? qv_ : 0.0;
            qc[j] = on ? qc_ : 0;
        }
        double sv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[j] = __hip_atomic_load(spr + qc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double Qs = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) Qs += qv[j] * sv[j];
        g[G::QS + lane] = own ? Qs : 0.0;
    }
    if (d.chunked != 1) return;
    const int nrow = (d.M + WAVE - 1) / WAVE;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        if (role_gather_row_wave(kk) != wv || kk >= nrow) continue;
        const int mm = lane + kk * WAVE;
        const bool on = mm < d.M;
        const double sv_ = __hip_atomic_load(spr + (on ? mm : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g[G::SX + kk * WAVE + lane] = on ? sv_ : 0.0;
    }
}
// What a role needs to know about the step's hand-off words: the number the tiles' words of THIS step carry and their buffer
// (steps alternate), the number the role's own tables for the NEXT step must carry, the chain's time-out counter
struct LeapLL {
    unsigned seq_in = 0, seq_out = 0;
    int pb = 0;
    unsigned *late = nullptr;
};
__device__ __forceinline__ size_t ll_tab_len(const Dims &d) { return (size_t)d.Tp + 2 * (size_t)d.Mp + 8; }
// N places in groups of up to six loads
template <int N, int RPRIO = -1> __device__ __forceinline__ void ll_poll_many(const uint4 *const (&p)[N], unsigned seq, unsigned *late, double (&v)[N]) {
    if constexpr (N <= 6) {
        ll_poll<N, RPRIO>(p, seq, late, v);
    } else {
        constexpr int H = N / 2;
        const uint4 *pa[H], *pb_[N - H];
        double va[H], vb[N - H];
#pragma unroll
        for (int j = 0; j < H; ++j) pa[j] = p[j];
#pragma unroll
        for (int j = 0; j < N - H; ++j) pb_[j] = p[H + j];
        ll_poll_many<H, RPRIO>(pa, seq, late, va);
        ll_poll_many<N - H, RPRIO>(pb_, seq, late, vb);
#pragma unroll
        for (int j = 0; j < H; ++j) v[j] = va[j];
#pragma unroll
        for (int j = 0; j < N - H; ++j) v[H + j] = vb[j];
    }
}
template <int NTC>
__device__ __forceinline__ void role_gather(const Dims &d, const Chains &ch, const LeapLL &ll, int b, int bx, int wv, int lane, double *g) {
    constexpr int NC = NTC > 0 ? NTC : CT_MAXC;
    using G = RoleGather<NC>;
    const int nmt = d.nmt, ntc = NTC > 0 ? NTC : d.ntc, ntile = nmt * ntc, M = d.M;
    const int nmt16 = d.Mp / 16;                                 // row tiles the arrays are laid out for
    const size_t cb = (size_t)b * 2 + ll.pb;                     // (chain, parity)
    const uint4 *TS = ch.llTS + cb * ((size_t)ntc * nmt16) * 4;
    if (bx < ntc) {
        const int t = bx * WAVE + lane;
        const uint4 *kp = ch.llK + cb * nmt16 * d.Tp + t;
        double cw = 0.0;
        if (nmt <= 16) {                                         // (uniform) four row tiles per wave
            const uint4 *pp[4];
            double x[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) pp[jj] = kp + (size_t)min(wv + 4 * jj, nmt - 1) * d.Tp;
            ll_poll<4, LL_RPRIO>(pp, ll.seq_in, ll.late, x);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) cw += wv + 4 * jj < nmt ? x[jj] : 0.0;
        } else {
            for (int j0 = 0; j0 < nmt; j0 += 24) {
                const uint4 *pp[6];
                double x[6];
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) pp[jj] = kp + (size_t)min(j0 + wv + 4 * jj, nmt - 1) * d.Tp;
                ll_poll<6, LL_RPRIO>(pp, ll.seq_in, ll.late, x);
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) cw += j0 + wv + 4 * jj < nmt ? x[jj] : 0.0;
            }
        }
        constexpr int NI = (NC + 3) / 4;
        const uint4 *tp[2 * NI];
        double ba[2 * NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int cc = wv + 4 * i;
            const uint4 *tp_ = TS + ((size_t)min(lane, nmt - 1) * ntc + min(cc, ntc - 1)) * 4;
            tp[2 * i] = tp_; tp[2 * i + 1] = tp_ + 1;
        }
        ll_poll_many<2 * NI, LL_RPRIO>(tp, ll.seq_in, ll.late, ba);
        g[G::C + wv * WAVE + lane] = cw;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int cc = wv + 4 * i;
            const bool on = cc < ntc && lane < nmt;
            if (cc < NC) { g[G::BS + cc * WAVE + lane] = on ? ba[2 * i] : 0.0; g[G::AS + cc * WAVE + lane] = on ? ba[2 * i + 1] : 0.0; }
        }
    } else {
        const int ci = bx - ntc, m = ci * WAVE + lane;
        const bool own = m < M;
        const bool rows_here = d.chunked == 1;
        const uint4 *R = ch.llR + cb * ntc * d.Mp;
        if (wv == 1) {
            const uint4 *rp[NC];
            double x[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) rp[j] = R + (size_t)min(j, ntc - 1) * d.Mp + (own ? m : 0);
            ll_poll_many<NC, LL_RPRIO>(rp, ll.seq_in, ll.late, x);
#pragma unroll
            for (int j = 0; j < NC; ++j) g[G::X + j * WAVE + lane] = (own && j < ntc) ? x[j] : 0.0;
        }
        if (wv == 2) {
            const uint4 *P = ch.llP + cb * ((size_t)ntc * nmt16);
            double ps = 0.0, rl = 0.0, rs = 0.0;
            for (int i0 = lane; i0 < ntile; i0 += 4 * WAVE) {
                const uint4 *pp[4];
                double x[4], y[4], z[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) pp[j] = P + (i0 + j * WAVE < ntile ? i0 + j * WAVE : 0);
                ll_poll<4, LL_RPRIO>(pp, ll.seq_in, ll.late, x);
#pragma unroll
                for (int j = 0; j < 4; ++j) { if (!(i0 + j * WAVE < ntile)) x[j] = 0.0; y[j] = 0.0; z[j] = 0.0; }
                if (!rows_here) {                                // uniform
                    const uint4 *yp[4], *zp[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ic = i0 + j * WAVE < ntile ? i0 + j * WAVE : 0;
                        yp[j] = TS + (size_t)ic * 4 + 2; zp[j] = TS + (size_t)ic * 4 + 3;
                    }
                    ll_poll<4, LL_RPRIO>(yp, ll.seq_in, ll.late, y);
                    ll_poll<4, LL_RPRIO>(zp, ll.seq_in, ll.late, z);
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (!(i0 + j * WAVE < ntile)) { y[j] = 0.0; z[j] = 0.0; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { ps += x[j]; rl += y[j]; rs += z[j]; }
            }
            g[G::PS + lane] = ps; g[G::RL + lane] = rl; g[G::RS + lane] = rs;
        }
        if (rows_here) {
            const int nrow = (M + WAVE - 1) / WAVE;              // uniform, <= 8 (Mp <= 512)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if (role_gather_acc_wave(kk) != wv || kk >= nrow) continue;
                const int mm = lane + kk * WAVE;
                const bool on = mm < M;
                const uint4 *rp[NC];
                double x[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) rp[j] = R + (size_t)(j < ntc ? j : 0) * d.Mp + (on ? mm : 0);
                ll_poll_many<NC, LL_RPRIO>(rp, ll.seq_in, ll.late, x);
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NC; ++j) acc += (on && j < ntc) ? x[j] : 0.0;
                g[G::ACC + kk * WAVE + lane] = acc;
            }
        }
    }
}

// PERS (k_leap: all inner steps in ONE launch): what the previous step's roles wrote -- position and momentum, the
// global parameters, the chunk sums, the spatial effects -- was written by workgroups of this same launch too, so those
// are read past the L1 as well (the caller makes sure every role of the previous step has finished).
// TRAJ: the role also knows the trajectory's end points (`traj`, below): always in k_leap, and in k_se_chunk, whose launches
// can carry the first and the last step of a trajectory as well (the stage kernels' work in chunk form: hmc_mode 6).
template <int NTC, bool COH, bool PERS = false, bool TRAJ = PERS, typename Wait>
__device__ __forceinline__ void hmc_chunk_role(const Dims &d, const Consts &c, const Work &w, const SamplerCfg &s,
                                               const Chains &ch, int par, int bx, int b, Wait wait, int lane_in = -1,
                                               unsigned long long *probe = nullptr, double *gbuf = nullptr, int traj = 0,
                                               double eps_in = 0.0, bool tab_ready = false, const LeapLL &ll = LeapLL{}) {
#ifdef LEAP_STAMPS
#define CPROBE(k) do { asm volatile("s_nop 0" ::: "memory"); if (probe && threadIdx.x == 0) probe[(k) < 8 ? (k) : 2 * 128 + (k) - 8] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CPROBE(k) do {} while (0)
#endif
    __shared__ double2 ltab[LOGTAB_N];
    auto LDP = [](const double *p_) {
        return COH ? __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p_;
    };
    auto LDQ = [](const double *p_) {
        return PERS ? __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p_;
    };
    constexpr int NC = NTC > 0 ? NTC : CT_MAXC;
    const int T = d.T, M = d.M, nmt = d.nmt;
    const int ntc = NTC > 0 ? NTC : d.ntc;
    const int ntile = nmt * ntc;
    // (k_leap passes the lane id through an opaque move made inside its step loop: every per-lane address then depends
    // on something defined in the loop, and the compiler cannot hoist -- and then spill -- a few dozen 64-bit addresses)
    const int lane = PERS ? lane_in : (int)threadIdx.x;
    double *q = ch.q + (size_t)b * d.Pp, *p = ch.p + (size_t)b * d.Pp;
    const double *qs0 = ch.q0 + (size_t)b * d.Pp;          // the trajectory's start point (== q before the first step; nobody writes it here)
    const double *var = ch.var + (size_t)b * d.Pp;
    double *sc = w.scal + (size_t)b * NSCAL;
    // (k_leap reads the step size once, ahead of its step loop, and fills the logarithm's table once: at the top of a role
    // each was a memory round trip of its own -- the table's two halves one after the other -- before the role's first useful
    // load was issued, about a microsecond of every step)
    const double eps = PERS ? eps_in : ch.hs[(size_t)b * NHS + HS_EPS];
    // traj (k_leap with the trajectory's end points folded in): 1 = this is the FIRST step of the trajectory -- what
    // k_hmc_step<0> does in the multi-launch forms: the momentum is drawn here (same Philox slots), the kick is half a step
    // and the kinetic energy's / log-probability's parts are left for the accept test.  Everything of the start point is
    // read from Chains::q0 (and psi, sigma, the sigmoids formed again from it): the other roles of this step are already
    // writing the next position to q and the scalar block.  2 = the step after it, whose first T-chunk adds the parts up
    // (every role of step one has finished)
    // 3 = the trajectory's LAST half kick (k_hmc_step<2>'s in the other forms): nothing of the position moves; the role leaves
    // its parts of the end point's kinetic energy and log-probability (Chains::finpart) for the accept test, which every
    // role makes for itself once all of them have counted in (hmc_final_apply)
    const bool first = TRAJ && traj == 1, fin = TRAJ && traj == 3;
    const double kick = (first || fin) ? 0.5 * eps : eps;
    const int oT = 6 - 1, oM = 6 + T - 1;
    const double *TS = w.TS + (size_t)b * ntile * 4;
    const double *gr = w.gst + ((size_t)b * 2 + par) * GST_N;
    double *gw = w.gst + ((size_t)b * 2 + (par ^ 1)) * GST_N;
    if (bx < ntc) {
        // ------------------------------------------------------------------ T-chunk
        const int ci = bx, t = ci * WAVE + lane;
        const bool own = t >= 1 && t < T;
        if (!(PERS && tab_ready)) {
            const double2 t0_ = c.logtab[lane], t1_ = c.logtab[lane + WAVE];
            ltab[lane] = t0_;
            ltab[lane + WAVE] = t1_;
        }
        const double alpha = own ? LDQ((first ? qs0 : q) + oT + t) : 0.0, v = own ? var[oT + t] : 0.0;
        const double wd_t = c.wd[t];
        const double va0 = var[5], vg0 = var[3], vg1 = var[4];
        double pm, a0, g0, g1, pa0, pg0, pg1;
        RngKey key{};
        if (first) {
            key = rng_key(s, ch, b);
            pm = own ? momentum_normal(key, oT + t) / sqrt(v) : 0.0;
            a0 = qs0[5]; g0 = qs0[3]; g1 = qs0[4];
            pa0 = momentum_normal(key, 5) / sqrt(va0); pg0 = momentum_normal(key, 3) / sqrt(vg0); pg1 = momentum_normal(key, 4) / sqrt(vg1);
        } else {
            pm = own ? LDQ(p + oT + t) : 0.0;
            a0 = LDQ(gr + 5); g0 = LDQ(gr + 3); g1 = LDQ(gr + 4); pa0 = LDQ(gr + 11); pg0 = LDQ(gr + 9); pg1 = LDQ(gr + 10);
        }
        // V at the chunk ends, the chunk sums of the current position (chunk = lane)
        double vend[NC];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) vend[cc] = cc < ntc ? w.Vt[(size_t)b * d.Tp + cc * WAVE + WAVE - 1] : 0.0;
        const double *ctr = w.CT + (((size_t)b * 2 + par) * CT_MAXC) * 4;
        double cta_l = 0.0, ctvp_l = 0.0, ctva_l = 0.0, cg0_l = 0.0, cg1_l = 0.0;
        // I->R gradient (gamma0, gamma1): the chunks' parts, left by the previous step (Work::CG) -- ntc pairs instead
        // of one series evaluation per day of the whole series in every chunk
        const double *cgr = w.CG + (((size_t)b * 2 + par) * CT_MAXC) * 2;
        const double kir_t = t < T ? w.Kir[(size_t)b * d.Tp + t] : 0.0, dir_t = t < T ? w.Dir[(size_t)b * d.Tp + t] : 0.0;
        if (!first) {
            if (lane < ntc) { cta_l = LDQ(ctr + lane * 4); ctvp_l = LDQ(ctr + lane * 4 + 1); ctva_l = LDQ(ctr + lane * 4 + 2); }
            if (lane < ntc) { cg0_l = LDQ(cgr + lane * 2); cg1_l = LDQ(cgr + lane * 2 + 1); }
        }
        // The I->R part of the step needs nothing of this step's tiles -- gamma0 and gamma1 move by the chunk parts the
        // previous step left (Work::CG) -- so all of it (two wave sums, the new rates, the series, two more wave sums: half of
        // the role's dependent operations) runs BEFORE the wait, under the tile phase.  Same operations, same results.
        CPROBE(10);                                         // entry loads issued
        if (PERS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the workgroup's other waves are alive there: no s_barrier)
        else lds_barrier();                                // ltab (single wave: orders the LDS writes)
        if (first) {
            // nobody left the chunk sums of the start point: this wave forms them -- of the chunks before its own (alpha, v p,
            // v alpha: the momenta of those chunks are drawn again here, Philox is a counter) and, for all chunks, the parts of
            // the I->R term's gradient and the term itself (the start point's log-probability, for the accept test).  All of
            // it before the wait for the tiles.
            for (int cc = 0; cc < ci; ++cc) {
                const int tt = cc * WAVE + lane;
                const bool on = tt >= 1 && tt < T;
                const double al = on ? qs0[oT + tt] : 0.0, vv = on ? var[oT + tt] : 0.0;
                const double pp = on ? momentum_normal(key, oT + tt) / sqrt(vv) : 0.0;
                const double sa = wave_sum(al), svp = wave_sum(vv * pp), sva = wave_sum(vv * al);
                if (lane == cc) { cta_l = sa; ctvp_l = svp; ctva_l = sva; }
            }
            double irl = 0.0;
            for (int cc = 0; cc < ntc; ++cc) {
                const int tt = cc * WAVE + lane;
                double x0 = 0.0, x1 = 0.0, xl = 0.0;
                if (tt < T) {
                    const double rr_ = exp(g0 + g1 * c.wd[tt]);
                    const double kk_ = w.Kir[(size_t)b * d.Tp + tt], dd_ = w.Dir[(size_t)b * d.Tp + tt];
                    double L, inv;
                    l1me_inv_wide(rr_ * d.dt, L, inv, ltab);
                    const double grr = d.dt * ((kk_ != 0.0 ? kk_ * inv : 0.0) - dd_);
                    x0 = grr * rr_; x1 = grr * rr_ * c.wd[tt];
                    xl = (kk_ != 0.0 ? kk_ * L : 0.0) - dd_ * (rr_ * d.dt);
                }
                const double s0_ = wave_sum(x0), s1_ = wave_sum(x1);
                irl += wave_sum(xl);
                if (lane == cc) { cg0_l = s0_; cg1_l = s1_; }
            }
            if (ci == 0 && lane == 0) ch.irl0[b] = irl;
        }
        if (TRAJ && traj == 2 && ci == 0) {
            // the step after the first: the start point's kinetic energy and log-probability from the parts the first step's
            // roles and tiles left (all of them have finished: the caller waited for that)
            const int nroles = ntc + d.Mp / WAVE;
            const double k0 = wave_sum(lane < nroles ? LDQ(ch.k0part + (size_t)b * ROLE_SLOTS + lane) : 0.0);
            double lk = 0.0;
            for (int i = lane; i < ntile; i += WAVE) lk += LDP(w.Lpart0 + (size_t)b * ntile + i);
            lk = wave_sum(lk);
            if (lane == 0) {
                double *hs = ch.hs + (size_t)b * NHS;
                hs[HS_LP0] = (lk + LDQ(ch.irl0 + b)) + sc[SC_PRIOR] + sc[SC_JAC];
                hs[HS_K0] = k0;
            }
        }
        const double gg0 = wave_sum(cg0_l), gg1 = wave_sum(cg1_l);
        if (probe) { asm volatile("" :: "v"(gg0), "v"(gg1)); }
        CPROBE(11);                                         // entry loads back, two wave sums
        const double pg0n = pg0 + kick * (gg0 - g0 / 1.0e4), g0n = g0 + eps * vg0 * pg0n;
        const double pg1n = pg1 + kick * (gg1 - g1 / 1.0e4), g1n = g1 + eps * vg1 * pg1n;
        double ng0p = 0.0, ng1p = 0.0, rnew = 0.0;
        if (t < T) {
            rnew = exp(g0n + g1n * wd_t);
            // this chunk's part of the I->R gradient at the new rates, for the next step
            double L, inv;
            l1me_inv_wide(rnew * d.dt, L, inv, ltab);
            const double grr = d.dt * ((kir_t != 0.0 ? kir_t * inv : 0.0) - dir_t);
            ng0p = grr * rnew;
            ng1p = grr * rnew * wd_t;
        }
        const double ng0s = wave_sum(ng0p), ng1s = wave_sum(ng1p);
        if (probe) { asm volatile("" :: "v"(ng0s), "v"(ng1s)); }
        double irl_fin = 0.0;
        if (fin) {
            // the I->R term itself at the end point's rates (the accept test's log-probability), this chunk's days
            double xl = 0.0;
            if (t < T) {
                const double rr_ = exp(g0 + g1 * wd_t);
                double L, inv;
                l1me_inv_wide(rr_ * d.dt, L, inv, ltab);
                xl = (kir_t != 0.0 ? kir_t * L : 0.0) - dir_t * (rr_ * d.dt);
            }
            irl_fin = wave_sum(xl);
        }
        CPROBE(12);                                         // I->R part done
        wait();
        // ---- from here on: this step's partial sums
        double col = 0.0;
        double bs[NC], as[NC];
        if (PERS) {
            // the four waves of the workgroup fetch a quarter each (role_gather); wave 0 is this one
            using G = RoleGather<NC>;
            role_gather<NTC>(d, ch, ll, b, bx, 0, lane, gbuf);
            lds_barrier();
            col = (gbuf[G::C + lane] + gbuf[G::C + WAVE + lane]) + (gbuf[G::C + 2 * WAVE + lane] + gbuf[G::C + 3 * WAVE + lane]);
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) { bs[cc] = gbuf[G::BS + cc * WAVE + lane]; as[cc] = gbuf[G::AS + cc * WAVE + lane]; }
            if (probe) { asm volatile("" :: "v"(col)); }
            CPROBE(8);                                          // column sums in
        } else {
        // column sums of this chunk
        {
            const double *kp = w.Kpart + (size_t)b * nmt * d.Tp + t;
            double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
            // twelve loads in flight at a time (twenty-four cost 24 more registers, and the kernel that carries this role
            // next to the gradient tile has to stay at five waves per SIMD); same order of additions as one batch of 24
            constexpr int KB = 12;
            for (int j0 = 0; j0 < nmt; j0 += KB) {
                double x[KB];
#pragma unroll
                for (int j = 0; j < KB; ++j) {             // clamped index + select: a conditional L1-bypassing load is a branch
                    const double v_ = LDP(kp + (size_t)min(j0 + j, nmt - 1) * d.Tp);
                    x[j] = j0 + j < nmt ? v_ : 0.0;
                }
#pragma unroll
                for (int j = 0; j < KB; j += 4) { c0 += x[j]; c1 += x[j + 1]; c2 += x[j + 2]; c3 += x[j + 3]; }
            }
            col = (c0 + c1) + (c2 + c3);
        }
        if (probe) { asm volatile("" :: "v"(col)); }
        CPROBE(8);                                          // column sums in
        // tile scalars of row tile `lane` (and lane+64, ... when there are more)
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const bool on = cc < ntc && lane < nmt;
            const double *tp_ = TS + ((size_t)min(lane, nmt - 1) * ntc + min(cc, ntc - 1)) * 4;
            const double b_ = LDP(tp_), a_ = LDP(tp_ + 1);
            bs[cc] = on ? b_ : 0.0;
            as[cc] = on ? a_ : 0.0;
        }
        // more than 64 row tiles (M > 1024): three chunks' scalars in flight at a time -- all of them at once were the
        // register peak of the whole role (24 more VGPRs), paid by every problem size
        for (int r = lane + WAVE; r < nmt; r += WAVE) {
#pragma unroll
            for (int c0 = 0; c0 < NC; c0 += 3) {
#pragma unroll
                for (int cc = c0; cc < c0 + 3 && cc < NC; ++cc)
                    if (cc < ntc) { bs[cc] += LDP(TS + ((size_t)r * ntc + cc) * 4); as[cc] += LDP(TS + ((size_t)r * ntc + cc) * 4 + 1); }
#pragma unroll
                for (int cc = c0; cc < c0 + 3 && cc < NC; ++cc) asm volatile("" : "+v"(bs[cc]), "+v"(as[cc]));   // the sums, here
            }
        }
        }
        // Everything that couples the chunks is linear in the tile scalars, so each lane forms its
        // row tile's share and three wave sums finish the job:
        //   later = sum_{c' > ci} B(c'),  allB = sum B,
        //   pre   = sum_{c' < ci} [new alpha summed over chunk c']
        //         = sum_{c' < ci} { CTa + eps (CTvp + eps [ (A - V0 B) + Vtot sum_{c''>c'} B - CTva / 0.005^2 ]) }
        constexpr double PREC = 1.0 / (0.005 * 0.005);
        double p_later = 0.0, p_all = 0.0, p_pre = 0.0, lat = 0.0;
#pragma unroll
        for (int cc = NC - 1; cc >= 0; --cc) {
            if (cc >= ntc) continue;
            const double V0 = cc > 0 ? vend[cc > 0 ? cc - 1 : 0] : 0.0, V1 = vend[cc];
            if (cc < ci) p_pre += (as[cc] - V0 * bs[cc]) + (V1 - V0) * lat;
            if (cc > ci) p_later += bs[cc];
            p_all += bs[cc];
            lat += bs[cc];
        }
        p_pre = eps * kick * p_pre + (lane < ci ? cta_l + eps * (ctvp_l - kick * PREC * ctva_l) : 0.0);
        if (probe) { asm volatile("" :: "v"(col), "v"(p_pre)); }
        CPROBE(5);                                          // loads back
        const double later = wave_sum(p_later), allB = wave_sum(p_all), pre = wave_sum(p_pre);
        if (probe) { asm volatile("" :: "v"(later), "v"(allB), "v"(pre)); }
        CPROBE(6);                                          // three wave sums
        // this chunk's entries
        const double insuf = wave_incl_suffix_scan(col, lane);
        const double g = own ? (insuf + later) - alpha * PREC : 0.0;
        const double pn = own ? pm + kick * g : 0.0;
        const double an = own ? alpha + eps * v * pn : 0.0;
        const double pa0n = pa0 + kick * (allB - a0 / 100.0), a0n = a0 + eps * va0 * pa0n;
        if (fin) {
            double kin = wave_sum(own ? 0.5 * v * pn * pn : 0.0);
            if (ci == 0) kin += (0.5 * vg0 * pg0n * pg0n + 0.5 * vg1 * pg1n * pg1n) + 0.5 * va0 * pa0n * pa0n;
            const double a2 = wave_sum(alpha * alpha);
            if (lane == 0) {
                double *fpw = ch.finpart + ((size_t)b * ROLE_SLOTS + bx) * 4;
                fpw[0] = kin; fpw[1] = irl_fin; fpw[2] = a2;
                if (ci == 0) fpw[3] = ch.hs[(size_t)b * NHS + HS_RV_N];      // (read by every role behind the hand-off; advanced by this one)
            }
            return;
        }
        if (first) {
            double kin = wave_sum(own ? 0.5 * v * pm * pm : 0.0);
            if (ci == 0) kin += (0.5 * vg0 * pg0 * pg0 + 0.5 * vg1 * pg1 * pg1) + 0.5 * va0 * pa0 * pa0;
            if (lane == 0) ch.k0part[(size_t)b * ROLE_SLOTS + bx] = kin;
        }
        const double a_new = a0n + pre + wave_incl_scan(an, lane);
        if (probe) { asm volatile("" :: "v"(a_new)); }
        CPROBE(7);                                          // two scans
        // (the chunk sums for the next step before the exponential: independent of it, so the two chains overlap)
        const double ca = wave_sum(an), cvp = wave_sum(v * pn), cva = wave_sum(v * an);
        if (own) { q[oT + t] = an; p[oT + t] = pn; }
        if (t < T) {
            const double ea_new = exp(a_new);
            w.acur[(size_t)b * d.Tp + t] = a_new;
            w.ea[(size_t)b * d.Tp + t] = ea_new;
            if (PERS) ll_store(ch.llT + (size_t)b * ll_tab_len(d) + t, ea_new, ll.seq_out);   // what the tiles of the next step wait for
            w.rir[(size_t)b * d.Tp + t] = rnew;            // read by nobody in this launch
        }
        if (lane == 0) {
            double *ctw = w.CT + (((size_t)b * 2 + (par ^ 1)) * CT_MAXC + ci) * 4;
            ctw[0] = ca; ctw[1] = cvp; ctw[2] = cva;
            double *cgw = w.CG + (((size_t)b * 2 + (par ^ 1)) * CT_MAXC + ci) * 2;
            cgw[0] = ng0s; cgw[1] = ng1s;
            if (ci == 0) {
                q[3] = g0n; q[4] = g1n; q[5] = a0n; p[3] = pg0n; p[4] = pg1n; p[5] = pa0n;
                gw[3] = g0n; gw[4] = g1n; gw[5] = a0n; gw[9] = pg0n; gw[10] = pg1n; gw[11] = pa0n;
                sc[SC_G0] = g0n; sc[SC_G1] = g1n; sc[SC_A0] = a0n;
            }
        }
    } else {
        // ------------------------------------------------------------------ M-chunk
        const int ci = bx - ntc, m = ci * WAVE + lane;
        const bool own = m < M;
        // the logarithm's table, for the softplus of the new psi and sigma_space (softplus_tab): in LDS before the wait
        if (!(PERS && tab_ready)) {
            const double2 t0_ = c.logtab[lane], t1_ = c.logtab[lane + WAVE];
            ltab[lane] = t0_;
            ltab[lane + WAVE] = t1_;
        }
        // (first step of a folded trajectory: the spatial effects of the start point are q's own)
        const double *spr = first ? qs0 + oM : w.sp + ((size_t)b * 2 + par) * d.Mp;
        double *spw = w.sp + ((size_t)b * 2 + (par ^ 1)) * d.Mp;
        const double sm = own ? LDQ((first ? qs0 : q) + oM + m) : 0.0, v = own ? var[oM + m] : 0.0;
        const double lm = own ? c.la[m] : 0.0, inN = own ? c.invN[m] : 0.0;
        const double v0 = var[0], v1 = var[1], v2 = var[2];
        double pm, u0, u1, beta, p0, p1, p2, psi, sig, s0, s1;
        if (first) {
            const RngKey key = rng_key(s, ch, b);
            pm = own ? momentum_normal(key, oM + m) / sqrt(v) : 0.0;
            u0 = qs0[0]; u1 = qs0[1]; beta = qs0[2];
            p0 = momentum_normal(key, 0) / sqrt(v0); p1 = momentum_normal(key, 1) / sqrt(v1); p2 = momentum_normal(key, 2) / sqrt(v2);
            // psi, sigma and the two sigmoids of the start point, as every kernel that leaves them in the scalar block forms them
            // (that block is being rewritten by the first M-chunk of this very step)
            const double e0_ = 2.220446049250313e-16;
            const double ux_ = lane == 0 ? u0 : u1;
            double sgx_;
            const double spx_ = softplus_sigmoid_tab(ux_, ltab, sgx_);
            psi = lane_value(spx_, 0) + e0_; sig = lane_value(spx_, 1) + e0_;
            s0 = lane_value(sgx_, 0); s1 = lane_value(sgx_, 1);
        } else {
            pm = own ? LDQ(p + oM + m) : 0.0;
            u0 = LDQ(gr + 0); u1 = LDQ(gr + 1); beta = LDQ(gr + 2); p0 = LDQ(gr + 6); p1 = LDQ(gr + 7); p2 = LDQ(gr + 8);
            psi = LDQ(gr + 12); sig = LDQ(gr + 13); s0 = LDQ(gr + 14); s1 = LDQ(gr + 15);
        }
        CPROBE(5);                                          // M-chunk: entry loads issued
        double Qs = 0.0;                                   // (Q s)_m at the current position
        const bool qs_lds = PERS && role_qs_by_helper(c);  // (k_leap: by a helper wave of the role's workgroup, through LDS)
        if (own && !qs_lds) {
            if (c.qw > 0 && c.qw <= 8) {
                double qv[8]; int qc[8];
                // (clamped index + select: sixteen loads in flight; as conditional loads they were eight branches with a wait each)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool on = j < c.qw;
                    const size_t qi = (size_t)(on ? j : 0) * d.Mp + m;
                    const double qv_ = c.Qell_val[qi];
                    const int qc_ = c.Qell_col[qi];
                    qv[j] = on ? qv_ : 0.0;
                    qc[j] = on ? qc_ : 0;
                }
                double sv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) sv[j] = LDQ(spr + qc[j]);
#pragma unroll
                for (int j = 0; j < 8; ++j) Qs += qv[j] * sv[j];
            } else {
                for (int e = c.Qrow[m]; e < c.Qrow[m + 1]; ++e) Qs += c.Qval[e] * LDQ(spr + c.Qcol[e]);
            }
        }
        if (probe) { asm volatile("" :: "v"(Qs)); }
        CPROBE(6);                                          // M-chunk: (Q s) formed
        const bool rows_here = d.chunked == 1;             // small M: sum_m l_m R_m, sum_m s_m R_m from the row partials
        constexpr int RPL = 8;                             // Mp <= 512: rows lane, lane+64, ...
        // (k_leap: the rows' l_m and s_m -- the previous step's -- before the wait; what waits for the tiles comes from the
        // workgroup's four waves through LDS, role_gather)
        double lxp[PERS ? RPL : 1], sxp[PERS ? RPL : 1];
        if (PERS && rows_here) {
#pragma unroll
            for (int kk = 0; kk < RPL; ++kk) {
                const int mm = lane + kk * WAVE;
                const bool on = mm < M;
                const int mc = on ? mm : 0;
                const double lv_ = c.la[mc];
                lxp[kk] = on ? lv_ : 0.0;
                sxp[kk] = 0.0;                              // (from the helper waves, through LDS: role_pregather)
            }
        }
        CPROBE(7);                                          // M-chunk: at the wait
        wait();
        // ---- from here on: this step's partial sums
        double R = 0.0;
        double ps = 0.0, rl = 0.0, rs = 0.0;
        if (PERS) {
            using G = RoleGather<NC>;
            role_gather<NTC>(d, ch, ll, b, bx, 0, lane, gbuf);
            lds_barrier();
#pragma unroll
            for (int j = 0; j < NC; ++j) R += gbuf[G::X + j * WAVE + lane];
            ps = gbuf[G::PS + lane]; rl = gbuf[G::RL + lane]; rs = gbuf[G::RS + lane];
            if (qs_lds) Qs = gbuf[G::QS + lane];
            if (rows_here) {
                const int nrow = (M + WAVE - 1) / WAVE;      // uniform
#pragma unroll
                for (int kk = 0; kk < RPL; ++kk)
                    if (kk < nrow) {
                        const double acc = gbuf[G::ACC + kk * WAVE + lane];
                        sxp[kk] = gbuf[G::SX + kk * WAVE + lane];
                        rl = fma(lxp[kk], acc, rl); rs = fma(sxp[kk], acc, rs);
                    }
            }
        } else {
        {
            const double *rp = w.Rpart + (size_t)b * ntc * d.Mp + (own ? m : 0);
            double x[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) x[j] = (own && j < ntc) ? LDP(rp + (size_t)j * d.Mp) : 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) R += x[j];
        }
        for (int i0 = lane; i0 < ntile; i0 += 4 * WAVE) {
            double x[4], y[4], z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * WAVE;
                const bool on = i < ntile;
                x[j] = on ? LDP(w.Ppart + (size_t)b * ntile + i) : 0.0;
                y[j] = (on && !rows_here) ? LDP(TS + (size_t)i * 4 + 2) : 0.0;
                z[j] = (on && !rows_here) ? LDP(TS + (size_t)i * 4 + 3) : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { ps += x[j]; rl += y[j]; rs += z[j]; }
        }
        if (rows_here) {
            // ntc partials per row, and the row's l_m and s_m with them: the same round trip, and nothing held in
            // registers across the wait (sixteen values per lane were: with them this role needed 113 VGPRs, and the
            // gradient launch that carries it, k_se_chunk, stays at five waves per SIMD only up to 96)
            const int nrow = (M + WAVE - 1) / WAVE;          // uniform
            constexpr int RB = 3;
            for (int k0 = 0; k0 < RPL; k0 += RB) {
                if (k0 >= nrow) break;
                double lx[RB], sx[RB], x[RB][NC];
#pragma unroll
                for (int kk = 0; kk < RB; ++kk) {
                    const int mm = lane + (k0 + kk) * WAVE;
                    const bool on = k0 + kk < RPL && mm < M;
                    const int mc = on ? mm : 0;
                    const double *rp = w.Rpart + (size_t)b * ntc * d.Mp + mc;
                    lx[kk] = on ? c.la[mc] : 0.0;
                    const double sv_ = LDQ(spr + mc);              // (clamped index + select: no branch around the load)
                    sx[kk] = on ? sv_ : 0.0;
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        const double v_ = LDP(rp + (size_t)(j < ntc ? j : 0) * d.Mp);
                        x[kk][j] = (on && j < ntc) ? v_ : 0.0;
                    }
                }
#pragma unroll
                for (int kk = 0; kk < RB; ++kk) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NC; ++j) acc += x[kk][j];
                    rl = fma(lx[kk], acc, rl); rs = fma(sx[kk], acc, rs);
                }
            }
        }
        }
        if (probe) { asm volatile("" :: "v"(ps), "v"(rl), "v"(rs), "v"(R)); }
        CPROBE(8);                                          // M-chunk: the partial sums are in
        ps = wave_sum(ps); rl = wave_sum(rl); rs = wave_sum(rs);
        if (probe) { asm volatile("" :: "v"(ps), "v"(rl), "v"(rs)); }
        CPROBE(10);                                         // M-chunk: three wave sums
        const double g = own ? sig * R - Qs : 0.0;
        const double pn = pm + kick * g;
        const double sn = sm + eps * v * pn;
        const double p0n = p0 + kick * ((ps + 2.0 / psi - 10.0) * s0 + (1.0 - s0)), u0n = u0 + eps * v0 * p0n;
        const double p1n = p1 + kick * ((rs - sig / 0.01) * s1 + (1.0 - s1)), u1n = u1 + eps * v1 * p1n;
        const double p2n = p2 + kick * (rl - beta), betan = beta + eps * v2 * p2n;
        if (fin) {
            double kin = wave_sum(own ? 0.5 * v * pn * pn : 0.0);
            if (ci == 0) kin += (0.5 * v0 * p0n * p0n + 0.5 * v1 * p1n * p1n) + 0.5 * v2 * p2n * p2n;
            const double sqs = wave_sum(own ? sm * Qs : 0.0);
            if (lane == 0) {
                double *fpw = ch.finpart + ((size_t)b * ROLE_SLOTS + bx) * 4;
                fpw[0] = kin; fpw[1] = 0.0; fpw[2] = sqs;
            }
            return;
        }
        if (first) {
            double kin = wave_sum(own ? 0.5 * v * pm * pm : 0.0);
            if (ci == 0) kin += (0.5 * v0 * p0 * p0 + 0.5 * v1 * p1 * p1) + 0.5 * v2 * p2 * p2;
            if (lane == 0) ch.k0part[(size_t)b * ROLE_SLOTS + bx] = kin;
        }
        // both softplus in one pass: lane 0 takes u0, the other lanes u1
        const double e0 = 2.220446049250313e-16;
        const double ux = lane == 0 ? u0n : u1n;
        double sgx;                                        // sigmoid(u) from the same exp(-|u|) as the softplus: one exponential, not two in series
        const double spx = softplus_sigmoid_tab(ux, ltab, sgx);
        const double psin = lane_value(spx, 0) + e0, sign = lane_value(spx, 1) + e0;
        const double s0n = lane_value(sgx, 0), s1n = lane_value(sgx, 1);
        if (probe) { asm volatile("" :: "v"(psin), "v"(sign), "v"(s0n), "v"(s1n)); }
        CPROBE(11);                                         // M-chunk: softplus and sigmoid of the new psi, sigma
        if (own) {
            const double eb_new = exp(betan * lm + sign * sn) * inN;
            q[oM + m] = sn; p[oM + m] = pn;
            spw[m] = sn;
            w.eb[(size_t)b * d.Mp + m] = eb_new;
            if (PERS) {                                     // the tiles of the next step wait for these
                uint4 *lt = ch.llT + (size_t)b * ll_tab_len(d) + d.Tp;
                ll_store(lt + m, eb_new, ll.seq_out);
                ll_store(lt + d.Mp + m, sn, ll.seq_out);
            }
        }
        if (PERS && ci == 0 && lane == 0) ll_store(ch.llT + (size_t)b * ll_tab_len(d) + d.Tp + 2 * (size_t)d.Mp, psin, ll.seq_out);
        if (ci == 0 && lane == 0) {
            q[0] = u0n; q[1] = u1n; q[2] = betan; p[0] = p0n; p[1] = p1n; p[2] = p2n;
            gw[0] = u0n; gw[1] = u1n; gw[2] = betan; gw[6] = p0n; gw[7] = p1n; gw[8] = p2n;
            gw[12] = psin; gw[13] = sign; gw[14] = s0n; gw[15] = s1n;
            sc[SC_PSI] = psin; sc[SC_SIG] = sign; sc[SC_BETA] = betan; sc[SC_S0] = s0n; sc[SC_S1] = s1n;
        }
    }
}


// The end of a trajectory inside k_leap (fold bit 2): what k_hmc_step<2> does after its half kick, by the chunk roles and
// without another hand-off.  Every role of the last step has left its parts (Chains::finpart) and counted in; each role --
// one wave -- now adds them up in the same order, so all of them arrive at the same log-ratio and, from the same Philox
// draw, at the same decision, and each applies it to the entries it owns: on acceptance q stays (and becomes q0, the next
// trajectory's start point; the tables are the end point's already), on rejection q goes back to q0 and the role
// rebuilds its share of the tables from it.  The first T-chunk also does the bookkeeping of the step (accept flag, log-probability,
// dual averaging, trace row), T-chunk 0 / M-chunk 0 the global parameters.  Nothing a role writes here is read by another
// role of this launch: they read finpart, the tiles' parts, the globals' block of the step (Work::gst), q0 and the start
// point's energy, none of which is written.
// (Sums are associated differently from k_hmc_step<2>'s block reductions: same draws up to rounding.)
template <int NTC>
__device__ __forceinline__ void hmc_final_apply(const Dims &d, const Consts &c, const Work &w, const SamplerCfg &s,
                                                const Chains &ch, int par, int bx, int b, int lane) {
    auto LDQ = [](const double *p_) { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const int T = d.T, M = d.M, ntc = NTC > 0 ? NTC : d.ntc, nroles = ntc + d.Mp / WAVE, ntile = d.nmt * ntc;
    const int oT = 6 - 1, oM = 6 + T - 1;
    double *q = ch.q + (size_t)b * d.Pp, *q0 = ch.q0 + (size_t)b * d.Pp, *var = ch.var + (size_t)b * d.Pp;
    double *hs = ch.hs + (size_t)b * NHS, *sc = w.scal + (size_t)b * NSCAL;
    const double *gr = w.gst + ((size_t)b * 2 + par) * GST_N;
    const double *fp = ch.finpart + (size_t)b * ROLE_SLOTS * 4;
    const double e0 = 2.220446049250313e-16;
    // ---- the sums: the same in every role
    const bool isr = lane < nroles, ist = lane < ntc;
    const double f0 = LDQ(fp + (isr ? lane : 0) * 4), f1 = LDQ(fp + (isr ? lane : 0) * 4 + 1), f2 = LDQ(fp + (isr ? lane : 0) * 4 + 2);
    const double n_old = LDQ(fp + 3);
    double lk = 0.0;
    for (int i = lane; i < ntile; i += WAVE) lk += LDQ(w.Lpart + (size_t)b * ntile + i);
    const double u0 = LDQ(gr + 0), u1 = LDQ(gr + 1), beta = LDQ(gr + 2), g0 = LDQ(gr + 3), g1 = LDQ(gr + 4), a0 = LDQ(gr + 5),
                 psi = LDQ(gr + 12), sig = LDQ(gr + 13);
    const double lp0 = LDQ(hs + HS_LP0), k0 = LDQ(hs + HS_K0);
    const double K1 = wave_sum(isr ? f0 : 0.0), irl = wave_sum(ist ? f1 : 0.0), a2 = wave_sum(ist ? f2 : 0.0),
                 sqs = wave_sum((isr && !ist) ? f2 : 0.0);
    lk = wave_sum(lk);
    // model_spec.py:140-198 and the bijector's Jacobian (inference.py:555-557), as k_hmc_step writes them
    double prior = d.prior_const;
    prior += -0.5 * a0 * a0 / 100.0 - 0.5 * beta * beta;
    prior += 2.0 * cold_log(psi) - 10.0 * psi;
    prior += -0.5 * a2 / (0.005 * 0.005);
    prior += -sig * sig / 0.02;
    prior += -0.5 * sqs;
    prior += -0.5 * (g0 * g0 + g1 * g1) / 1.0e4;
    const double jac = (u0 - (psi - e0)) + (u1 - (sig - e0));
    const double lp_theta = (lk + irl) + prior + jac;
    const double log_ratio = (lp_theta - lp0) - (K1 - k0);
    const RngKey key = rng_key(s, ch, b);
    double ua, ub;
    rng_uniform2(key, RS_HMC_ACCEPT, 0u, ua, ub);
    const bool acc = cold_log(ua) < log_ratio && !(s.disable_mask & 1);            // NaN compares false -> reject
    const unsigned slot = ch.sweep[b] - ch.slot0[0];
    double *tr = slot < (unsigned)s.cap ? ch.tr_theta + ((size_t)slot * s.B + b) * d.P : nullptr;
    const double n1 = n_old + 1.0;
    auto welford = [&](int i, double x) {                      // running variance with the new state (ddof 0)
        double *mean = ch.rv_mean + (size_t)b * d.Pp, *m2 = ch.rv_m2 + (size_t)b * d.Pp;
        const double dlt = x - mean[i];
        const double mu = mean[i] + dlt / n1;
        const double ss = m2[i] + dlt * (x - mu);
        mean[i] = mu; m2[i] = ss;
        var[i] = ss / n1;
    };
    if (bx < ntc) {
        // ------------------------------------------------------------------ T-chunk
        const int ci = bx, t = ci * WAVE + lane;
        const bool own = t >= 1 && t < T;
        const double xq = own ? (acc ? LDQ(q + oT + t) : q0[oT + t]) : 0.0;
        if (own) {
            if (acc) q0[oT + t] = xq; else q[oT + t] = xq;
            if (s.adapt_mass) welford(oT + t, xq);
            if (tr) tr[oT + t] = xq;
        }
        double ga0 = a0, gg0 = g0, gg1 = g1;                   // alpha_0, gamma_0, gamma_1 of the state the chain is left in
        if (!acc) {
            // back at the start point: this chunk's days of exp(a_t) and of the I->R rate from q0 (the sum of alpha over the
            // chunks before this one from their entries, as the first step of a trajectory forms it)
            ga0 = q0[5]; gg0 = q0[3]; gg1 = q0[4];
            double pre = 0.0;
            for (int cc = 0; cc < ci; ++cc) {
                const int tt = cc * WAVE + lane;
                pre += wave_sum((tt >= 1 && tt < T) ? q0[oT + tt] : 0.0);
            }
            const double a_t = ga0 + pre + wave_incl_scan(xq, lane);
            if (t < T) {
                w.acur[(size_t)b * d.Tp + t] = a_t;
                w.ea[(size_t)b * d.Tp + t] = exp(a_t);
                w.rir[(size_t)b * d.Tp + t] = exp(gg0 + gg1 * c.wd[t]);
            }
        }
        if (ci == 0 && lane == 0) {
            if (acc) { q0[3] = gg0; q0[4] = gg1; q0[5] = ga0; }
            else { q[3] = gg0; q[4] = gg1; q[5] = ga0; sc[SC_G0] = gg0; sc[SC_G1] = gg1; sc[SC_A0] = ga0; }
            if (s.adapt_mass) { welford(3, gg0); welford(4, gg1); welford(5, ga0); hs[HS_RV_N] = n1; }
            if (tr) { tr[3] = gg0; tr[4] = gg1; tr[5] = ga0; }
            if (acc) { sc[SC_PRIOR] = prior; sc[SC_JAC] = jac; }
            // the step's bookkeeping (k_hmc_step<2>)
            const double eps = hs[HS_EPS];
            hs[HS_ACC] = acc ? 1.0 : 0.0;
            hs[HS_LOGACC] = log_ratio;
            const double lpt = acc ? lp_theta : lp0;
            hs[HS_LP_THETA] = lpt;
            double eps_traced = eps;
            if (s.adapt_step) {                       // dual averaging (Hoffman & Gelman alg. 5, TFP defaults)
                const double a = isfinite(log_ratio) ? fmin(1.0, cold_exp(log_ratio)) : 0.0;
                const double prev_step = hs[HS_DA_STEP];
                const double n = prev_step + 1.0;
                const double err = hs[HS_DA_ERR] + s.target_accept - a;
                const double log_step = hs[HS_DA_MU] - err * sqrt(n) / ((n + 10.0) * 0.05);
                const double eta = cold_exp(-0.75 * cold_log(n));
                const double log_avg = eta * log_step + (1.0 - eta) * hs[HS_DA_LOGAVG];
                hs[HS_DA_ERR] = err; hs[HS_DA_STEP] = n; hs[HS_DA_LOGAVG] = log_avg;
                if (prev_step <= (double)s.n_adapt) {
                    eps_traced = cold_exp(prev_step < (double)s.n_adapt ? log_step : log_avg);
                    hs[HS_EPS] = eps_traced;
                }
            }
            if (slot < (unsigned)s.cap) {
                double *th = ch.tr_hmc + ((size_t)slot * s.B + b) * 3;
                th[0] = acc ? 1.0 : 0.0;
                th[1] = lpt + hs[HS_LP_CONST];
                th[2] = eps_traced;
            }
        }
    } else {
        // ------------------------------------------------------------------ M-chunk
        const int ci = bx - ntc, m = ci * WAVE + lane;
        const bool own = m < M;
        const double xq = own ? (acc ? LDQ(q + oM + m) : q0[oM + m]) : 0.0;
        double bu0 = u0, bu1 = u1, bbeta = beta, bpsi = psi, bsig = sig;
        if (own) {
            if (acc) q0[oM + m] = xq; else q[oM + m] = xq;
            if (s.adapt_mass) welford(oM + m, xq);
            if (tr) tr[oM + m] = xq;
        }
        if (!acc) {
            // back at the start point: psi, sigma_space and the rows' exp(b_m)/N_m from q0
            bu0 = q0[0]; bu1 = q0[1]; bbeta = q0[2];
            const double ux = lane == 0 ? bu0 : bu1;
            // (as the role that left this point formed them, so that a rejected draw repeats the previous one to the bit;
            // the table straight from memory: a rare path)
            const double spx = softplus_tab(ux, c.logtab);
            const double sgx = cold_exp(ux - spx);
            bpsi = lane_value(spx, 0) + e0; bsig = lane_value(spx, 1) + e0;
            const double s0n = lane_value(sgx, 0), s1n = lane_value(sgx, 1);
            if (own) w.eb[(size_t)b * d.Mp + m] = exp(bbeta * c.la[m] + bsig * xq) * c.invN[m];
            if (ci == 0 && lane == 0) {
                q[0] = bu0; q[1] = bu1; q[2] = bbeta;
                sc[SC_PSI] = bpsi; sc[SC_SIG] = bsig; sc[SC_BETA] = bbeta; sc[SC_S0] = s0n; sc[SC_S1] = s1n;
            }
        }
        if (ci == 0 && lane == 0) {
            if (acc) { q0[0] = bu0; q0[1] = bu1; q0[2] = bbeta; }
            if (s.adapt_mass) { welford(0, bu0); welford(1, bu1); welford(2, bbeta); }
            if (tr) { tr[0] = bpsi; tr[1] = bsig; tr[2] = bbeta; }
        }
    }
}

template <int NTC>
__global__ __launch_bounds__(WAVE) void k_hmc_chunk(Dims d, Consts c, Work w, SamplerCfg s, Chains ch, int par) {
    debug_skew(d);
    int bx = blockIdx.x, by = blockIdx.y;
    if (d.aff_nb > 0) xcd_affine(blockIdx.x, (NTC > 0 ? NTC : d.ntc) + d.Mp / WAVE, d.aff_nb, by, bx);
    hmc_chunk_role<NTC, false>(d, c, w, s, ch, par, bx, d.b0 + by, [] {});
}

// k_se_chunk: the gradient tiles of a leapfrog step AND the chunk roles of that step in one launch -- what it saves is
// the chunk kernel's launch ramp and boundary (~2.5 us, fifteen times per sweep) and, because a role issues every
// load that does not depend on the tiles while they are still running, most of its memory latency as well.
// Grid: the XCD-affine tile grid of k_se (ntile x nb blocks, nb a multiple of 8) followed by (ntc + Mp/64) x nb role
// blocks (block id mod 8 = chain mod 8 there too).  A tile workgroup evaluates its tile exactly as k_se does and then counts itself in on the chain's
// counter (Chains::tail, one cache line per chain); the tile that brings the counter to `target` -- every tile of the
// chain, of every launch so far -- raises the chain's flag (a line of its own); a role workgroup (one wave) issues its
// independent loads, waits for the flag and runs the role.  Blocks
// are dispatched in id order, so a role is placed only after every tile has been: it can never hold a slot that an
// unplaced tile needs, whatever the residency.
// Memory: the hand-off uses NO agent-scope release/acquire (an L2 write-back / invalidate costs 7-30 us here,
// tools/probes/xcd_barrier_probe.hip).  It relies on all workgroups of a chain sharing one XCD and hence one L2 --
// block ids congruent mod 8, checked at sampler creation through XCC_ID (k_xcc_probe); the host uses this kernel
// only then.  A tile's stores are acknowledged by that L2 (s_waitcnt vmcnt(0) in __syncthreads) before it counts
// itself in, and the roles read the partial sums past their L1.  The roles write the NEXT position's tables (ea,
// eb, psi ...) only after every tile of the chain has counted in, i.e. has long read the current ones.  Results are
// bit-identical to k_se followed by k_hmc_chunk.
constexpr int TAIL_BACKOFF = 30;     // x 64 cycles
// a chain's flag sits 4 KB + one line from the next chain's and well away from its counter: in the line next to the
// counters' (stride 16) the polls still cost 6 us per sweep -- presumably the same L2 channel
constexpr int TAIL_FLAG_STRIDE = 528;
#define TAIL_FLAG_AT(B_, b_) ((size_t)(B_) * TAIL_STRIDE + (size_t)(b_) * TAIL_FLAG_STRIDE)
// (No waves-per-SIMD bound here: stating the occupancy the compiler arrives at by itself -- three waves for the 12-chunk
// instances, 158 VGPRs either way -- made k_se_chunk<2,12> 16 % slower at SYN-2048, 1 518 against 1 317 us per trajectory: the
// scheduler clusters the tile's loads differently once it is given a target.  Four waves (128 VGPRs, 132 B of scratch): 1 477.)
template <int TSM, int NTC>
__global__ __launch_bounds__(256)
void k_se_chunk(Dims d, Consts c, Work w, SamplerCfg s, Chains ch, int par, unsigned long long target, int traj) {
    const int ntile = d.ntc * d.nmt, n_tiles = ntile * d.aff_nb;
    if ((int)blockIdx.x < n_tiles) {
        int bz, tile;
        xcd_affine(blockIdx.x, ntile, d.aff_nb, bz, tile);
        if (d.nlive > 0 && bz >= d.nlive) return;      // a chain of the layout that does not exist
#ifdef TAIL_STAMPS
        // developer timeline (tools/dev/tail_timeline.py), in the unused words 8..15 of the chain's own counter line (the
        // counter is word 0: the stamps disturb the hand-off they time -- good for its shape, not for its duration):
        // 8 first tile start, 9 last tile arrival,
        // 10 first role past its wait, 11 last role past its wait, 12 last role done, 13 first role start -- of the launches
        // with par = 1 since the last reset (the probe runs trajectories of three leapfrog steps: exactly one such launch)
        unsigned long long *stp = ch.tail + (size_t)(d.b0 + bz) * TAIL_STRIDE + 8;
        if (threadIdx.x == 0 && par == 1) __hip_atomic_fetch_min(stp + 0, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        se_tile<true, 1, TSM>(d, c, w, tile % d.ntc, tile / d.ntc, bz);
        __syncthreads();                               // vmcnt(0): this tile's partial sums are in the XCD's L2
        if (threadIdx.x == 0) {
            const unsigned long long old = __hip_atomic_fetch_add(ch.tail + (size_t)(d.b0 + bz) * TAIL_STRIDE, 1ull, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT);
            // the last tile of the chain raises the chain's flag, in a line of its own: the roles poll that one -- polling
            // the counter's line, which 144 tiles are still adding to, cost 11 us per sweep
            if (old + 1 == target)
                __hip_atomic_store(ch.tail + TAIL_FLAG_AT(s.B, d.b0 + bz), target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef TAIL_STAMPS
        if (threadIdx.x == 0 && par == 1) __hip_atomic_fetch_max(stp + 1, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        return;
    }
    if (threadIdx.x >= WAVE) return;                   // a role is one wave
#ifndef SE_CHUNK_ROLE_PRIO
#define SE_CHUNK_ROLE_PRIO 3
#endif
    // ... of serial code, on SIMDs it shares with the tile waves of chains that are still at their cells: ahead of them (as k_leap's)
    __builtin_amdgcn_s_setprio(SE_CHUNK_ROLE_PRIO);
    const int L = (int)blockIdx.x - n_tiles;
    const int bz = L % d.aff_nb, role = L / d.aff_nb, b = d.b0 + bz;
    if (d.nlive > 0 && bz >= d.nlive) return;
    const unsigned long long *flag = ch.tail + TAIL_FLAG_AT(s.B, b);   // raised by the chain's last tile
#ifdef TAIL_STAMPS
    unsigned long long *stp = ch.tail + (size_t)b * TAIL_STRIDE + 8;
    if (threadIdx.x == 0 && par == 1) __hip_atomic_fetch_min(stp + 5, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    // traj (hmc_mode 6: the whole trajectory as L + 1 of these launches): 1 = the trajectory's first step (the momentum is
    // drawn here, the start point read from Chains::q0: k_hmc_step<0>'s work in chunk form), 2 = the step after it, 3 = the last
    // half kick (the accept test follows in k_hmc_final); 0 = an inner step, the only kind the other forms launch
    hmc_chunk_role<NTC, true, false, true>(d, c, w, s, ch, par, role, b, [&] {
        int spins = 0;
        // back off before the first look: since the roles fit beside all the tiles (96 VGPRs, five waves per SIMD) they
        // are resident from the start of the launch, and no tile phase of this size is shorter than the ~0.9 us slept
        // through.  (While the roles still polled the counters' own lines, 96 polling waves made the launch slower than
        // at four waves per SIMD, 12.7 us against 12.1, and a 1.8 us sleep was worth 0.9 us; with the flag lines the
        // launch takes 11.0 us and the sleep is worth little.)
        if (ntile >= 32) __builtin_amdgcn_s_sleep(TAIL_BACKOFF);
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if ((spins & 255) == 0 && __hip_atomic_load(ch.late + ch.late_fatal + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // see leap_wait
            if (spins > (1 << 22)) {                   // never seen; counted like k_move_pair's time-outs, no hang
                if (threadIdx.x == 0) __hip_atomic_fetch_add(ch.late + ch.late_fatal + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
#ifdef TAIL_STAMPS
        if (threadIdx.x == 0 && par == 1) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            __hip_atomic_fetch_min(stp + 2, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(stp + 3, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
    }, -1, nullptr, nullptr, traj);
#ifdef TAIL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && par == 1) __hip_atomic_fetch_max(stp + 4, __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// ---------------------------------------------------------------------------------------------
// k_leap: ALL inner leapfrog steps 1..L-1 of a trajectory in ONE persistent launch.
//
// During a trajectory the events are fixed: of the 20 B per cell a gradient tile reads (F fp64; I, k_se, S int32) nothing
// changes from step to step -- only the T + M table entries exp(a_t), exp(b_m)/N_m and psi do.  k_se_chunk, one launch per
// step, streams those 24.5 MB (UK-380, 8 chains) through the fabric again for every one of the 15 steps; here a tile
// workgroup loads its 16 x 64 cells ONCE, keeps them in registers (4 cells x 20 B = 20 VGPRs per lane) and per step reads
// only its 64 + 16 table entries and writes its partial sums.  Grid and roles as in k_se_chunk (tile blocks, then one
// single-wave workgroup per 64-day / 64-row chunk role), all of a chain's workgroups on one XCD (checked at creation),
// every workgroup of the launch resident at once (checked by the host through the occupancy query: tiles wait here).
// Two one-directional hand-offs per step, both through the XCD's L2 without fences (see k_se_chunk).  Since the second half
// of round 4 what is handed over travels as HAND-OFF WORDS ("Hand-off words" above: every value carries the
// step's number, the consumer's load of the value is its wait) -- no acknowledged store, no ticket, no flag between tiles
// and roles:
//   tiles -> roles  column sums, row sums, psi parts, tile scalars (Chains::llK / llR / llP / llTS, two step parities);
//   roles -> tiles  exp(a_t), exp(b_m)/N_m, spatial effects (M > 512), psi (Chains::llT).
// The counters and flags of the first version remain, in the chain's own block of Chains::leap, for what still needs a count:
// the roles among themselves (each counts in after its stores are acknowledged; the one that completes the step's count writes
// the step number to eight flag words in eight lines; a role polls copy (index mod 8) before it reads what the previous
// step's roles wrote), and the tiles' arrival at a launch's last step (Work::Lpart, plain, for the roles' accept test;
// the tiles count in on LEAP_NSH counters in lines of their own -- returning atomics on ONE address are served one after the
// other, ~26 ns apiece -- and whoever completes a counter raises that counter's flag).
// Everything a workgroup reads that another workgroup of this launch wrote is read past the L1 (agent-scope loads).
// The arithmetic is se_tile's and hmc_chunk_role's, operand for operand and in the same order: results are
// bit-identical to k_se_chunk and to k_se + k_hmc_chunk (tests/test_sampler_gpu.py).
// Every wait is bounded; a time-out is counted in Chains::late (fatal part) and the workgroup goes on, so the grid
// always drains.
// ---------------------------------------------------------------------------------------------
#ifndef LEAP_BACKOFF_N
#define LEAP_BACKOFF_N 48
#endif
constexpr int LEAP_BACKOFF = LEAP_BACKOFF_N;     // x 64 cycles slept before a tile first looks for the roles' tables: they come ~4 us
                                                 // after its sums have left (12 / 24 / 36 / 48 / 72 / 100: 135.1 / 134.4 / 133.9 / 133.8 /
                                                 // 133.7 / 133.5 us per launch at UK-380; NI-11, whose steps are shorter: level up to 48,
                                                 // 108 against 95 at 100)
constexpr int LEAP_BACKOFF_ROLE = 12;            // ... and a role before its second look at the tiles' flags (the trajectory's end)
constexpr int LEAP_NSH = 8;          // counters / flag copies per chain, 128 bytes apart
constexpr int LEAP_CH = 1024;        // 64-bit words of Chains::leap per chain
#define LEAP_CNT1(b_, k_) (ch.leap + (size_t)(b_) * LEAP_CH + (k_) * 16)            // tiles counted in on shard k, over all launches
#define LEAP_FLAG1(b_, k_) (ch.leap + (size_t)(b_) * LEAP_CH + 128 + (k_) * 16)     // last step whose shard-k tiles are all in
#define LEAP_CNT2(b_) (ch.leap + (size_t)(b_) * LEAP_CH + 256)                      // roles done, over all launches
#define LEAP_FLAG2(b_, k_) (ch.leap + (size_t)(b_) * LEAP_CH + 272 + (k_) * 16)     // last step whose roles are all done (8 copies)
#ifdef LEAP_STAMPS
// developer timeline (tools/dev/leap_timeline.py): per chain and step, min / max over the tile workgroups of "past the wait
// for the tables" (0, 1) and "counted in" (2, 3), min / max over the roles of "past the wait for the tiles" (4, 5) and "done" (6, 7)
// (LEAP_STAMPS=2; they are atomics on one line per chain and stretch what they time: =1 keeps only the probes below)
#if LEAP_STAMPS >= 2
#define LSTAMP_MIN(k) __hip_atomic_fetch_min(ch.leap_st + ((size_t)b * 16 + (it & 15)) * 8 + (k), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LSTAMP_MAX(k) __hip_atomic_fetch_max(ch.leap_st + ((size_t)b * 16 + (it & 15)) * 8 + (k), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define LSTAMP_MIN(k) do {} while (0)
#define LSTAMP_MAX(k) do {} while (0)
#endif
// and of two tile workgroups of chain 0 (the first and one in the middle) and two roles (T-chunk 0, M-chunk 0): plain stores of their own
// stamps, tiles behind the chains' blocks, roles in the blocks of chains 1 and 2
#define LPROBE(k) do { if (b == 0 && threadIdx.x == 0 && (tix == 0 || tix == nwg / 2 + 5)) \
    ch.leap_st[((size_t)s.B * 16 + (tix == 0 ? 0 : 16) + (it & 15)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// every tile workgroup of chain 0 in step 7: [tix][k], behind the probes' blocks (k = 0 past the wait, 1 cells done, 2 counted in, 3 HW_ID)
#define LALL(k) do { if (b == 0 && threadIdx.x == 0 && it == 7 && tix < 1024) \
    ch.leap_st[((size_t)s.B + 2) * 128 + (size_t)tix * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define RPROBE(k) do { if (b == 0 && threadIdx.x == 0 && (role == 0 || role == d.ntc)) \
    ch.leap_st[((size_t)(role == 0 ? 1 : 2) * 16 + (it & 15)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LSTAMP_MIN(k) do {} while (0)
#define LSTAMP_MAX(k) do {} while (0)
#define LPROBE(k) do {} while (0)
#define LALL(k) do {} while (0)
#define RPROBE(k) do {} while (0)
#endif
// A wait of k_leap.  Every wait is bounded (~1 s), and once ANY wait of the chain has timed out -- its workgroups cannot
// all have been resident: something else holds part of the chip, e.g. a second process with a launch of the same kind -- the
// chain's fatal counter is non-zero and every later wait of the chain gives up at its first look at it (every 256 polls), so
// that a launch that cannot complete drains in about a second instead of a second per step; the host finds the counter at
// the next read of the trace and fails loudly (check_handoffs).
__device__ __forceinline__ void leap_wait(const unsigned long long *flag, unsigned long long target, unsigned *late) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if ((spins & 255) == 0 && __hip_atomic_load(late, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (spins > (1 << 21)) { if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(late, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
}

// NST: 16-row gradient tiles per workgroup (the same day chunk, consecutive row tiles) -- wave w owns rows 4w..4w+3 of each of
// them, NST x 4 cells per lane.  Each tile's sums are formed exactly as se_tile forms them (bit-identical partial sums,
// whatever NST); what grows with NST is the work per wave and what shrinks is everything paid per workgroup: arrivals on
// the chain's counters, polls, barriers, reductions' fixed parts, and the number of waves the literals of the polynomials
// are materialised for.
// RW: rows per wave (a gradient tile is TM = 4 RW rows x 64 days).  RW = 4 is se_tile's own shape (its partial sums to the bit);
// RW = 6 with NST = 1 gives 24-row workgroups -- at UK-380 (Mp = 384) 96 tile workgroups per chain, exactly three per CU of the
// chain's XCD and 18 cells per lane on every SIMD, where the 32-row workgroups (72 per chain) put three on eight CUs and two on
// the other twenty-four: the step waited for the SIMDs that carry 24.  The row sums then take another order of additions
// (eight lanes per row, eight entries each): the same draws up to rounding, like every other difference between the forms.
template <int TSM, int NST, int RW = SE_RW>
__device__ __forceinline__ void leap_tile(const Dims &d, const Consts &c, const Work &w, const SamplerCfg &s, const Chains &ch,
                                          int bx, int byg, int bz, int par0, int nsteps, unsigned long long step_base,
                                          unsigned long long role_base, int fold) {
    // fold (the trajectory's end points inside this launch): bit 0 = the first of the nsteps evaluations is the gradient at the
    // START point (the roles' first step draws the momentum: k_hmc_step<0>'s work), bit 1 = the last one is the gradient at the END
    // point, for k_hmc_step<2> (no role follows it).  At those two the S->E term's value is summed as well (Lpart0 / Lpart).
    constexpr int TM = 4 * RW;
    __shared__ double colbuf[NST][4][WAVE];
    __shared__ double psibuf[NST][4][WAVE], llbuf[NST][4][WAVE];
    __shared__ double rowbuf[NST][4 * RW * SE_RS];
    __shared__ double rlbuf[TSM == 2 ? NST : 1][4][WAVE], rsbuf[TSM == 2 ? NST : 1][4][WAVE];
    __shared__ double2 ltab[LDSTAB_N];
    // the step's tables, fetched past the L1 by one wave each and handed to the others through LDS: exp(a_t) of the 64 days
    // (wave 0) | exp(b_m)/N_m of the rows (wave 1) | the rows' spatial effects (TSM 2, wave 2) | psi (wave 3).  With every
    // wave loading its own operands the 576 waves of a chain all asked the L2 for the same few lines at the same moment
    __shared__ double tabbuf[WAVE + 2 * NST * TM + 8];
    constexpr int TB_EB = WAVE, TB_SP = WAVE + NST * TM, TB_PSI = WAVE + 2 * NST * TM;
    static_assert(NST * TM <= WAVE, "one wave fetches the rows' table entries");
    auto LDP = [](const double *p_) { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    debug_skew(d);
    const int b = d.b0 + bz, wave0 = threadIdx.x >> 6, lane0 = threadIdx.x & 63;
    const int wave = wave0;                                      // (ahead of the step loop; inside it: an opaque copy)
    const int t0 = bx * WAVE + lane0;
    const int mg = byg * NST * TM;                            // first row of the workgroup's tiles
    const int ntile = d.ntc * d.nmt, nwg = ntile / NST;          // gradient tiles / tile workgroups of a chain
    if (threadIdx.x < LOGTAB_N) ltab[threadIdx.x] = c.logtab[threadIdx.x];
    // ---- the step-invariant operands, once: the tiles' cells in registers, the per-day and per-row constants
    const double Wt = c.W[t0];
    double F[NST][RW];
    int Ii[NST][RW], ki[NST][RW], si[NST][RW];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const size_t q = ((size_t)b * d.Mp + mg + st * TM + wave * RW + r) * d.Tp + t0;
            F[st][r] = w.F[q];
            ki[st][r] = w.K[0][q];
            Ii[st][r] = w.St[2][q];
            si[st][r] = w.St[0][q];
        }
    constexpr bool ts_rows = TSM == 2;
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    double ts_l[NST][RW];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int r = 0; r < RW; ++r) ts_l[st][r] = ts_rows ? c.la[mg + st * TM + wu * RW + r] : 0.0;
    const double ts_vt = w.Vt[(size_t)b * d.Tp + t0];
    // this workgroup's shard of the chain's counters: workgroups g, g + nsh, g + 2 nsh ... count in together
    const int tix = byg * d.ntc + bx, nsh = min(LEAP_NSH, nwg), shard = tix % nsh;
    const unsigned long long shard_size = (unsigned long long)((nwg - shard + nsh - 1) / nsh);
    unsigned long long *cnt1 = LEAP_CNT1(b, shard), *flag1 = LEAP_FLAG1(b, shard);
    __syncthreads();                                             // ltab
    // The S->E term's VALUE at the trajectory's two end points (fold bits 0 and 1), from the tables in `tabbuf`: NOT part of
    // the step loop.  Inside it -- a cold block behind the gradient, taken twice per launch -- its registers (the table
    // logarithm of every cell, libm for the rare rate outside the series' range) came on top of the cells the workgroup
    // keeps in registers across the loop, and the allocator parked loop-invariant cell registers in scratch and reloaded
    // them in EVERY step (20 B per lane in the 32-row form, 60-190 B in the 24-row form at 128 registers).  As passes of
    // their own before and after the loop they cost the hot path nothing.  Same operations on the same operands as
    // se_cells / se_cells_L, the sums in the order the in-loop block formed them: the same bits.
    auto ll_pass = [&](double *Lp, int wave, int lane0) {
        SeK sk;
        sk.load();
        const double psiW = tabbuf[TB_PSI] * Wt, ea_t = tabbuf[lane0];
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            double ll = 0.0;
            constexpr int CBL = RW == 6 ? 3 : RW;
#pragma unroll
            for (int r0 = 0; r0 < RW; r0 += CBL) {
                double rr[CBL], L[CBL];
#pragma unroll
                for (int r = 0; r < CBL; ++r) {
                    const double ee = ea_t * tabbuf[TB_EB + st * TM + wave * RW + r0 + r];
                    const double lam0 = ee * ((double)Ii[st][r0 + r] + psiW * F[st][r0 + r]);
                    rr[r] = (lam0 + d.rate_floor) * d.dt;
                }
                se_cells_L<CBL>(rr, ltab, sk, L);
#pragma unroll
                for (int r = 0; r < CBL; ++r) {
                    const double kse = (double)ki[st][r0 + r], snk = (double)(si[st][r0 + r] - ki[st][r0 + r]);
                    ll += (kse != 0.0 ? kse * L[r] : 0.0) - snk * rr[r];
                }
            }
            llbuf[st][wave][lane0] = ll;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const size_t tile = (size_t)b * d.nmt * d.ntc + (size_t)(byg * NST + st) * d.ntc + bx;
                const double v = wave_sum((llbuf[st][0][lane0] + llbuf[st][1][lane0]) + (llbuf[st][2][lane0] + llbuf[st][3][lane0]));
                if (lane0 == 0) Lp[tile] = v;
            }
        }
    };
    // the step's tables, one wave each (`first`: the start point of a folded trajectory -- its spatial effects are q0's own)
    auto fetch_tables = [&](int lane, int t, int par, bool first) {
        if (wu == 0) {
            tabbuf[lane] = LDP(w.ea + (size_t)b * d.Tp + t);
        } else if (wu == 1) {
            if (lane < NST * TM) tabbuf[TB_EB + lane] = LDP(w.eb + (size_t)b * d.Mp + mg + lane);
        } else if (wu == 2) {
            if (ts_rows && lane < NST * TM) {
                const int row = mg + lane;
                double v_;
                if (first) v_ = row < d.M ? LDP(ch.q0 + (size_t)b * d.Pp + 6 + d.T - 1 + row) : 0.0;
                else v_ = LDP(w.sp + ((size_t)b * 2 + par) * d.Mp + row);
                tabbuf[TB_SP + lane] = v_;
            }
        } else {
            if (lane == 0) tabbuf[TB_PSI] = LDP(w.scal + (size_t)b * NSCAL + SC_PSI);
        }
    };
    // ... from the second step on as hand-off words: each wave looks at ONE of its entries until the roles of the step before
    // have left it (all lanes at one address: one request per look, as the flag used to be), then loads all of them and looks
    // again until every one shows the step.  Padding days / rows are nobody's: they stay what the plain arrays hold.
    auto fetch_tables_ll = [&](int lane, int t, int par, unsigned seq) {
        unsigned *late = ch.late + ch.late_fatal + b;
        const uint4 *lt = ch.llT + (size_t)b * ll_tab_len(d);
        double v[1];
        if (wu == 0) {
            const uint4 *hint[1] = {lt + bx * WAVE};             // the chunk's first day: always a day of the series
            ll_poll<1>(hint, seq, late, v);
            const uint4 *pp[1] = {lt + min(t, d.T - 1)};
            ll_poll<1>(pp, seq, late, v);
            double x = v[0];
            if ((bx + 1) * WAVE > d.T) { const double pad = LDP(w.ea + (size_t)b * d.Tp + t); x = t < d.T ? x : pad; }
            tabbuf[lane] = x;
        } else if (wu == 1 || (wu == 2 && ts_rows)) {
            const int off = wu == 1 ? d.Tp : d.Tp + d.Mp;
            const double *plain = wu == 1 ? w.eb + (size_t)b * d.Mp : w.sp + ((size_t)b * 2 + par) * d.Mp;
            const int row = mg + min(lane, NST * TM - 1);
            double x;
            if (mg < d.M) {
                const uint4 *hint[1] = {lt + off + mg};
                ll_poll<1>(hint, seq, late, v);
                const uint4 *pp[1] = {lt + off + min(row, d.M - 1)};
                ll_poll<1>(pp, seq, late, v);
                x = v[0];
                if (mg + NST * TM > d.M) { const double pad = LDP(plain + row); x = row < d.M ? x : pad; }
            } else {
                x = LDP(plain + row);
            }
            if (lane < NST * TM) tabbuf[(wu == 1 ? TB_EB : TB_SP) + lane] = x;
        } else if (wu == 3) {
            const uint4 *pp[1] = {lt + d.Tp + 2 * (size_t)d.Mp};
            ll_poll<1>(pp, seq, late, v);
            if (lane == 0) tabbuf[TB_PSI] = v[0];
        }
    };
    if (fold & 1) {                                              // the start point's value: before the first step
        fetch_tables(lane0, t0, par0, true);
        __syncthreads();
        ll_pass(w.Lpart0, wave0, lane0);
        __syncthreads();                                         // (tabbuf and llbuf are written again by the first step)
    }
    for (int it = 0; it < nsteps; ++it) {
        const int par = par0 ^ (it & 1);
        // (opaque copies made inside the loop: the per-lane addresses of the step are then not loop-invariant for the
        // compiler, which would otherwise hoist a few dozen of them out of the loop and spill them)
        // (the wave index too: as a loop invariant it took the LDS and global base addresses of the step's reductions
        // with it, and two of those register pairs were parked in scratch across the loop)
        int lane = lane0, t = t0, wave = wave0;
        asm volatile("" : "+v"(lane), "+v"(t), "+v"(wave));
        // this step's number over all launches: what the tile's partial sums carry (hand-off words), and their buffer
        const unsigned long long stepno_ = step_base + (unsigned long long)(it + 1);
        const unsigned llseq = ll_seq(stepno_);
        const size_t llcb = (size_t)b * 2 + (size_t)(stepno_ & 1ull);
        {
            // every wave waits (from the second step on) for the tables of this step's position, written by the roles of
            // step it-1, and fetches its share: four requests per workgroup
            if (it > 0) {
                __builtin_amdgcn_s_sleep(LEAP_BACKOFF);
                fetch_tables_ll(lane, t, par, ll_seq(role_base + (unsigned long long)it));
            } else {
                fetch_tables(lane, t, par, (fold & 1) != 0);
            }
            if (threadIdx.x == 0) { LSTAMP_MIN(0); LSTAMP_MAX(1); }
            LPROBE(0);
            LALL(0);
        }
        SeK sk;
        sk.load();                                               // the series' literals as scalars (device_math.h), per step
        __syncthreads();
        LPROBE(1);                                               // tables in LDS
        const double psi = tabbuf[TB_PSI];
        const double ea_t = tabbuf[lane];
        const double psiW = psi * Wt;
        constexpr bool with_ll = false;                          // (the term's value: ll_pass, outside the loop)
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            double eb[RW], ts_s[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                eb[r] = tabbuf[TB_EB + st * TM + wave * RW + r];
                ts_s[r] = ts_rows ? tabbuf[TB_SP + st * TM + wave * RW + r] : 0.0;
            }
            double *myrow = rowbuf[st] + wave * RW * SE_RS;
            double gpsi = 0.0, colacc = 0.0, rlacc = 0.0, rsacc = 0.0, ll = 0.0;
            // the cells side by side (se_cells: se_tile's own evaluation).  Only the GRADIENT of the S->E term drives a
            // leapfrog step: the term's value (and with it log(1 - e^-r), a third of a cell's instructions) is needed at the
            // trajectory's end points alone (with_ll) -- everywhere else the series' logarithm is dead code on the taken path.
            // CB cells in flight at a time: all four of a 16-row tile's; of six rows per wave two batches of three, which is
            // what fits the 128 registers of four waves per SIMD (six interleaved chains of the series: 171) -- with four
            // waves on a SIMD three independent chains per wave already cover the fp64 latency
            #ifndef LEAP_CB6
#define LEAP_CB6 3
#endif
            constexpr int CB = RW == 6 ? LEAP_CB6 : RW;
#pragma unroll
            for (int r0 = 0; r0 < RW; r0 += CB) {
                double Id[CB], ebb[CB], Fb[CB], ee[CB], lam0[CB], rr[CB], L[CB], inv[CB];
#pragma unroll
                for (int r = 0; r < CB; ++r) { Id[r] = (double)Ii[st][r0 + r]; ebb[r] = eb[r0 + r]; Fb[r] = F[st][r0 + r]; }
                se_cells<CB, false>(ea_t, ebb, Id, psiW, Fb, d.rate_floor, d.dt, ltab, sk, ee, lam0, rr, L, inv);
#pragma unroll
                for (int r = 0; r < CB; ++r) {
                    const double kse = (double)ki[st][r0 + r], snk = (double)(si[st][r0 + r] - ki[st][r0 + r]);
                    const bool has = kse != 0.0;
                    const double gl = d.dt * ((has ? kse * inv[r] : 0.0) - snk);
                    const double ge = gl * lam0[r];
                    myrow[(r0 + r) * SE_RS + lane] = ge;
                    colacc += ge;
                    if (ts_rows) {
                        rlacc = fma(ge, ts_l[st][r0 + r], rlacc);
                        rsacc = fma(ge, ts_s[r0 + r], rsacc);
                    }
                    gpsi += gl * ee[r] * Wt * Fb[r];
                }
                if (with_ll) {                                   // the trajectory's end points only: a cold block, behind the hot one
                    se_cells_L<CB>(rr, ltab, sk, L);
#pragma unroll
                    for (int r = 0; r < CB; ++r) {
                        const double kse = (double)ki[st][r0 + r], snk = (double)(si[st][r0 + r] - ki[st][r0 + r]);
                        ll += (kse != 0.0 ? kse * L[r] : 0.0) - snk * rr[r];
                    }
                }
                if (CB != RW) __builtin_amdgcn_sched_barrier(0);  // one batch after the other: interleaved they need the registers of RW
            }
            psibuf[st][wave][lane] = gpsi;
            if (with_ll) llbuf[st][wave][lane] = ll;
            colbuf[st][wave][lane] = colacc;
            {
                // the row sums through one LDS transpose: LPR lanes per row, each adds WAVE / LPR entries, then shuffles
                // (RW = 4: se_tile's sixteen lanes per row; otherwise eight -- 8 RW <= 64 lanes at work)
                constexpr int LPR = RW == 4 ? 16 : 8, NE = WAVE / LPR;
                static_assert(RW * LPR <= WAVE, "rows per wave");
                const int rr_ = lane / LPR, ss = lane % LPR;
                const bool rrow = rr_ < RW;
                const double *src = myrow + (rrow ? rr_ : 0) * SE_RS + ss;
                double v = 0.0;
#pragma unroll
                for (int j = 0; j < NE; j += 2) v += src[j * LPR] + src[(j + 1) * LPR];
#pragma unroll
                for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, WAVE);
                if (ss == 0 && rrow) {
                    w.Rpart[((size_t)b * d.ntc + bx) * d.Mp + mg + st * TM + wave * RW + rr_] = v;
                    ll_store(ch.llR + (llcb * d.ntc + bx) * d.Mp + mg + st * TM + wave * RW + rr_, v, llseq);
                }
                if (ts_rows) { rlbuf[st][wave][lane] = rlacc; rsbuf[st][wave][lane] = rsacc; }
            }
        }
        LPROBE(2);                                               // cells done (wave 0)
        LALL(1);
        __syncthreads();
        LPROBE(3);                                               // past the middle barrier
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int by = byg * NST + st;
            const size_t tile = (size_t)b * d.nmt * d.ntc + (size_t)by * d.ntc + bx;
            const size_t lltile = llcb * ((size_t)d.ntc * (d.Mp / 16)) + (size_t)by * d.ntc + bx;
            if (wave == 0) {
                if (with_ll) {                                   // the end points only
                    const double v = wave_sum((llbuf[st][0][lane] + llbuf[st][1][lane]) + (llbuf[st][2][lane] + llbuf[st][3][lane]));
                    if (lane == 0) (((fold & 1) && it == 0) ? w.Lpart0 : w.Lpart)[tile] = v;
                }
            } else if (wave == 1) {
                const double v = wave_sum((psibuf[st][0][lane] + psibuf[st][1][lane]) + (psibuf[st][2][lane] + psibuf[st][3][lane]));
                if (lane == 0) { w.Ppart[tile] = v; ll_store(ch.llP + lltile, v, llseq); }
            } else if (wave == 2) {
                const double cs = (colbuf[st][0][lane] + colbuf[st][1][lane]) + (colbuf[st][2][lane] + colbuf[st][3][lane]);
                w.Kpart[((size_t)b * d.nmt + by) * d.Tp + t] = cs;
                ll_store(ch.llK + (llcb * (d.Mp / 16) + by) * d.Tp + t, cs, llseq);
                if (ts_rows) {
                    const double bs = wave_sum(cs), as = wave_sum(cs * ts_vt);
                    if (lane == 0) {
                        w.TS[tile * 4 + 0] = bs; w.TS[tile * 4 + 1] = as;
                        ll_store(ch.llTS + lltile * 4 + 0, bs, llseq); ll_store(ch.llTS + lltile * 4 + 1, as, llseq);
                    }
                }
            } else {
                if (!ts_rows) {
                    const double cs = (colbuf[st][0][lane] + colbuf[st][1][lane]) + (colbuf[st][2][lane] + colbuf[st][3][lane]);
                    const double bs = wave_sum(cs), as = wave_sum(cs * ts_vt);
                    if (lane == 0) {
                        w.TS[tile * 4 + 0] = bs; w.TS[tile * 4 + 1] = as;
                        ll_store(ch.llTS + lltile * 4 + 0, bs, llseq); ll_store(ch.llTS + lltile * 4 + 1, as, llseq);
                    }
                }
                if (ts_rows) {
                    const double rl = wave_sum((rlbuf[st][0][lane] + rlbuf[st][1][lane]) + (rlbuf[st][2][lane] + rlbuf[st][3][lane]));
                    const double rs = wave_sum((rsbuf[st][0][lane] + rsbuf[st][1][lane]) + (rsbuf[st][2][lane] + rsbuf[st][3][lane]));
                    if (lane == 0) {
                        w.TS[tile * 4 + 2] = rl; w.TS[tile * 4 + 3] = rs;
                        ll_store(ch.llTS + lltile * 4 + 2, rl, llseq); ll_store(ch.llTS + lltile * 4 + 3, rs, llseq);
                    }
                }
            }
        }
        LPROBE(4);                                               // wave 0's reductions issued
        if ((fold & 2) && it == nsteps - 1) {                    // the end point: its value as well, before the tile counts in
            __syncthreads();                                     // (llbuf: nobody reads the step's LDS sums any more)
            ll_pass(w.Lpart, wave, lane);
        }
        // The partial sums are on their way as hand-off words: nobody waits for this workgroup's count any more, except at the
        // trajectory's end, where the roles' accept test reads Work::Lpart (plain): there the stores are acknowledged first.
        // (No barrier otherwise: the next LDS writes are the tables', which nobody reads past the middle barrier.)
        if (it == nsteps - 1) __syncthreads();                   // vmcnt(0)
        LPROBE(5);
        if (threadIdx.x == 0) {
            const unsigned long long stepno = stepno_;
            const unsigned long long old = __hip_atomic_fetch_add(cnt1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == stepno * shard_size) __hip_atomic_store(flag1, stepno, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            LSTAMP_MIN(2); LSTAMP_MAX(3);
            LPROBE(6);                                           // counted in (the atomic has returned)
            LALL(2);
#ifdef LEAP_STAMPS
            if (b == 0 && it == 7 && tix < 1024) {
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                ch.leap_st[((size_t)s.B + 2) * 128 + (size_t)tix * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
            }
#endif
        }
    }
}

// waves per SIMD the instance is compiled for: the launch needs every workgroup resident, so the register budget follows from
// how many workgroups a CU must hold (RW = 6, one 24-row tile per workgroup: three tile workgroups per CU and the roles beside
// them -- four waves per SIMD, 128 VGPRs)
#ifndef LEAP_RW6_WAVES
#define LEAP_RW6_WAVES 4
#endif
constexpr int leap_waves_per_simd(int nst, int rw = SE_RW) { return rw == 6 ? LEAP_RW6_WAVES : nst == 1 ? 5 : 3; }
template <int TSM, int NTC, int NST, int RW = SE_RW>
__global__ __launch_bounds__(256, (NTC == 12 && RW == SE_RW) ? (leap_waves_per_simd(NST) < 3 ? leap_waves_per_simd(NST) : 3) : leap_waves_per_simd(NST, RW))
void k_leap(Dims d, Consts c, Work w, SamplerCfg s, Chains ch, int par0, int nsteps, unsigned long long step_base,
            unsigned long long role_base, int fold) {
    // (d.nmt: row tiles of THIS launch's shape, Mp / (4 RW) -- the host passes it; the roles read the partial sums by it)
    const int nwg = d.ntc * d.nmt / NST, n_tiles = nwg * d.aff_nb;   // tile workgroups per chain: NST gradient tiles each
    const int nroles = d.ntc + d.Mp / WAVE;
    if ((int)blockIdx.x < n_tiles) {
        int bz, tile;
        xcd_affine(blockIdx.x, nwg, d.aff_nb, bz, tile);
        if (d.nlive > 0 && bz >= d.nlive) return;      // a chain of the layout that does not exist
#if !defined(LEAP_PART) || LEAP_PART == 1
        leap_tile<TSM, NST, RW>(d, c, w, s, ch, tile % d.ntc, tile / d.ntc, bz, par0, nsteps, step_base, role_base, fold);
#endif
        return;
    }
#if defined(LEAP_PART) && LEAP_PART == 1       // (developer builds: register use of the tile part alone)
    return;
#endif
    // a role: wave 0 runs it, the workgroup's other three waves share its loads of the tiles' partial sums (role_gather)
    __shared__ double gbuf[RoleGather<(NTC > 0 ? NTC : CT_MAXC)>::SIZE];
    const int L = (int)blockIdx.x - n_tiles;
    const int bz = L % d.aff_nb, role_in = L / d.aff_nb, b_in = d.b0 + bz;
    const int role = role_in, b = b_in;
    if (d.nlive > 0 && bz >= d.nlive) return;
    debug_skew(d);
    // The role waves ahead of the tile waves they share SIMDs with: a role is one wave of serial code per step and the step ends
    // with it, while the tiles -- since the hand-off words -- are back at their cells before the roles have finished counting
    // in, and a wave that gets an issue slot only when three dense fp64 waves leave one took 1.5 us for its last ten instructions
    // (141 -> 135 us per launch; measured level while the tiles still waited for the roles' flag)
    __builtin_amdgcn_s_setprio(LEAP_ROLE_PRIO);
#ifdef LEAP_HELPER_PRIO                            // (developer builds: the helper waves at another priority than the role's own)
    if ((threadIdx.x >> 6) != 0) __builtin_amdgcn_s_setprio(LEAP_HELPER_PRIO);
#endif
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane_w = (int)(threadIdx.x & 63);
    const int nsh = min(LEAP_NSH, nwg);
    const unsigned long long *flag1 = LEAP_FLAG1(b, lane_w < nsh ? lane_w : 0);   // lane k < nsh looks at shard k
    unsigned long long *cnt2 = LEAP_CNT2(b);
    const unsigned long long *flag2 = LEAP_FLAG2(b, role & (LEAP_NSH - 1));
    unsigned *late = ch.late + ch.late_fatal + b;
    // fold bit 2 (value 4): the roles follow the end point's gradient as well -- the last half kick and the accept test
    // (hmc_chunk_role's traj 3, hmc_final_apply); without it k_hmc_step<2> does that as a launch of its own
    const bool fin_in = (fold & 4) != 0;
    const int nrole_steps = nsteps - (((fold & 2) && !fin_in) ? 1 : 0);
    const double eps_l = ch.hs[(size_t)b_in * NHS + HS_EPS];   // the trajectory's step size (written again only at its end)
    for (int it = 0; it < nrole_steps; ++it) {
        // nothing of a role lives across the steps: without this the compiler hoists the step-invariant loads of the role
        // (variances, V(t), the I->R statistics, CAR rows ...) out of the loop and spills 500+ bytes per lane to hold them
        asm volatile("" ::: "memory");
        // ... and nothing derived from the chain or the role either: scalar addresses of the role's three dozen arrays,
        // computed once ahead of the loop, were 70 of the ~370 SGPRs this kernel spills to vector lanes
        int b = b_in, role = role_in;
        asm volatile("" : "+s"(b), "+s"(role));
        const int par = par0 ^ (it & 1);
        const unsigned long long stepno = step_base + (unsigned long long)(it + 1);      // what the tiles' flags show
        const unsigned long long rstep = role_base + (unsigned long long)(it + 1);       // what the roles' flag will show
        int lane_op = lane_w;
        asm volatile("" : "+v"(lane_op));
        LeapLL ll;
        ll.seq_in = ll_seq(stepno); ll.seq_out = ll_seq(rstep); ll.pb = (int)(stepno & 1ull); ll.late = late;
        auto wait_tiles = [&] {
            int spins = 0;                                       // every shard's tiles are in: all of (up to) eight flags show the step
            // (the first look at once: a role that reaches its wait after the tiles -- most do, their waves get few issue slots
            // while the tile waves of their SIMDs are at their cells -- must not sleep first; then the back-off, once)
            while (__builtin_amdgcn_ballot_w64(__hip_atomic_load(flag1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < stepno) != 0ull) {
                if (spins == 0 && nwg >= 32) __builtin_amdgcn_s_sleep(LEAP_BACKOFF_ROLE); else __builtin_amdgcn_s_sleep(1);
                ++spins;
                if ((spins & 255) == 0 && __hip_atomic_load(late, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // see leap_wait
                if (spins > (1 << 21)) { if (lane_w == 0) __hip_atomic_fetch_add(late, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        };
        if (wv != 0) {
            // the helper waves: the rows' spatial effects of the current position once the previous step's roles are done
            // (the first step of a folded trajectory reads q itself), then the partial sums once the tiles are in
            if (it > 0) {
#ifndef LEAP_WAIT_KEEP_PRIO
                __builtin_amdgcn_s_setprio(0);        // (asleep-and-look: not ahead of the tiles it shares the SIMD with)
#endif
                leap_wait(flag2, role_base + (unsigned long long)it, late);
#ifndef LEAP_WAIT_KEEP_PRIO
                __builtin_amdgcn_s_setprio(LEAP_ROLE_PRIO);
#endif
            }
            {
                const bool first_ = (fold & 1) && it == 0;
                const double *spr_ = first_ ? ch.q0 + (size_t)b * d.Pp + 6 + d.T - 1 : w.sp + ((size_t)b * 2 + par) * d.Mp;
                role_pregather<NTC>(d, c, spr_, role, wv, lane_op, gbuf);
            }
            role_gather<NTC>(d, ch, ll, b, role, wv, lane_op, gbuf);   // (its loads are the wait for the tiles)
            lds_barrier();                                       // wave 0 is at its own, inside the role
            continue;
        }
        // what the roles of the previous step wrote (chunk sums, global parameters, spatial effects, q and p)
        if (it > 0) {
#ifndef LEAP_WAIT_KEEP_PRIO
            __builtin_amdgcn_s_setprio(0);            // (asleep-and-look: not ahead of the tiles it shares the SIMD with: 133.9 -> 132.7 us)
#endif
            leap_wait(flag2, role_base + (unsigned long long)it, late);
#ifndef LEAP_WAIT_KEEP_PRIO
            __builtin_amdgcn_s_setprio(LEAP_ROLE_PRIO);
#endif
        }
        RPROBE(0);                                               // the previous step's roles are done
        hmc_chunk_role<NTC, true, true>(d, c, w, s, ch, par, role, b, [&] {
            if (threadIdx.x == 0) { LSTAMP_MIN(4); LSTAMP_MAX(5); }
            RPROBE(1);                                           // at the wait for the step's tiles (role_gather)
        }, lane_op,
#if defined(LEAP_STAMPS) && defined(LEAP_CPROBE)      // (the stamps INSIDE a role force its loads back early: they show the order of things, not the role's duration)
        (b == 0 && role == 0) ? ch.leap_st + ((size_t)1 * 16 + (it & 15)) * 8 :
        (b == 0 && role == d.ntc) ? ch.leap_st + ((size_t)2 * 16 + (it & 15)) * 8 : nullptr,
#else
        nullptr,
#endif
        gbuf, (fin_in && it == nsteps - 1) ? 3 : (fold & 1) ? (it == 0 ? 1 : it == 1 ? 2 : 0) : 0, eps_l, it > 0, ll);
        RPROBE(2);                                               // stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this role's stores are in the XCD's L2
        RPROBE(3);
        {
            unsigned long long old = 0ull;
            if (threadIdx.x == 0) old = __hip_atomic_fetch_add(cnt2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = __builtin_amdgcn_readfirstlane((int)(old + 1 == rstep * (unsigned long long)nroles)) != 0;
            if (last && threadIdx.x < LEAP_NSH)                  // the step's last role: eight copies of the flag, one store
                __hip_atomic_store(LEAP_FLAG2(b, threadIdx.x), rstep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 0) { LSTAMP_MIN(6); LSTAMP_MAX(7); }
            RPROBE(4);                                           // counted in
        }
        if (fin_in && it == nsteps - 1) {
            // the trajectory's end: once every role has left its parts, each makes the accept test and applies it to its entries
            leap_wait(flag2, rstep, late);
            wait_tiles();                                        // ... and the tiles' value of the S->E term (Work::Lpart) is in the L2
            hmc_final_apply<NTC>(d, c, w, s, ch, par, role, b, lane_op);
        }
    }
}

// V(t) = sum_{s=1..t} var[alpha_t[s-1]] (Work::Vt), which the chunked leapfrog steps need and k_hmc_step<0> writes on its
// way: with the trajectory's first step folded into k_leap somebody else has to, whenever the variances change (set_kernel,
// set_adaptation, and every sweep of a mass-adaptation window).  One wave per chain; the additions in the order of
// k_hmc_step<0>'s block scan (blocks of HB days, wave by wave), so that the table is the same to the bit.
__global__ __launch_bounds__(WAVE) void k_vt(Dims d, Work w, Chains ch) {
    const int b = d.b0 + blockIdx.x, lane = threadIdx.x;
    const double *var = ch.var + (size_t)b * d.Pp;
    const int oT = 6 - 1;
    double vcarry = 0.0;
    for (int t0 = 0; t0 < d.Tp; t0 += HB) {
        double base = 0.0, tot = 0.0;
        for (int wv = 0; wv < HWV; ++wv) {
            const int t = t0 + wv * WAVE + lane;
            const double vv = (t >= 1 && t < d.T) ? var[oT + t] : 0.0;
            const double inc = wave_incl_scan(vv, lane);
            const double excl = base + inc - vv;
            if (t < d.Tp) w.Vt[(size_t)b * d.Tp + t] = vcarry + excl + vv;
            const double x = lane_value(inc, 63);
            tot += x;
            base += x;
        }
        vcarry += tot;
    }
}

// XCC_ID of every workgroup of a grid laid out like the XCD-affine grids: the host checks that blocks with the same
// id mod 8 share an XCD before it uses k_se_chunk.
// The trajectory's end as a launch of its own for the per-step form (hmc_mode 6): every chunk role -- one wave -- makes the
// accept test from the parts the last k_se_chunk launch's roles left and applies it to its entries (hmc_final_apply: what
// k_leap's roles do behind their last hand-off, what k_hmc_step<2> does with one workgroup per chain in the stage forms).
// Grid: (ntc + Mp / 64) x nb single-wave workgroups.
template <int NTC>
__global__ __launch_bounds__(WAVE) void k_hmc_final(Dims d, Consts c, Work w, SamplerCfg s, Chains ch, int par) {
    int bx = blockIdx.x, by = blockIdx.y;
    if (d.aff_nb > 0) xcd_affine(blockIdx.x, (NTC > 0 ? NTC : d.ntc) + d.Mp / WAVE, d.aff_nb, by, bx);
    hmc_final_apply<NTC>(d, c, w, s, ch, par, bx, d.b0 + by, (int)threadIdx.x);
}

// The spatial effects of a trajectory's start point into the buffer the first gradient launch's tiles read them from
// (Work::sp[par], M > 512 only: there the tiles form the row scalars) -- k_hmc_step<0> leaves them there in the stage forms.
__global__ __launch_bounds__(256) void k_sp_prep(Dims d, Work w, Chains ch, int par) {
    const int b = d.b0 + blockIdx.x;
    const double *q0 = ch.q0 + (size_t)b * d.Pp + 6 + d.T - 1;
    double *sp = w.sp + ((size_t)b * 2 + par) * d.Mp;
    for (int m = threadIdx.x; m < d.Mp; m += 256) sp[m] = m < d.M ? q0[m] : 0.0;
}

__global__ void k_xcc_probe(unsigned *out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v & 0xfu;
}

// Parameter tables + full parameter-dependent log-prob for the current state
// (used when the state is (re)loaded): one workgroup per chain, after k_se<false,1>.
__global__ __launch_bounds__(256) void k_chain_refresh(Dims d, Consts c, Work w, Chains ch) {
    extern __shared__ double lds_col[];
    __shared__ double sh[4];
    __shared__ double seg[256];
    __shared__ double2 ltab[LDSTAB_N];
    const int b = d.b0 + blockIdx.x;
    log_table_to_lds(ltab, c.logtab);
    const double lik = reduce_chain<false>(d, c, w, b, ch.q + (size_t)b * d.Pp, nullptr, lds_col, seg, sh, ltab);
    if (threadIdx.x == 0) {
        const double *sc = w.scal + (size_t)b * NSCAL;
        ch.hs[(size_t)b * NHS + HS_LP_THETA] = lik + sc[SC_PRIOR] + sc[SC_JAC];
        ch.hs[(size_t)b * NHS + HS_LP_CONST] = w.constsum[b];
    }
}

__global__ __launch_bounds__(256) void k_chain_tables(Dims d, Consts c, Work w, Chains ch) {
    __shared__ double sh[4];
    __shared__ double seg[256];
    param_tables(d, c, w, d.b0 + blockIdx.x, ch.q + (size_t)(d.b0 + blockIdx.x) * d.Pp, seg, sh);
}

// ---------------------------------------------------------------------------
// Event moves.
// ---------------------------------------------------------------------------
// state of compartment `comp` of row m at the start of day tau, tau in [0, T]
// (tau == T: state after the last day, i.e. the closed end of the series)
__device__ inline int comp_start(const Dims &d, const Work &w, size_t rowoff, int comp, int tau) {
    if (tau < d.T) return w.St[comp][rowoff + tau];
    const int t = d.T - 1;
    int v = w.St[comp][rowoff + t];
    if (comp == 0) v -= w.K[0][rowoff + t];
    else if (comp == 1) v += w.K[0][rowoff + t] - w.K[1][rowoff + t];
    else v += w.K[1][rowoff + t] - w.K[2][rowoff + t];
    return v;
}

struct MoveSpec { int kind, tgt, slot, scan; };   // slot 0..3 = S->E move, E->I move, S->E occult, E->I occult

// Per-cell log-likelihood pieces (the full form the differences below are taken from):
//   cn = lbinom(S,k_se) + lbinom(E,k_ei) + lbinom(I,k_ir) + k_ei L_ei - (E - k_ei) r_ei       (parameter free)
//   th = k_se log(1 - e^-rr) - (S - k_se) rr + k_ir L_ir - (I - k_ir) r_ir,  rr = (ee (I + psi W F) + floor) dt

// Log-ratio contribution of the rows a proposal updates (their own state and events change, and
// for E->I updates F moves as well): sum over the touched days of [terms(new) - terms(old)].
// Rows outside [r_lo, r_hi) are skipped (k_move_delta: the block that owns the row; the fused
// S->E kernel passes the whole range).  NT threads of the calling block take part.
// fp != nullptr: an accepted E->I update whose F band has not been written yet (see Chains::fpend);
// its contribution is added to the F values read here.
template <int NT>
__device__ __forceinline__ void own_rows_delta(const Dims &d, const Consts &c, const Work &w, int b, const Move &mv,
                                               double psi, int r_lo, int r_hi, const double2 *ltab, double &dth,
                                               double &dcn, const Move *fp = nullptr) {
    const double r_ei = d.nu * d.dt, L_ei = d.L_ei;
    const double *ea = w.ea + (size_t)b * d.Tp;
    // the (updated row, day of the hull) pairs are spread over the threads: the rows of a proposal are
    // independent latency chains (loads, then a few hundred dependent fp64 operations per cell), so they
    // run side by side instead of one after the other
    // The cells an update can change, enumerated without the gaps of the hull [LO, HI]: the closed windows [lo_k, hi_k]
    // of its sub-moves, for the row of the sub-move itself (an S->E-type update: nothing else of a row moves) or for
    // every updated row (an E->I-type update: F moves under all of them in every window); a day that lies in two
    // windows is taken with the first.  Two event-time moves at distant days made the hull -- and with it the number
    // of (mostly idle) threads -- several times the days that matter, and kept the split below from applying.
    int wpre[MMAX + 1];
    wpre[0] = 0;
#pragma unroll
    for (int i = 0; i < MMAX; ++i) wpre[i + 1] = wpre[i] + (i < mv.n ? mv.hi[i] - mv.lo[i] + 1 : 0);
    const int total_w = wpre[MMAX];
    const int total = mv.tgt == 1 ? mv.n * total_w : total_w;
    // A cell's difference is three independent pieces: the S->E term (one or two series of log(1 - e^-r)) and two
    // differences of binomial coefficients (three log-factorials each).  Each piece of each cell is an item of its own
    // for a thread -- a third of the dependent fp64 chain per item (the own-rows part of an E->I-type proposal:
    // 5.4 -> ~3 us) -- as long as the items are at most two per thread; the two binomial pieces are one code path with
    // selected operands, so that their threads share waves without diverging, and the S->E pieces sit in waves of
    // their own (`base`) where that fits too.  Otherwise (a long occult hull) a thread does all three pieces of its
    // cell, in as many passes as it takes: one round of loads per cell instead of three.
    const int al = (total + WAVE - 1) / WAVE * WAVE;
    const bool split = 3 * total <= 2 * NT;
    const int base = al + 2 * total <= NT ? al : total;
    const int nslots = split ? base + 2 * total : total;
    for (int idx = (int)threadIdx.x; idx < nslots; idx += NT) {
        int cell = idx, mask = 7;
        if (split) {
            if (idx < total) mask = 1;
            else if (idx < base) continue;
            else if (idx < base + total) { cell = idx - base; mask = 2; }
            else { cell = idx - base - total; mask = 4; }
        }
        int i0 = 0, pos = cell;
        if (mv.tgt == 1) { i0 = cell / total_w; pos = cell - i0 * total_w; }
        int k = 0, p0 = 0;
#pragma unroll
        for (int i = 1; i < MMAX; ++i)
            if (i < mv.n && pos >= wpre[i]) { k = i; p0 = wpre[i]; }
        const int t = mv.lo[k] + (pos - p0);
        if (mv.tgt == 0) i0 = k;
        bool dup = false;
#pragma unroll
        for (int i = 0; i < MMAX - 1; ++i) dup |= (mv.tgt == 1 && i < k && t >= mv.lo[i] && t <= mv.hi[i]);
        if (dup) continue;
        const int j = mv.m[i0];
        if (j < r_lo || j >= r_hi) continue;
        const size_t rowoff = ((size_t)b * d.Mp + j) * d.Tp;
        {
            // which cells can change at all is known from the descriptor alone; every load of a touched cell --
            // the mobility coefficients of the F shift as well as the planes -- is then issued in ONE batch
            const bool in_state = t > mv.lo[i0] && t <= mv.hi[i0];
            int dS = 0, dE = 0, dI = 0, dkt = 0;
            if (in_state) {
                if (mv.tgt == 0) { dS = mv.dsrc[i0]; dE = -mv.dsrc[i0]; }
                else { dE = mv.dsrc[i0]; dI = -mv.dsrc[i0]; }
            }
            if (t == mv.a[i0]) dkt += mv.dka[i0];
            if (t == mv.b[i0]) dkt += mv.dkb[i0];
            bool f_moves = false;
            if (mv.tgt == 1) {
#pragma unroll
                for (int i = 0; i < MMAX; ++i) f_moves |= (i < mv.n && t > mv.lo[i] && t <= mv.hi[i]);
            }
            if (dS == 0 && dE == 0 && dI == 0 && dkt == 0 && !f_moves) continue;
            const int dk0 = mv.tgt == 0 ? dkt : 0, dk1 = mv.tgt == 1 ? dkt : 0;
            if (mask & 1) {
                double coef[MMAX];
#pragma unroll
                for (int i = 0; i < MMAX; ++i)
                    coef[i] = (i < mv.n && mv.tgt == 1)
                                  ? c.Cstar[(size_t)mv.m[i] * d.Kp0 + j] * c.invN[mv.m[i]] * (double)(-mv.dsrc[i])
                                  : 0.0;
                double cfp[MMAX];
#pragma unroll
                for (int i = 0; i < MMAX; ++i)
                    cfp[i] = (fp && i < fp->n) ? c.Cstar[(size_t)fp->m[i] * d.Kp0 + j] * c.invN[fp->m[i]] * (double)(-fp->dsrc[i])
                                               : 0.0;
                double dF = 0.0;
#pragma unroll
                for (int i = 0; i < MMAX; ++i)
                    if (i < mv.n && t > mv.lo[i] && t <= mv.hi[i]) dF += coef[i];
                const double S = w.St[0][rowoff + t], I = w.St[2][rowoff + t], kse = w.K[0][rowoff + t];
                const double eb = w.eb[(size_t)b * d.Mp + j];
                double F = w.F[rowoff + t];
                if (fp) {                                  // summed first, as apply_f_band does: F + (c0 + c1) is then
                    double dFp = 0.0;                      // bit for bit the value the band will leave in memory
#pragma unroll
                    for (int i = 0; i < MMAX; ++i)
                        if (i < fp->n && t > fp->lo[i] && t <= fp->hi[i]) dFp += cfp[i];
                    if (dFp != 0.0) F += dFp;
                }
                const double ee = ea[t] * eb, psiW = psi * c.W[t];
                // only the terms the update changes (terms(new) - terms(old) with the rest cancelled): no I->R log
                const double rr0 = (ee * (I + psiW * F) + d.rate_floor) * d.dt;
                // (branch-free series: one straight-line block per case -- device_math.h)
                if (mv.tgt == 0) {
                    // S, k_se and E move; the S->E rate (I, F) does not
                    bool odd = false;
                    double L0 = log1mexp_series(rr0, ltab, odd);
                    if (odd) L0 = log1mexp(rr0, ltab);
                    const double k1 = kse + dk0;
                    dth += ((k1 != 0.0 ? k1 * L0 : 0.0) - (kse != 0.0 ? kse * L0 : 0.0)) - (double)(dS - dk0) * rr0;
                } else {
                    // E, k_ei and I move, and F with them; S and k_se do not
                    const double rr1 = (ee * ((I + dI) + psiW * (F + dF)) + d.rate_floor) * d.dt;
                    const double r_ir = w.rir[(size_t)b * d.Tp + t] * d.dt;
                    bool odd = false;
                    double L1 = log1mexp_series(rr1, ltab, odd), L0 = log1mexp_series(rr0, ltab, odd);
                    if (odd) { L1 = log1mexp(rr1, ltab); L0 = log1mexp(rr0, ltab); }
                    dth += (kse != 0.0 ? kse * (L1 - L0) : 0.0) - (S - kse) * (rr1 - rr0) -
                           (double)dI * r_ir;
                }
            }
            // the binomial coefficients: 4 instead of 6 (the third compartment's does not change).  Piece 0: the compartment
            // that loses or gains by the event count of the target transition (S for an S->E-type update, E for an E->I-type
            // one), piece 1: the next compartment
            const int npc = split ? 1 : 2;
            for (int h = 0; h < npc; ++h) {
                if (!(mask & 6)) break;
                const int pc = split ? ((mask & 4) ? 1 : 0) : h;
                const int cn = mv.tgt + pc;                                    // compartment whose coefficient it is: 0 S, 1 E, 2 I
                // (the piece differs from thread to thread: a select between two uniform pointers, not an indexed one)
                const auto *Sp = pc ? w.St[mv.tgt + 1] : w.St[mv.tgt];
                const auto *Kp = pc ? w.K[mv.tgt + 1] : w.K[mv.tgt];
                const double n0 = Sp[rowoff + t], k0 = Kp[rowoff + t];
                const double dn = cn == 0 ? (double)dS : cn == 1 ? (double)dE : (double)dI;
                const double dk = pc == 0 ? (double)(dk0 + dk1) : 0.0;        // the target transition's events leave compartment mv.tgt
                double extra = 0.0;
                if (cn == 1) extra = (double)dk1 * L_ei - (double)(dE - dk1) * r_ei;   // the E->I term of the row (dk1 = 0 for an S->E-type update)
                dcn += (lbinom_bf(n0 + dn, k0 + dk, ltab) - lbinom_bf(n0, k0, ltab)) + extra;
            }
        }
    }
}

// F[j][t] += sum_i Cstar[j][m_i] dI_i / N_{m_i} on each update's day window, rows [r_lo, r_hi),
// one wave per row (NW waves in the calling block)
template <int NW>
__device__ __forceinline__ void apply_f_band(const Dims &d, const Consts &c, const Work &w, int b, const Move &mv,
                                             int r_lo, int r_hi) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = r_lo + wave; j < r_hi; j += NW) {
        double coef[MMAX];
#pragma unroll
        for (int i = 0; i < MMAX; ++i)
            coef[i] = i < mv.n ? c.Cstar[(size_t)mv.m[i] * d.Kp0 + j] * c.invN[mv.m[i]] * (double)(-mv.dsrc[i]) : 0.0;
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

// End of a sweep in the paired form: the F band of the last accepted E->I update (k_move_delta applies
// the others on its way in).  grid (nrb_d, B).
__global__ __launch_bounds__(256) void k_apply_fpend(Dims d, Consts c, Work w, SamplerCfg s, Chains ch) {
    __shared__ Move fp;
    debug_skew(d);
    int bx = blockIdx.x, by = blockIdx.y;
    if (d.aff_nb > 0) xcd_affine(blockIdx.x, s.nrb_d, d.aff_nb, by, bx);
    const int b = d.b0 + by;
    if (threadIdx.x == 0) fp = ch.fpend[b];
    __syncthreads();
    if (fp.valid != 1) return;
    const int rows_per_blk = (d.M + s.nrb_d - 1) / s.nrb_d;
    apply_f_band<4>(d, c, w, b, fp, bx * rows_per_blk, min(d.M, (bx + 1) * rows_per_blk));
}

// rarely taken branch of band_delta, kept out of line so that its libm calls do not set the register
// budget of the band kernel
__device__ __attribute__((noinline)) double log1mexp_diff_slow(double r1, double r0, const double2 *ltab) {
    return log1mexp(r1, ltab) - log1mexp(r0, ltab);
}

// Change of the S->E term of a cell whose F moves by dF while its own state is unchanged:
//   k [L(r1) - L(r0)] - (S-k)(r1 - r0),  L(r) = log(1-exp(-r)),  r1 = r0 + a.
// In the small-rate regime L(r1)-L(r0) = log(r1/r0) + g(r1) - g(r0) with log(r1/r0) = 2 atanh(z),
// z = a/(2 r0 + a): one reciprocal and two short polynomials instead of two table logs, and more
// accurate than differencing them.
__device__ __forceinline__ double band_delta(double S, double I, double K0, double F, double dF, double ee,
                                             double psiW, double floor_dt, double dt, const double2 *ltab) {
    const double r0 = ee * (I + psiW * F) * dt + floor_dt;
    const double a = ee * psiW * dF * dt;
    const double r1 = r0 + a;
    double out = -(S - K0) * a;
    if (K0 != 0.0) {
        const double z = a * fast_rcp(r0 + r1);
        double dL;
        if (r0 >= L1ME_SERIES_MIN && r1 >= L1ME_SERIES_MIN && r0 <= L1ME_SERIES_MAX && r1 <= L1ME_SERIES_MAX &&
            fabs(z) <= 0.1) {
            const double z2 = z * z;
            const double at = z * (2.0 + z2 * (0.66666666666666663 + z2 * (0.4 + z2 * (0.2857142857142857 + z2 * (0.22222222222222221 +
                              z2 * (0.18181818181818182 + z2 * (0.15384615384615385 + z2 * 0.13333333333333333)))))));
            const double a2 = r0 * r0, b2 = r1 * r1;
            const double g0 = r0 * (-0.5 + r0 * (4.1666666666666664e-2 - a2 * (3.4722222222222224e-4 - a2 * (5.5114638447971785e-6 - a2 * 1.0333994708994709e-7))));
            const double g1 = r1 * (-0.5 + r1 * (4.1666666666666664e-2 - b2 * (3.4722222222222224e-4 - b2 * (5.5114638447971785e-6 - b2 * 1.0333994708994709e-7))));
            dL = at + (g1 - g0);
        } else {
            dL = log1mexp_diff_slow(r1, r0, ltab);
        }
        out += K0 * dL;
    }
    return out;
}

// k_move_delta: log-likelihood change of the pending proposal, over the cells it touches.
// grid (nrb_d, B).  Rows that only see a changed F (E->I moves): one wave per row, lanes
// over the days of the hull.  The (<= m) rows whose own state changes carry the expensive
// binomial-coefficient terms, so the whole workgroup spreads over that row's days.
// apply_f != 0 (paired form): the block first writes the F band of the previously accepted E->I update
// (Chains::fpend) for its own rows -- the only rows whose F it reads below.
// OWN: also the updated rows' part (split forms); the paired form's instance carries none of that code
// (fewer registers: more blocks in flight when there are many chains).
constexpr int DELTA_THREADS = 256;   // k_move_delta: 4 waves, 2 rows each (8 waves and 4-row blocks measured slower)
template <bool OWN>
__global__ __launch_bounds__(DELTA_THREADS) void k_move_delta(Dims d, Consts c, Work w, SamplerCfg s, Chains ch, int buf,
                                                    int apply_f) {
    __shared__ Move mvA, mvB;
    __shared__ int mv_sel;
    __shared__ Move fp;
    debug_skew(d);
    constexpr int NW = DELTA_THREADS / WAVE;
    __shared__ double sh_th[NW], sh_cn[NW];
    __shared__ double2 ltab[LDSTAB_N];
    int bx = blockIdx.x, by = blockIdx.y;
    if (d.aff_nb > 0) xcd_affine(blockIdx.x, s.nrb_d, d.aff_nb, by, bx);
    const int b = d.b0 + by, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef SEIR_STAMPS
#ifndef SEIR_STAMP_SLOT
#define SEIR_STAMP_SLOT 1
#endif
#ifndef SEIR_STAMP_BLOCK
#define SEIR_STAMP_BLOCK 0
#endif
    double *dst_hs = ch.hs + (size_t)b * NHS;
#ifdef SEIR_STAMP_PROPOSE
#define DSTAMP(i) do {} while (0)
    bool dstamp_on = false;
#else
#define DSTAMP(i) do { if (threadIdx.x == 0 && b == 0 && bx == SEIR_STAMP_BLOCK && dstamp_on) ((unsigned long long *)(dst_hs + 16))[12 + i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    bool dstamp_on = true;
#endif
#else
#define DSTAMP(i) do {} while (0)
#endif
    DSTAMP(0);
    move_copy(&mvA, ch.mv + (size_t)buf * s.B + b, 0);
    move_copy(&mvB, ch.mvfix + (size_t)buf * s.B + b, 64);
    if (threadIdx.x == 192) mv_sel = ch.mvsel[(size_t)buf * s.B + b];
    if (apply_f) move_copy(&fp, ch.fpend + b, 128);
    log_table_to_lds(ltab, c.logtab);              // includes the barrier that publishes the descriptors
    const Move &mv = mv_sel ? mvB : mvA;           // see load_pending() in moves_kernel.h
    if (apply_f && fp.valid == 1) {
        const int rows_per_blk = (d.M + s.nrb_d - 1) / s.nrb_d;
        apply_f_band<NW>(d, c, w, b, fp, bx * rows_per_blk, min(d.M, (bx + 1) * rows_per_blk));
        __syncthreads();                           // the band written above is read below
    }
#ifdef SEIR_STAMPS
    dstamp_on = mv.slot == SEIR_STAMP_SLOT;
#endif
    DSTAMP(1);
    double dth = 0.0, dcn = 0.0;
    if (mv.valid && mv.n > 0) {
        const double psi = w.scal[(size_t)b * NSCAL + SC_PSI];
        const double r_ei = d.nu * d.dt, L_ei = d.L_ei;
        const int rows_per_blk = (d.M + s.nrb_d - 1) / s.nrb_d;
        const int r_lo = bx * rows_per_blk, r_hi = min(d.M, r_lo + rows_per_blk);
        const double *ea = w.ea + (size_t)b * d.Tp;
        if (mv.any_dI) {
            for (int j = r_lo + wave; j < r_hi; j += NW) {
                bool mine = false;
#pragma unroll
                for (int i = 0; i < MMAX; ++i) mine |= (i < mv.n && mv.m[i] == j);
                if (mine) continue;                  // wave-uniform
                const size_t rowoff = ((size_t)b * d.Mp + j) * d.Tp;
                const double eb = w.eb[(size_t)b * d.Mp + j];
                double coef[MMAX];
#pragma unroll
                for (int i = 0; i < MMAX; ++i)      // Cstar is symmetric: read row m_i contiguously
                    coef[i] = i < mv.n ? c.Cstar[(size_t)mv.m[i] * d.Kp0 + j] * c.invN[mv.m[i]] * (double)(-mv.dsrc[i])
                                       : 0.0;
                for (int t = mv.LO + lane; t <= mv.HI; t += WAVE) {
                    double dF = 0.0;
#pragma unroll
                    for (int i = 0; i < MMAX; ++i)
                        if (i < mv.n && t > mv.lo[i] && t <= mv.hi[i]) dF += coef[i];
                    if (dF == 0.0) continue;
                    const double S = w.St[0][rowoff + t], I = w.St[2][rowoff + t], kse = w.K[0][rowoff + t];
                    const double F = w.F[rowoff + t];
                    dth += band_delta(S, I, kse, F, dF, ea[t] * eb, psi * c.W[t], d.rate_floor * d.dt, d.dt, ltab);
                }
            }
        }
        DSTAMP(2);
        // paired form (apply_f): k_move_pair's workgroups have the updated rows' part already (Chains::Down)
        if (OWN) own_rows_delta<DELTA_THREADS>(d, c, w, b, mv, psi, r_lo, r_hi, ltab, dth, dcn);
    }
    DSTAMP(3);
    dth = wave_sum(dth);
    dcn = wave_sum(dcn);
    if (lane == 0) { sh_th[wave] = dth; sh_cn[wave] = dcn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *out = ch.Dpart + ((size_t)b * s.nrb_d + bx) * 2;
        double a = 0.0, e = 0.0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { a += sh_th[k]; e += sh_cn[k]; }
        out[0] = a;
        out[1] = e;
    }
}

// End of sweep: record the event tensor in the reference's [M][T][3] order.  One wave per row,
// lanes over days (coalesced plane reads, no index division); grid (ceil(M/4), B).
// `advanced`: the sweep counter was already incremented by the closing k_move_pa2.
// apply_f (paired form): each wave first writes the F band of the sweep's last accepted E->I update
// (Chains::fpend) for its row, which saves the separate k_apply_fpend launch.
__global__ __launch_bounds__(256) void k_record(Dims d, Consts c, Work w, SamplerCfg s, Chains ch, int advanced, int apply_f) {
    __shared__ Move fp;
    debug_skew(d);
    const int b = d.b0 + blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wave;
    const unsigned slot = ch.sweep[b] - (unsigned)advanced - ch.slot0[0];
    if (apply_f) {                                           // uniform
        if (threadIdx.x == 0) fp = ch.fpend[b];
        __syncthreads();
        if (fp.valid == 1 && m < d.M) {
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
    }
    if (slot >= (unsigned)s.cap || m >= d.M) return;
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

// Events of planes 0 and 1 inside the occult range, per row (state (re)load; afterwards every accepted
// update keeps Work::rngtot current).  One wave per row; grid (ceil(M/4), B).
__global__ __launch_bounds__(256) void k_range_totals(Dims d, Work w, SamplerCfg s) {
    const int b = d.b0 + blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wave;
    if (m >= d.M) return;
    const size_t q0 = ((size_t)b * d.Mp + m) * d.Tp;
    int a0 = 0, a1 = 0;
    for (int t = s.tr_lo + lane; t < s.tr_hi; t += WAVE) { a0 += w.K[0][q0 + t]; a1 += w.K[1][q0 + t]; }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (lane == 0) {
        w.rngtot[((size_t)b * 2 + 0) * d.Mp + m] = a0;
        w.rngtot[((size_t)b * 2 + 1) * d.Mp + m] = a1;
    }
}

// slot0 = sweep counter of chain 0 minus `first`: the next sweep is recorded in trace slot `first`
// (unsigned arithmetic: slot = sweep - slot0 stays right across the wrap)
__global__ void k_set_slot0(Chains ch, unsigned first) { ch.slot0[0] = ch.sweep[0] - first; }

__global__ void k_advance(Chains ch, int b0, int nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb) ch.sweep[b0 + i] += 1;
}

}  // namespace seir
