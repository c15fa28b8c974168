// Device-side math helpers shared by the log-prob and sampler kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace seir {

constexpr int WAVE = 64;
constexpr int LFACT_TABLE = 64;

// log(n!) for n = 0..63, filled by seir_create (hipMemcpyToSymbol).
__constant__ double c_lfact[LFACT_TABLE];

// log Gamma(n+1) for integer-valued n >= 0.  Table below 64, Stirling series
// above: (x-1/2)ln x - x + ln(2 pi)/2 + 1/(12x) - 1/(360x^3) + 1/(1260x^5) - 1/(1680x^7),
// whose first dropped term is < 1e-19 at x >= 65.
__device__ __forceinline__ double lfact(double n, const double2 *tab);
__device__ __forceinline__ double lfact(double n) {
    if (n < (double)LFACT_TABLE) return c_lfact[(int)n];
    const double x = n + 1.0;
    const double xi = 1.0 / x, xi2 = xi * xi;
    const double corr = xi * (8.333333333333333e-2 - xi2 * (2.777777777777778e-3 - xi2 * (7.936507936507937e-4 - xi2 * 5.952380952380952e-4)));
    return (x - 0.5) * log(x) - x + 0.9189385332046727 + corr;
}

// log C(n,k); -inf for k<0 or k>n (TFP's log_combinations hits lgamma poles there).
__device__ __forceinline__ double lbinom(double n, double k) {
    if (k < 0.0 || k > n) return -INFINITY;
    return lfact(n) - lfact(k) - lfact(n - k);
}

// ---------------------------------------------------------------------------
// Table-driven log for positive normal x (the ocml log costs ~400 cycles per
// wave-op on gfx950, a third of the whole cell).  x = 2^k m, m in [1,2);
// i = top 7 mantissa bits; tab[i] = (invc, logc) with c ~ 1 + (i+1/2)/128,
// invc = fl(1/c), logc = -log(invc) (long double on the host); f = m invc - 1
// (one fma, |f| < 2^-7.9); log x = k ln2 + logc + log1p(f), log1p by its
// degree-8 Taylor polynomial (next term < 2^-75).  Absolute error < 2e-16 +
// 1 ulp(result): used only where |log x| >= 2 (x <= 1/8), so ~1e-16 relative.
// The table lives in LDS (filled from Consts.logtab by log_table_to_lds).
// ---------------------------------------------------------------------------
constexpr int LOGTAB_N = 128;
// LDS copy of the math tables: the log table followed by log(n!) for n < LFACT_TABLE packed two per entry
constexpr int LDSTAB_N = LOGTAB_N + LFACT_TABLE / 2;
__device__ __forceinline__ double fast_log(double x, const double2 *tab) {
    const long long bits = __double_as_longlong(x);
    const int k = (int)((bits >> 52) & 0x7ff) - 1023;
    const int i = (int)((bits >> 45) & 127);
    const double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    const double2 tc = tab[i];
    const double f = fma(m, tc.x, -1.0);
    double p = fma(f, -0.125, 0.14285714285714285);
    p = fma(f, p, -0.16666666666666666);
    p = fma(f, p, 0.2);
    p = fma(f, p, -0.25);
    p = fma(f, p, 0.33333333333333331);
    p = fma(f, p, -0.5);
    p = fma(f * f, p, f);
    const double kd = (double)k;
    return fma(kd, 0.69314718055994529, tc.y) + fma(kd, 2.3190468138462996e-17, p);
}

__device__ __forceinline__ double fast_rcp(double x);
// softplus(x) = max(x, 0) + log1p(e^-|x|) in ONE path for every lane (libm's two-sided form runs both sides when the
// lanes' signs differ) with the table logarithm: log1p(e) = log(w) + (e - (w - 1)) / w, w = fl(1 + e) -- the rounding of
// 1 + e given back, so the result is good to the last bit or two for any e in (0, 1].  For the chunk roles of a leapfrog
// step, where psi and sigma_space of the new point sit on the step's critical path (the stage kernels and the
// parameter tables, once per launch, keep libm's: the two agree to ~1e-16 relative).
__device__ __forceinline__ double softplus_tab(double x, const double2 *tab) {
    const double e = exp(-fabs(x));
    const double wv = 1.0 + e;
    return fmax(x, 0.0) + (fast_log(wv, tab) + (e - (wv - 1.0)) * fast_rcp(wv));
}

// reciprocal of a positive normal double: v_rcp_f64 + two Newton steps
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

// softplus(x) as above and sigmoid(x) = d softplus / dx from the same e = exp(-|x|): 1 / (1 + e) for x >= 0, e / (1 + e) below
// (k_hmc_chunk's roles formed it as exp(x - softplus(x)): a second exponential behind the first on the step's critical path)
__device__ __forceinline__ double softplus_sigmoid_tab(double x, const double2 *tab, double &sig) {
    const double e = exp(-fabs(x));
    const double wv = 1.0 + e;
    const double r = fast_rcp(wv);
    sig = x >= 0.0 ? r : e * r;
    return fmax(x, 0.0) + (fast_log(wv, tab) + (e - (wv - 1.0)) * r);
}

// Stirling with the table log (the hot kernels' form of lfact / lbinom)
__device__ __forceinline__ double lfact(double n, const double2 *tab) {
    // small n from the LDS copy: a divergent read of the __constant__ table is a global load
    if (n < (double)LFACT_TABLE) return reinterpret_cast<const double *>(tab + LOGTAB_N)[(int)n];
    const double x = n + 1.0;
    const double xi = fast_rcp(x), xi2 = xi * xi;   // the correction is < 1.3e-3: 1 ulp of 1/x is far below fp64 here
    const double corr = xi * (8.333333333333333e-2 - xi2 * (2.777777777777778e-3 - xi2 * (7.936507936507937e-4 - xi2 * 5.952380952380952e-4)));
    return (x - 0.5) * fast_log(x, tab) - x + 0.9189385332046727 + corr;
}
__device__ __forceinline__ double lbinom(double n, double k, const double2 *tab) {
    if (k < 0.0 || k > n) return -INFINITY;
    return lfact(n, tab) - lfact(k, tab) - lfact(n - k, tab);
}


__device__ __forceinline__ void log_table_to_lds(double2 *lds_tab, const double2 *__restrict__ gtab) {
    if (threadIdx.x < LDSTAB_N) lds_tab[threadIdx.x] = gtab[threadIdx.x];
    __syncthreads();
}

// L = log(1 - exp(-r)) and inv = 1/expm1(r) for a rate*dt r.  r < 0 gives NaN
// (the reference's log(1 - exp(-r)) does too).  Daily hazards are small, so the
// common path is a series around 0 (|rel err| < 3e-16 for r <= 1/8, checked
// against mpmath in tests/test_abi.py::test_series_constants):
//   log((1-e^-r)/r) = -r/2 + r^2/24 - r^4/2880 + r^6/181440 - r^8/9676800
//   1/expm1(r)      = 1/r - 1/2 + r/12 - r^3/720 + r^5/30240 - r^7/1209600
// with the table log and the Newton reciprocal above instead of expm1 + log + divide.
constexpr double L1ME_SERIES_MAX = 0.125;
constexpr double L1ME_SERIES_MIN = 1e-300;
__device__ __forceinline__ void l1me_inv(double r, double &L, double &inv, const double2 *tab) {
    if (r >= L1ME_SERIES_MIN && r <= L1ME_SERIES_MAX) {
        const double r2 = r * r, ri = fast_rcp(r);
        L = fast_log(r, tab) + r * (-0.5 + r * (4.1666666666666664e-2 - r2 * (3.4722222222222224e-4 - r2 * (5.5114638447971785e-6 - r2 * 1.0333994708994709e-7))));
        inv = ri - 0.5 + r * (8.3333333333333329e-2 - r2 * (1.3888888888888889e-3 - r2 * (3.3068783068783071e-5 - r2 * 8.2671957671957672e-7)));
    } else {                       // r > 1/8: 1 - e^-r has no cancellation; r < 0: log of a negative -> NaN
        const double e = exp(-r), om = 1.0 - e;
        L = log(om);
        inv = e / om;
    }
}
// The same pair with the literals of the common path in SCALAR registers.  A Horner step fma(x, p, c) with a literal c
// compiles to two v_mov_b32 (the literal into the accumulator's registers) + v_fmac_f64: three vector instructions where
// one v_fma_f64 with c as its scalar operand would do -- 34 of the ~62 vector instructions of a cell of the S->E term.
// In a kernel that streams its cells from memory that is hidden behind the loads (k_se: 7.0 us either way); the
// persistent leapfrog kernel keeps its cells in registers and is bound by exactly this instruction count.  SeK::load()
// makes the constants by s_mov_b32 (scalar ALU, its own issue port) through inline asm, so that the compiler sees
// run-time scalars and uses them as the one constant-bus operand a VOP3 instruction may have.  Same operations in the
// same order as l1me_inv / fast_log: bit-identical results.
template <unsigned long long BITS>
__device__ __forceinline__ double sgpr_literal() {
    unsigned lo, hi;
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "i"((unsigned)(BITS & 0xffffffffull)));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "i"((unsigned)(BITS >> 32)));
    return __hiloint2double((int)hi, (int)lo);
}
#define SEIR_SK(x) sgpr_literal<__builtin_bit_cast(unsigned long long, (double)(x))>()
struct SeK {
    double g1, g2, g3, g4, g5, g6, g7, ln2hi, ln2lo;       // fast_log
    double l1, l2, l3, l4, i1, i2, i3, i4, mhalf;          // the two series
    __device__ __forceinline__ void load() {
        g1 = SEIR_SK(-0.125); g2 = SEIR_SK(0.14285714285714285); g3 = SEIR_SK(-0.16666666666666666); g4 = SEIR_SK(0.2);
        g5 = SEIR_SK(-0.25); g6 = SEIR_SK(0.33333333333333331); g7 = SEIR_SK(-0.5);
        ln2hi = SEIR_SK(0.69314718055994529); ln2lo = SEIR_SK(2.3190468138462996e-17);
        l1 = SEIR_SK(4.1666666666666664e-2); l2 = SEIR_SK(3.4722222222222224e-4); l3 = SEIR_SK(5.5114638447971785e-6);
        l4 = SEIR_SK(1.0333994708994709e-7);
        i1 = SEIR_SK(8.3333333333333329e-2); i2 = SEIR_SK(1.3888888888888889e-3); i3 = SEIR_SK(3.3068783068783071e-5);
        i4 = SEIR_SK(8.2671957671957672e-7);
        mhalf = SEIR_SK(-0.5);
    }
};
__device__ __forceinline__ double fast_log_k(double x, const double2 *tab, const SeK &k) {
    const long long bits = __double_as_longlong(x);
    const int e = (int)((bits >> 52) & 0x7ff) - 1023;
    const int i = (int)((bits >> 45) & 127);
    const double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    const double2 tc = tab[i];
    const double f = fma(m, tc.x, -1.0);
    double p = fma(f, k.g1, k.g2);
    p = fma(f, p, k.g3);
    p = fma(f, p, k.g4);
    p = fma(f, p, k.g5);
    p = fma(f, p, k.g6);
    p = fma(f, p, k.g7);
    p = fma(f * f, p, f);
    const double kd = (double)e;
    return fma(kd, k.ln2hi, tc.y) + fma(kd, k.ln2lo, p);
}
// the series alone (callers redo the rare arguments outside [L1ME_SERIES_MIN, L1ME_SERIES_MAX] with l1me_inv): no branch,
// so the cells of a lane can be interleaved -- a dependent fp64 operation has ~32 cycles of latency against 4 of issue
__device__ __forceinline__ void l1me_inv_series_k(double r, double &L, double &inv, const double2 *tab, const SeK &k) {
    const double r2 = r * r, ri = fast_rcp(r);
    L = fast_log_k(r, tab, k) + r * (k.mhalf + r * (k.l1 - r2 * (k.l2 - r2 * (k.l3 - r2 * k.l4))));
    inv = ri - 0.5 + r * (k.i1 - r2 * (k.i2 - r2 * (k.i3 - r2 * k.i4)));
}
__device__ __forceinline__ void l1me_inv_k(double r, double &L, double &inv, const double2 *tab, const SeK &k) {
    if (r >= L1ME_SERIES_MIN && r <= L1ME_SERIES_MAX) {
        const double r2 = r * r, ri = fast_rcp(r);
        L = fast_log_k(r, tab, k) + r * (k.mhalf + r * (k.l1 - r2 * (k.l2 - r2 * (k.l3 - r2 * k.l4))));
        inv = ri - 0.5 + r * (k.i1 - r2 * (k.i2 - r2 * (k.i3 - r2 * k.i4)));
    } else {
        const double e = exp(-r), om = 1.0 - e;
        L = log(om);
        inv = e / om;
    }
}
// The series branch alone, for callers that sort out the rare lanes outside [L1ME_SERIES_MIN, L1ME_SERIES_MAX]
// themselves (the contraction's epilogue: branch-free cells can be interleaved, and the libm branch costs ~40
// registers wherever it is inlined)
__device__ __forceinline__ void l1me_inv_series(double r, double &L, double &inv, const double2 *tab) {
    const double r2 = r * r, ri = fast_rcp(r);
    L = fast_log(r, tab) + r * (-0.5 + r * (4.1666666666666664e-2 - r2 * (3.4722222222222224e-4 - r2 * (5.5114638447971785e-6 - r2 * 1.0333994708994709e-7))));
    inv = ri - 0.5 + r * (8.3333333333333329e-2 - r2 * (1.3888888888888889e-3 - r2 * (3.3068783068783071e-5 - r2 * 8.2671957671957672e-7)));
}
// The same pair for rates that are not small: the I->R rate exp(gamma0 + gamma1 wd) is ~0.25-0.5 per
// day, beyond the 4-term range above, and the libm branch (exp, log, divide: ~400 instructions) would be
// taken by every lane of the single-wave HMC kernels.  8 Bernoulli terms reach r <= 3/4 at < 1e-16:
//   log((1-e^-r)/r) = -r/2 + sum_n B_2n r^2n / (2n (2n)!),   1/expm1(r) = 1/r - 1/2 + sum_n B_2n r^(2n-1) / (2n)!
constexpr double L1ME_WIDE_MAX = 0.75;
__device__ __forceinline__ void l1me_inv_wide(double r, double &L, double &inv, const double2 *tab) {
    if (r >= L1ME_SERIES_MIN && r <= L1ME_WIDE_MAX) {
        const double r2 = r * r, ri = fast_rcp(r);
        double pl = fma(r2, -2.1185501852016143e-14, 9.5589546647747706e-13);
        pl = fma(r2, pl, -4.4034917822395777e-11);
        pl = fma(r2, pl, 2.0876756987868099e-9);
        pl = fma(r2, pl, -1.0333994708994709e-7);
        pl = fma(r2, pl, 5.5114638447971785e-6);
        pl = fma(r2, pl, -3.4722222222222224e-4);
        pl = fma(r2, pl, 4.1666666666666664e-2);
        L = fast_log(r, tab) + r * (-0.5 + r * pl);
        double pi = fma(r2, -3.3896802963225829e-13, 1.3382536530684679e-11);
        pi = fma(r2, pi, -5.2841901386874932e-10);
        pi = fma(r2, pi, 2.0876756987868099e-8);
        pi = fma(r2, pi, -8.2671957671957672e-7);
        pi = fma(r2, pi, 3.3068783068783071e-5);
        pi = fma(r2, pi, -1.3888888888888889e-3);
        pi = fma(r2, pi, 8.3333333333333329e-2);
        inv = ri - 0.5 + r * pi;
    } else {
        const double e = exp(-r), om = 1.0 - e;
        L = log(om);
        inv = e / om;
    }
}
__device__ __forceinline__ double log1mexp(double r, const double2 *tab) {
    if (r >= L1ME_SERIES_MIN && r <= L1ME_SERIES_MAX) {
        const double r2 = r * r;
        return fast_log(r, tab) + r * (-0.5 + r * (4.1666666666666664e-2 - r2 * (3.4722222222222224e-4 - r2 * (5.5114638447971785e-6 - r2 * 1.0333994708994709e-7))));
    }
    return r > L1ME_SERIES_MAX ? log(1.0 - exp(-r)) : log(-expm1(-r));
}
// same without a table (cold paths)
__device__ __forceinline__ double log1mexp(double r) { return log(-expm1(-r)); }

// Branch-free forms for code that evaluates a handful of these per lane in one go (own_rows_delta: twelve log-factorials
// and two log(1 - e^-r) per touched cell).  With the branches of lfact / lbinom each call is a basic block of its own and
// the calls run one after the other -- ~30 dependent fp64 operations of ~32 cycles each, twelve times: the ~5 us the
// own-rows part of an event update took; as selects between the table value and the Stirling value (the same values the
// branches return: bit-identical results) the calls are one straight-line block and the compiler interleaves them.
__device__ __forceinline__ double lfact_bf(double n, const double2 *tab) {
    const bool small = n < (double)LFACT_TABLE;
    const double tv = reinterpret_cast<const double *>(tab + LOGTAB_N)[small ? (n > 0.0 ? (int)n : 0) : 0];
    const double x = n + 1.0;
    const double xi = fast_rcp(x), xi2 = xi * xi;
    const double corr = xi * (8.333333333333333e-2 - xi2 * (2.777777777777778e-3 - xi2 * (7.936507936507937e-4 - xi2 * 5.952380952380952e-4)));
    const double st = (x - 0.5) * fast_log(x, tab) - x + 0.9189385332046727 + corr;
    return small ? tv : st;
}
__device__ __forceinline__ double lbinom_bf(double n, double k, const double2 *tab) {
    const double v = lfact_bf(n, tab) - lfact_bf(k, tab) - lfact_bf(n - k, tab);
    return (k < 0.0 || k > n) ? -INFINITY : v;
}
// log(1 - e^-r) by the series alone; `odd` is raised for an argument outside the series' range (the caller redoes those
// with log1mexp in a cold block)
__device__ __forceinline__ double log1mexp_series(double r, const double2 *tab, bool &odd) {
    const double r2 = r * r;
    odd = odd || !(r >= L1ME_SERIES_MIN && r <= L1ME_SERIES_MAX);
    return fast_log(r, tab) + r * (-0.5 + r * (4.1666666666666664e-2 - r2 * (3.4722222222222224e-4 - r2 * (5.5114638447971785e-6 - r2 * 1.0333994708994709e-7))));
}


// Out-of-line libm for the once-per-launch scalar work: inlining every exp/log/log1p copy makes
// the single-workgroup kernels several thousand instructions of straight-line code that is
// fetched cold on every launch.
// The once-per-launch scalar paths (softplus of two parameters, the accept test's log, the Box-Muller normals) used to
// be real calls, to keep the single-pass kernels' images small.  A kernel that contains a call needs a private segment,
// and the dispatcher places the waves of such a kernel more slowly: k_se_chunk's 384 workgroups and the stage kernels
// lost ~12 us per sweep to it (r02).  SEIR_COLD_CALLS restores the calls for comparison.
#ifdef SEIR_COLD_CALLS
#define SEIR_COLD __attribute__((noinline))
#else
#define SEIR_COLD __forceinline__
#endif
__device__ SEIR_COLD double cold_exp(double x) { return exp(x); }
__device__ SEIR_COLD double cold_log(double x) { return log(x); }
__device__ SEIR_COLD double softplus(double x) {
    return x > 0.0 ? x + log1p(cold_exp(-x)) : log1p(cold_exp(x));
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains every outstanding
// global load/store (s_waitcnt vmcnt(0)), which serialises prefetched loads behind each reduction.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// DPP data movement of a double (two v_mov_b32_dpp): lanes that are masked off or whose
// source falls outside the 16-lane row receive 0.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_mov0(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes with the GCN DPP reduction (row_shr 1,2,3,4,8, row_bcast 15,31):
// register-to-register moves (~8 cycles) instead of ds_bpermute round trips through the LDS
// crossbar (~130 cycles each).  The result is uniform (read from lane 63).
template <>
__device__ __forceinline__ double wave_sum<double>(double v0) {
    double v = v0 + dpp_mov0<0x111, 0xf, 0xf>(v0);
    v += dpp_mov0<0x112, 0xf, 0xf>(v0);
    v += dpp_mov0<0x113, 0xf, 0xf>(v0);
    v += dpp_mov0<0x114, 0xf, 0xe>(v);
    v += dpp_mov0<0x118, 0xf, 0xc>(v);
    v += dpp_mov0<0x142, 0xa, 0xf>(v);
    v += dpp_mov0<0x143, 0xc, 0xf>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, WAVE));
    return v;
}

// inclusive prefix sum across the 64 lanes of a wave
// DPP moves for the scans: a lane whose source is outside its 16-lane row, or whose row is
// masked off, receives 0 (old = 0, bound_ctrl off).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_get0(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get0(double v) {
    return dpp_mov0<CTRL, ROW_MASK, 0xf>(v);
}
__device__ __forceinline__ int lane_value(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double lane_value(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Inclusive prefix sum across the 64 lanes of a wave: Hillis-Steele inside each row of 16 with
// row_shr 1,2,4,8, then the row totals with row_bcast 15 (into rows 1,3) and row_bcast 31 (into
// rows 2,3) -- six register-to-register steps instead of six ds_bpermute round trips per word.
template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v, int lane) {
    (void)lane;
    v += dpp_get0<0x111, 0xf>(v);
    v += dpp_get0<0x112, 0xf>(v);
    v += dpp_get0<0x114, 0xf>(v);
    v += dpp_get0<0x118, 0xf>(v);
    v += dpp_get0<0x142, 0xa>(v);
    v += dpp_get0<0x143, 0xc>(v);
    return v;
}

// Inclusive suffix sum across the 64 lanes of a wave (lane l gets sum_{j>=l}): row_shl 1,2,4,8
// inside each row, then the totals of the rows behind (lane 0 of each row) through readlane.
__device__ __forceinline__ double wave_incl_suffix_scan(double v, int lane) {
    v += dpp_get0<0x101, 0xf>(v);
    v += dpp_get0<0x102, 0xf>(v);
    v += dpp_get0<0x104, 0xf>(v);
    v += dpp_get0<0x108, 0xf>(v);
    const double t1 = lane_value(v, 16), t2 = lane_value(v, 32), t3 = lane_value(v, 48);
    const int row = lane >> 4;
    const double behind = row == 0 ? (t1 + t2) + t3 : row == 1 ? t2 + t3 : row == 2 ? t3 : 0.0;
    return v + behind;
}

// Exclusive prefix sum over a 256-thread block (thread order); `sh` needs 4
// elements.  Two barriers.  `total` receives the block sum.
template <typename T>
__device__ __forceinline__ T block_excl_scan_256(T v, T *sh, T &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T inc = wave_incl_scan(v, lane);
    __syncthreads();
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    T base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < wave) base += sh[k];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    return base + inc - v;
}

// Inclusive suffix sum over a 256-thread block: thread i gets sum_{j>=i} v_j.
__device__ __forceinline__ double block_incl_suffix_scan_256(double v, double *sh, double &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double inc = wave_incl_suffix_scan(v, lane);
    __syncthreads();
    if (lane == 0) sh[wave] = inc;
    __syncthreads();
    double tail = 0.0;
#pragma unroll
    for (int k = 3; k >= 0; --k)
        if (k > wave) tail += sh[k];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    return inc + tail;
}

// Sum over a 256-thread block; `sh` needs 4 doubles.  Result in every thread.
__device__ __forceinline__ double block_sum_256(double v, double *sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

}  // namespace seir
