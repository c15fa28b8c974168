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

// log(1 - exp(-r)); NaN for r < 0 exactly as log(1 - exp(-r)) in the reference.
__device__ __forceinline__ double log1mexp(double r) { return log(-expm1(-r)); }

__device__ __forceinline__ double softplus(double x) {
    return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x));
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, WAVE));
    return v;
}

// inclusive prefix sum across the 64 lanes of a wave
template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v, int lane) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        T n = __shfl_up(v, o, WAVE);
        if (lane >= o) v += n;
    }
    return v;
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
