// Developer probe: would the inner leapfrog steps be faster as ONE persistent launch?
// A step of the sampler's HMC part is k_se_chunk: ~1152 tile workgroups stream a chain's planes (20 B per cell, the
// same bytes every step), count themselves in, and 12 one-wave role workgroups per chain turn the tiles' partial sums
// into the next step's tables.  Every launch fetches the planes through the fabric again (FETCH_SIZE: 24.5 MB per
// launch).  This probe runs the same shape of work
//   persistent: tiles loop over the steps inside one launch; per step  [tiles stream | A: tiles -> roles | roles | B:
//               roles -> tiles], both hand-offs XCD-local (relaxed L2 atomics, L1-bypassing loads of what was handed over)
//   per launch: one launch per step, tiles + roles with the A hand-off inside (k_se_chunk's form)
// and prints us per step for both.  Chains = XCDs = 8, block id mod 8 = chain.
// Build: make -C tools/probes persist_step_probe;  run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int G = 8, MP = 384, TP = 384, NTC = TP / 64;     // UK-380: padded rows x days
constexpr int STRIDE = 32;                                  // counters: one 128-B line per chain

struct Args {
    const double *F; const int *K, *S, *I;                  // [G][MP][TP]
    double *part;                                           // [G][ntile][64]   tiles' partial sums (per day)
    double *tab;                                            // [G][2][TP]       roles' table for the next step
    unsigned long long *cntA, *cntB;                        // [G][STRIDE]
    int rows_per_wg, nsteps;
    unsigned long long baseA, baseB;
};

__device__ __forceinline__ double ld2(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void spin(const unsigned long long *c, unsigned long long target) {
    int n = 0;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++n > (1 << 22)) break;
    }
}

// one tile: rows [r0, r0 + rows) x days [64 bx, 64 bx + 64) of chain g; wave = rows/4 rows, lane = day
__device__ __forceinline__ void tile(const Args &a, int g, int bx, int r0, int slot, int par, bool coh) {
    __shared__ double red[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = bx * 64 + lane;
    const int rpw = a.rows_per_wg / 4;
    const double tb = coh ? ld2(a.tab + ((size_t)g * 2 + par) * TP + t) : a.tab[((size_t)g * 2 + par) * TP + t];
    double acc = 0.0;
    for (int r4 = 0; r4 < rpw; r4 += 4) {
        double f[4]; int k[4], s[4], i[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t q = ((size_t)g * MP + r0 + wave * rpw + r4 + r) * TP + t;
            f[r] = a.F[q]; k[r] = a.K[q]; s[r] = a.S[q]; i[r] = a.I[q];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double x = tb * ((double)i[r] + 0.3 * f[r]);
#pragma unroll
            for (int j = 0; j < 12; ++j) x = fma(x, 0.999, 1e-3 * (double)k[r]);      // ~ the cell's arithmetic
            acc += x - (double)(s[r] - k[r]) * tb;
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) a.part[((size_t)g * (NTC * (MP / a.rows_per_wg)) + slot) * 64 + lane] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    __syncthreads();                                        // vmcnt(0)
}

// role c of chain g (one wave): days 64 c .. 64 c + 63: sum the row tiles' partials, write the next table
__device__ __forceinline__ void role(const Args &a, int g, int c, int par) {
    const int lane = threadIdx.x, nmt = MP / a.rows_per_wg;
    double v[24], sum = 0.0;
    for (int j0 = 0; j0 < nmt; j0 += 24) {
#pragma unroll
        for (int j = 0; j < 24; ++j) v[j] = ld2(a.part + ((size_t)g * (NTC * nmt) + (size_t)min(j0 + j, nmt - 1) * NTC + c) * 64 + lane);
#pragma unroll
        for (int j = 0; j < 24; ++j) sum += j0 + j < nmt ? v[j] : 0.0;
    }
    double x = sum * 1e-9;
#pragma unroll
    for (int j = 0; j < 200; ++j) x = fma(x, 0.99, 1e-4);    // ~ the role's arithmetic (exp, series ...)
    a.tab[((size_t)g * 2 + (par ^ 1)) * TP + c * 64 + lane] = 1.0 + 1e-6 * x;
}

__global__ __launch_bounds__(256) void k_persistent(Args a) {
    const int nmt = MP / a.rows_per_wg, ntile = NTC * nmt, n_tiles = ntile * G;
    const int L = blockIdx.x, g = L & 7;
    if (L < n_tiles) {
        const int tl = L >> 3, bx = tl % NTC, by = tl / NTC;
        for (int st = 0; st < a.nsteps; ++st) {
            if (st > 0) { if (threadIdx.x == 0) spin(a.cntB + g * STRIDE, a.baseB + (unsigned long long)st * 2 * NTC); __syncthreads(); }
            tile(a, g, bx, by * a.rows_per_wg, by * NTC + bx, st & 1, st > 0);
            if (threadIdx.x == 0) __hip_atomic_fetch_add(a.cntA + g * STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (threadIdx.x >= 64) return;
    const int c = (L - n_tiles) >> 3;                        // 2 NTC roles per chain (T-chunks, M-chunks): same work here
    for (int st = 0; st < a.nsteps; ++st) {
        spin(a.cntA + g * STRIDE, a.baseA + (unsigned long long)(st + 1) * ntile);
        role(a, g, c % NTC, st & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.cntB + g * STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void k_step(Args a, int st) {          // k_se_chunk's form: one launch per step
    const int nmt = MP / a.rows_per_wg, ntile = NTC * nmt, n_tiles = ntile * G;
    const int L = blockIdx.x, g = L & 7;
    if (L < n_tiles) {
        const int tl = L >> 3, bx = tl % NTC, by = tl / NTC;
        tile(a, g, bx, by * a.rows_per_wg, by * NTC + bx, st & 1, false);
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.cntA + g * STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (threadIdx.x >= 64) return;
    const int c = (L - n_tiles) >> 3;
    spin(a.cntA + g * STRIDE, a.baseA + (unsigned long long)(st + 1) * ntile);
    role(a, g, c % NTC, st & 1);
}

int main() {
    hipStream_t stq; CK(hipStreamCreateWithFlags(&stq, hipStreamNonBlocking));
    const size_t cells = (size_t)G * MP * TP;
    Args a{};
    double *F; int *K, *S, *I;
    CK(hipMalloc(&F, cells * 8)); CK(hipMalloc(&K, cells * 4)); CK(hipMalloc(&S, cells * 4)); CK(hipMalloc(&I, cells * 4));
    std::vector<double> hf(cells, 0.5); std::vector<int> hk(cells, 3), hs(cells, 1000), hi(cells, 50);
    CK(hipMemcpy(F, hf.data(), cells * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(K, hk.data(), cells * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(S, hs.data(), cells * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(I, hi.data(), cells * 4, hipMemcpyHostToDevice));
    a.F = F; a.K = K; a.S = S; a.I = I;
    CK(hipMalloc(&a.part, (size_t)G * NTC * (MP / 16) * 64 * 8)); CK(hipMalloc(&a.tab, (size_t)G * 2 * TP * 8));
    std::vector<double> ht((size_t)G * 2 * TP, 1.0);
    CK(hipMemcpy(a.tab, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&a.cntA, G * STRIDE * 8)); CK(hipMalloc(&a.cntB, G * STRIDE * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_persistent, 256, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("CUs %d, resident workgroups per CU %d\n", prop.multiProcessorCount, occ);
    for (int rows : {16, 32, 48}) {
        a.rows_per_wg = rows;
        const int nmt = MP / rows, ntile = NTC * nmt, grid = (ntile + 2 * NTC) * G;
        const int nsteps = 15, reps = 20;
        // ---- one launch per step
        float ms_l = 0.f;
        {
            CK(hipMemsetAsync(a.cntA, 0, G * STRIDE * 8, stq));
            unsigned long long base = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, stq));
                for (int r = 0; r < reps; ++r) {
                    a.baseA = base;                                   // the counter only grows: one base per trajectory
                    for (int st = 0; st < nsteps; ++st) hipLaunchKernelGGL(k_step, dim3(grid), dim3(256), 0, stq, a, st);
                    base += (unsigned long long)nsteps * ntile;
                }
                CK(hipEventRecord(e1, stq)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_l, e0, e1));
            }
        }
        // ---- persistent
        float ms_p = 0.f;
        if ((long)grid <= (long)occ * prop.multiProcessorCount) {
            CK(hipMemsetAsync(a.cntA, 0, G * STRIDE * 8, stq)); CK(hipMemsetAsync(a.cntB, 0, G * STRIDE * 8, stq));
            a.nsteps = nsteps;
            unsigned long long bA = 0, bB = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, stq));
                for (int r = 0; r < reps; ++r) {
                    a.baseA = bA; a.baseB = bB;
                    hipLaunchKernelGGL(k_persistent, dim3(grid), dim3(256), 0, stq, a);
                    bA += (unsigned long long)nsteps * ntile; bB += (unsigned long long)nsteps * 2 * NTC;
                }
                CK(hipEventRecord(e1, stq)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_p, e0, e1));
            }
        }
        printf("rows per tile workgroup %2d (%4d tile workgroups): one launch per step %.2f us/step; persistent %.2f us/step (%.1f us per 15-step launch)\n",
               rows, ntile * G, ms_l * 1e3 / (reps * nsteps), ms_p * 1e3 / (reps * nsteps), ms_p * 1e3 / reps);
    }
    return 0;
}
