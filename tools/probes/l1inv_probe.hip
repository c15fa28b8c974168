// Developer probe: does `buffer_inv sc0` drop a CU's vector L1 on gfx950 (as `buffer_inv sc1` does), and what does each cost?
// Two workgroups on one XCD (block ids 0 and 8 of a 16-block grid), different CUs.  The consumer reads X with a plain
// load (the line is now in its L1), tells the producer, which writes X, drains and raises a flag; the consumer polls the flag
// past the L1, then [nothing | buffer_inv sc0 | buffer_inv sc1] and reads X again with a plain load.
//   hipcc --offload-arch=gfx950 -O2 -o l1inv_probe l1inv_probe.hip && ./l1inv_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned ld_plain(const unsigned *p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ void k_probe(unsigned *X, unsigned *sig, unsigned *out, unsigned long long *tm, int rounds, unsigned *big, int nbig) {
    const int bx = blockIdx.x;
    if (bx != 0 && bx != 8) return;
    if (threadIdx.x != 0) return;
    for (int r = 1; r <= rounds; ++r) {
        if (bx == 0) {                                       // consumer
            // touch more lines so that L2 holds a working set (what an L2-wide invalidate would cost shows in the reload)
            unsigned acc = 0;
            for (int i = 0; i < nbig; i += 32) acc += ld_plain(big + i);
            const unsigned before = ld_plain(X + 32 * (r & 7));   // (last, so that the line is in the L1 when the flag comes)
            __hip_atomic_store(sig, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(sig + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r) __builtin_amdgcn_s_sleep(1);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            if (MODE == 1) asm volatile("buffer_inv sc0" ::: "memory");
            if (MODE == 2) asm volatile("buffer_inv sc1" ::: "memory");
            const unsigned after = ld_plain(X + 32 * (r & 7));
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            for (int i = 0; i < nbig; i += 32) acc += ld_plain(big + i);
            const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
            out[r * 4 + 0] = before; out[r * 4 + 1] = after; out[r * 4 + 2] = acc;
            tm[r * 2] = t1 - t0; tm[r * 2 + 1] = t2 - t1;
        } else {                                             // producer
            while (__hip_atomic_load(sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r) __builtin_amdgcn_s_sleep(1);
            X[32 * (r & 7)] = 1000u + (unsigned)r;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(sig + 32, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main() {
    const int rounds = 20, nbig = 16 * 1024;                 // 64 KB of other lines
    unsigned *X, *sig, *out, *big; unsigned long long *tm;
    hipMalloc(&X, 4096); hipMalloc(&sig, 4096); hipMalloc(&out, 4096); hipMalloc(&tm, 4096); hipMalloc(&big, nbig * 4);
    const char *names[3] = {"nothing", "buffer_inv sc0", "buffer_inv sc1"};
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(X, 0, 4096); hipMemset(sig, 0, 4096); hipMemset(out, 0, 4096); hipMemset(tm, 0, 4096); hipMemset(big, 0, nbig * 4);
        if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(16), dim3(64), 0, 0, X, sig, out, tm, rounds, big, nbig);
        if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(16), dim3(64), 0, 0, X, sig, out, tm, rounds, big, nbig);
        if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(16), dim3(64), 0, 0, X, sig, out, tm, rounds, big, nbig);
        hipDeviceSynchronize();
        std::vector<unsigned> h(1024); std::vector<unsigned long long> t(512);
        hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost); hipMemcpy(t.data(), tm, 4096, hipMemcpyDeviceToHost);
        int fresh = 0; double a = 0, b2 = 0;
        for (int r = 9; r <= rounds; ++r) {                   // from round 9 on the slot of X was read (cached) 8 rounds ago as well
            fresh += h[r * 4 + 1] == 1000u + (unsigned)r;
            a += t[r * 2] * 10.0; b2 += t[r * 2 + 1] * 10.0;
        }
        printf("%-16s fresh re-reads %d of %d (before: %u, after: %u in the last round); [inv + load] %.0f ns, reload of 512 other lines %.0f ns\n",
               names[mode], fresh, rounds - 8, h[rounds * 4], h[rounds * 4 + 1], a / (rounds - 8), b2 / (rounds - 8));
    }
    return 0;
}
