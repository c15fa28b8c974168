// Developer probe: fp64 instruction / libm-call throughput per wave on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o math_probe math_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP, int ILP>
__global__ __launch_bounds__(256) void k(unsigned long long *out, double *sink, int iters, double seed) {
    double a[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) a[j] = seed + 1e-3 * threadIdx.x + 1e-2 * j;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            if (OP == 0) a[j] = fma(a[j], 1.0000001, 1e-9);
            if (OP == 1) a[j] = log(a[j]) + 2.0;
            if (OP == 2) a[j] = 1.0 / a[j] + 0.5;
            if (OP == 3) a[j] = expm1(-a[j] * 1e-3) + 1.5;
            if (OP == 4) a[j] = exp(-a[j] * 1e-3) + 0.5;
            if (OP == 5) a[j] = (double)((float)a[j] * 1.0001f) + 1e-9;     // f32 round trip
            if (OP == 6) a[j] = a[j] * 1.0000001 + 1e-9;
            if (OP == 7) a[j] = sqrt(a[j]) + 1.0;
            if (OP == 8) a[j] = __builtin_amdgcn_rcp(a[j]) + 0.5;
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0; 
#pragma unroll
    for (int j = 0; j < ILP; ++j) s += a[j];
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
    if (s == 123.456) sink[0] = s;
}

template <int OP, int ILP>
static int go(const char *name, unsigned long long *out, double *sink, int waves_per_simd) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 1 wave per SIMD each
    hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), 0, 0, out, sink, iters, 1.5);
    hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), 0, 0, out, sink, iters, 1.5);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks);
    CK(hipMemcpy(h.data(), out, 8 * blocks, hipMemcpyDeviceToHost));
    double cyc = (double)h[blocks / 2] / iters / ILP;
    printf("%-8s ILP=%d waves/SIMD=%d : %.1f cycles per op per wave  (%.1f cycles/op/SIMD)\n", name, ILP, waves_per_simd, cyc, cyc / waves_per_simd);
    return 0;
}

int main() {
    unsigned long long *out; CK(hipMalloc(&out, 8 * 4096));
    double *sink; CK(hipMalloc(&sink, 64));
    go<0, 1>("fma", out, sink, 1); go<0, 4>("fma", out, sink, 1); go<0, 8>("fma", out, sink, 1); go<0, 8>("fma", out, sink, 2); go<0, 8>("fma", out, sink, 4);
    go<6, 8>("mul+add", out, sink, 1);
    go<1, 1>("log", out, sink, 1); go<1, 4>("log", out, sink, 1); go<1, 8>("log", out, sink, 1); go<1, 8>("log", out, sink, 2);
    go<2, 1>("div", out, sink, 1); go<2, 8>("div", out, sink, 1); go<2, 8>("div", out, sink, 2);
    go<8, 8>("rcp", out, sink, 1);
    go<3, 1>("expm1", out, sink, 1); go<3, 8>("expm1", out, sink, 1);
    go<4, 8>("exp", out, sink, 1);
    go<7, 8>("sqrt", out, sink, 1);
    go<5, 8>("f32trip", out, sink, 1);
    return 0;
}
