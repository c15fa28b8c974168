// Developer probe: cost of executing cold straight-line code once (instruction fetch) vs a loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N>
__global__ __launch_bounds__(256) void k_straight(unsigned long long *out, double *sink) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + 1e-3 * threadIdx.x + j;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) a[i & 7] = fma(a[i & 7], 1.0000001 + 1e-9 * i, 1e-9 * (i + 1));   // distinct constants: no folding
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
    if (s == 123.456) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_loop(unsigned long long *out, double *sink, int n) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + 1e-3 * threadIdx.x + j;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fma(a[j], 1.0000001, 1e-9);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
    if (s == 123.456) sink[0] = s;
}
__global__ void k_big(double *p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001 + 1.0; }

int main() {
    unsigned long long *out; CK(hipMalloc(&out, 8 * 4096));
    double *sink; CK(hipMalloc(&sink, 1 << 24)); CK(hipMemset(sink, 0, 1 << 24));
    std::vector<unsigned long long> h(4096);
    auto rd = [&](int blocks) { hipDeviceSynchronize(); hipMemcpy(h.data(), out, 8 * blocks, hipMemcpyDeviceToHost); unsigned long long mx = 0, mn = ~0ull; for (int i = 0; i < blocks; ++i) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; } printf("min %llu max %llu cycles", mn, mx); };
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_big, dim3(2048), dim3(256), 0, 0, sink, 1 << 21);
        hipLaunchKernelGGL(k_straight<4096>, dim3(8), dim3(256), 0, 0, out, sink);
        printf("straight 4096 fma (8 blocks), after big kernel, rep %d: ", rep); rd(8); printf("\n");
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_straight<4096>, dim3(8), dim3(256), 0, 0, out, sink);
        printf("straight 4096 fma (8 blocks), back-to-back rep %d: ", rep); rd(8); printf("\n");
    }
    hipLaunchKernelGGL(k_straight<4096>, dim3(1024), dim3(256), 0, 0, out, sink);
    printf("straight 4096 fma (1024 blocks): "); rd(1024); printf("\n");
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_big, dim3(2048), dim3(256), 0, 0, sink, 1 << 21);
        hipLaunchKernelGGL(k_loop, dim3(8), dim3(256), 0, 0, out, sink, 4096);
        printf("loop 4096 fma (8 blocks), after big kernel: "); rd(8); printf("\n");
    }
    hipLaunchKernelGGL(k_straight<512>, dim3(8), dim3(256), 0, 0, out, sink);
    printf("straight 512 fma (8 blocks), cold: "); rd(8); printf("\n");
    return 0;
}
