// Developer probe: launch floor of dependent kernels (eager, graph) and the
// shader clock held in a short fp64 loop.  Build: hipcc --offload-arch=gfx950 -O3 -o launch_probe launch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(int *p) { if (threadIdx.x == 0 && p[0] == 12345) p[1] = 1; }
__global__ void k_small(double *p, int n) {   // one block, reads+writes n doubles, a few syncs
    __shared__ double sh[256];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += p[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    for (int i = threadIdx.x; i < n; i += 256) p[i] = p[i] * 0.5 + sh[0] * 1e-9;
}
__global__ void k_clock(unsigned long long *out, double *sink, int iters) {
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a = threadIdx.x * 1e-3, b = 1.0000001;
    for (int i = 0; i < iters; ++i) a = fma(a, b, 1e-9);
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    if (a == 123.456) sink[0] = a;
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int *p; CK(hipMalloc(&p, 1024)); CK(hipMemset(p, 0, 1024));
    double *d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    float ms;
    const int N = 2000;
    for (int grid : {1, 8, 256, 1152}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, p);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("eager  empty grid=%4d : %.2f us/launch\n", grid, ms * 1e3 / N);
    }
    for (int grid : {1, 8, 256, 1152}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, p);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("graph  empty grid=%4d : %.2f us/launch\n", grid, ms * 1e3 / 2000);
    }
    for (int n : {256, 4096, 32768}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_small, dim3(8), dim3(256), 0, st, d, n);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("graph  small(8 blocks, n=%5d doubles, 9 syncs) : %.2f us/launch\n", n, ms * 1e3 / 2000);
    }
    unsigned long long *out; CK(hipMalloc(&out, 16 * 2048));
    for (int grid : {8, 1024}) for (int iters : {2000, 200000}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_clock, dim3(grid), dim3(256), 0, st, out, d, iters);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(2 * grid);
        CK(hipMemcpy(h.data(), out, 16 * grid, hipMemcpyDeviceToHost));
        printf("clock probe grid=%4d iters=%6d : %llu shader cycles / %llu ticks(100MHz) -> %.0f MHz; %.2f cyc/fma\n", grid,
               iters, h[0], h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0, (double)h[0] / iters);
    }
    return 0;
}
