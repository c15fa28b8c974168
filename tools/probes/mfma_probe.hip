// Developer probe: issue rate of v_mfma_f64_16x16x4_f64 (1, 2, 4 accumulators; 1 or 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using d4 = __attribute__((ext_vector_type(4))) double;
template <int NACC>
__global__ __launch_bounds__(256) void k(unsigned long long *out, double *sink, int iters) {
    d4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
    if (s == 123.456) sink[0] = s;
}
template <int NACC> void go(unsigned long long *out, double *sink, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, out, sink, iters);
    hipLaunchKernelGGL((k<NACC>), dim3(blocks), dim3(256), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), out, 8 * blocks, hipMemcpyDeviceToHost);
    printf("NACC=%d blocks=%d : %.1f cycles per MFMA per wave\n", NACC, blocks, (double)h[blocks / 2] / iters / NACC);
}
int main() {
    unsigned long long *out; hipMalloc(&out, 8 * 4096); double *sink; hipMalloc(&sink, 64);
    go<1>(out, sink, 256); go<2>(out, sink, 256); go<4>(out, sink, 256); go<4>(out, sink, 512); go<8>(out, sink, 256);
    return 0;
}
