// Developer probe: cost of a barrier among the workgroups of ONE chain inside a persistent kernel, when those
// workgroups share an XCD (block ids congruent mod 8) and when they do not, against the launch boundary it
// would replace.  Variants of the hand-off:
//   A  agent-scope release (atomic add) / acquire (fence) -- what the memory model asks for across XCDs
//   B  relaxed L2 atomics, s_waitcnt vmcnt(0) before arriving, and agent-scope (L1-bypassing) loads of the data
//      handed over: sufficient only if producer and consumer share an L2, i.e. an XCD
// Every workgroup records its XCC_ID; every round each workgroup checks a value written by another one.
// Build: make -C tools/probes xcd_barrier_probe;  run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// group g of W workgroups; mapping 0: g = L % 8 (same XCD), 1: g = L / W (spread over the XCDs)
__global__ __launch_bounds__(256) void k_persist(int W, int rounds, int mode, int mapping, int work,
                                                 unsigned *cnt, double *slots, double *bulk, unsigned *xcc,
                                                 unsigned long long *ticks, int *err) {
    const int L = blockIdx.x;
    const int g = mapping == 0 ? (L & 7) : L / W, me = mapping == 0 ? (L >> 3) : L % W;
    const int tid = threadIdx.x;
    if (tid == 0) xcc[L] = xcc_id();
    __shared__ int bail;
    if (tid == 0) bail = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        const int par = r & 1;
        // phase work: a slot for the neighbour, and `work` KB of bulk stores per workgroup
        double *myb = bulk + ((size_t)(g * W + me) * 2 + par) * 128 * (size_t)work;
        for (int i = tid; i < 128 * work; i += 256) myb[i] = (double)(r * 7 + i);
        if (tid == 0) slots[((size_t)g * W + me) * 2 + par] = (double)(r * 1000 + me);
        // ---- barrier of the group
        const unsigned target = (unsigned)(r + 1) * (unsigned)W;
        if (mode == 0) {
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(cnt + g * 32, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(cnt + g * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 2000000) { bail = 1; break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        } else {
            __syncthreads();                          // includes s_waitcnt vmcnt(0): this workgroup's stores are in L2
            if (tid == 0) {
                __hip_atomic_fetch_add(cnt + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(cnt + g * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 2000000) { bail = 1; break; }
                }
            }
            __syncthreads();
        }
        if (bail) { if (tid == 0) atomicAdd(err + 1, 1); break; }
        // ---- consume what the neighbour produced (L1-bypassing loads)
        const int nb = (me + 1) % W;
        if (tid == 0) {
            const double v = __hip_atomic_load(slots + ((size_t)g * W + nb) * 2 + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (double)(r * 1000 + nb)) atomicAdd(err, 1);
        }
        const double *nbb = bulk + ((size_t)(g * W + nb) * 2 + par) * 128 * (size_t)work;
        for (int i = tid; i < 128 * work; i += 256) {
            const double v = __hip_atomic_load(nbb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (double)(r * 7 + i)) atomicAdd(err, 1);
            acc += v;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) ticks[L] = t1 - t0;
    if (acc == 1.2345) slots[0] = acc;
}

__global__ void k_phase(int W, int r, double *slots) {      // the launch-per-phase alternative: same tiny work
    const int L = blockIdx.x, g = L & 7, me = L >> 3;
    if (threadIdx.x == 0) slots[((size_t)g * W + me) * 2 + (r & 1)] = (double)(r * 1000 + me);
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int G = 8, rounds = 200;
    unsigned *cnt, *xcc; double *slots, *bulk; unsigned long long *ticks; int *err;
    const int Wmax = 144;
    CK(hipMalloc(&cnt, G * 32 * 4)); CK(hipMalloc(&xcc, G * Wmax * 4)); CK(hipMalloc(&slots, G * Wmax * 2 * 8));
    CK(hipMalloc(&bulk, (size_t)G * Wmax * 2 * 128 * 16 * 8)); CK(hipMalloc(&ticks, G * Wmax * 8)); CK(hipMalloc(&err, 8));
    int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_persist, 256, 0));
    printf("CUs %d, resident 256-thread workgroups per CU %d\n", prop.multiProcessorCount, occ);
    for (int W : {36, 72, 144}) {
        if ((long)G * W > (long)occ * prop.multiProcessorCount) { printf("W=%d: grid does not fit, skipped\n", W); continue; }
        for (int mapping : {0, 1}) for (int mode : {0, 1}) for (int work : {0, 4, 16}) {
            CK(hipMemsetAsync(cnt, 0, G * 32 * 4, st)); CK(hipMemsetAsync(err, 0, 8, st));
            hipLaunchKernelGGL(k_persist, dim3(G * W), dim3(256), 0, st, W, rounds, mode, mapping, work, cnt, slots, bulk, xcc, ticks, err);
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> t(G * W); std::vector<unsigned> x(G * W); int errs[2];
            CK(hipMemcpy(t.data(), ticks, G * W * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(x.data(), xcc, G * W * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(errs, err, 8, hipMemcpyDeviceToHost));
            unsigned long long mx = *std::max_element(t.begin(), t.end());
            // do the groups sit on one XCD each?
            int mixed = 0;
            for (int g = 0; g < G; ++g) {
                unsigned first = 99;
                for (int L = 0; L < G * W; ++L) {
                    const int gg = mapping == 0 ? (L & 7) : L / W;
                    if (gg != g) continue;
                    if (first == 99) first = x[L]; else if (x[L] != first) { ++mixed; break; }
                }
            }
            printf("W=%3d mapping=%s mode=%s work=%2d KB: %.2f us/round  data errors %d  time-outs %d  groups spanning XCDs %d\n",
                   W, mapping == 0 ? "L%8 " : "L/W ", mode == 0 ? "A(rel/acq)" : "B(relaxed)", work,
                   mx * 0.01 / rounds, errs[0], errs[1], mixed);
        }
    }
    // launch-per-phase reference
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int W : {72, 144}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_phase, dim3(G * W), dim3(256), 0, st, W, r, slots);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("W=%3d one launch per phase: %.2f us/phase\n", W, ms * 1e3 / rounds);
    }
    return 0;
}
