// Developer probe: ablation of the S->E gradient kernel (loads / math / reductions).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../covid19uk_amd/csrc -o se_probe se_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cmath>
#include "logprob_kernels.h"
using namespace seir;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0 full, 1 loads only, 2 math only (no plane loads), 3 full without row/col reductions
template <int MODE, int RW>
__global__ __launch_bounds__(256) void k_var(Dims d, Consts c, Work w) {
    __shared__ double colbuf[4][WAVE];
    __shared__ double shl[4], shp[4];
    __shared__ double2 ltab[LOGTAB_N];
    log_table_to_lds(ltab, c.logtab);
    const int b = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * WAVE + lane;
    const int m0 = blockIdx.y * (4 * RW) + wave * RW;
    const double psi = w.scal[(size_t)b * NSCAL + SC_PSI];
    const double ea_t = w.ea[(size_t)b * d.Tp + t];
    const double Wt = c.W[t];
    const double psiW = psi * Wt;
    const size_t q0 = ((size_t)b * d.Mp + m0) * d.Tp + t;
    double F[RW], I[RW], kse[RW], snk[RW], eb[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const size_t q = q0 + (size_t)r * d.Tp;
        if (MODE != 2) {
            F[r] = w.F[q];
            eb[r] = w.eb[(size_t)b * d.Mp + m0 + r];
            const int ki = w.K[0][q];
            I[r] = (double)w.St[2][q];
            kse[r] = (double)ki; snk[r] = (double)(w.St[0][q] - ki);
        } else {
            F[r] = 1e-3 * lane + r; eb[r] = 1e-6 * (r + 1); I[r] = 10.0 + lane; kse[r] = (lane + r) & 3; snk[r] = 1e5 + lane;
        }
    }
    double ll = 0.0, gpsi = 0.0, colacc = 0.0, rowacc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        if (MODE == 1) { ll += F[r] + I[r] + kse[r] + snk[r] + eb[r]; rowacc[r] = 0; continue; }
        const double ee = ea_t * eb[r];
        const double lam0 = ee * (I[r] + psiW * F[r]);
        const double rr = (lam0 + d.rate_floor) * d.dt;
        double L, inv;
        l1me_inv(rr, L, inv, ltab);
        const bool has = kse[r] != 0.0;
        ll += (has ? kse[r] * L : 0.0) - snk[r] * rr;
        const double gl = d.dt * ((has ? kse[r] * inv : 0.0) - snk[r]);
        const double ge = gl * lam0;
        rowacc[r] = ge;
        colacc += ge;
        gpsi += gl * ee * Wt * F[r];
    }
    ll = wave_sum(ll);
    gpsi = wave_sum(gpsi);
    if (MODE == 0 || MODE == 2) {
        colbuf[wave][lane] = colacc;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const double v = wave_sum(rowacc[r]);
            if (lane == 0) w.Rpart[((size_t)b * d.ntc + blockIdx.x) * d.Mp + m0 + r] = v;
        }
    } else {
        double s = colacc;
#pragma unroll
        for (int r = 0; r < RW; ++r) s += rowacc[r];
        ll += s * 1e-300;
    }
    if (lane == 0) { shl[wave] = ll; shp[wave] = gpsi; }
    __syncthreads();
    const size_t tile = (size_t)b * gridDim.y * gridDim.x + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) { w.Lpart[tile] = shl[0] + shl[1] + shl[2] + shl[3]; w.Ppart[tile] = shp[0] + shp[1] + shp[2] + shp[3]; }
    if ((MODE == 0 || MODE == 2) && threadIdx.x < WAVE)
        w.Kpart[((size_t)b * gridDim.y + blockIdx.y) * d.Tp + t] = colbuf[0][lane] + colbuf[1][lane] + colbuf[2][lane] + colbuf[3][lane];
}

template <int MODE, int RW>
static float run(Dims d, Consts c, Work w, int B, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    dim3 grid(d.Tp / 64, d.Mp / (4 * RW), B);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_var<MODE, RW>), grid, dim3(256), 0, st, d, c, w);
    hipEventRecord(e0, st);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k_var<MODE, RW>), grid, dim3(256), 0, st, d, c, w);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / 200;
}

int main() {
    const int B = 8, M = 380, T = 365;
    Dims d{}; d.M = M; d.T = T; d.Mp = 384; d.Tp = 384; d.Kp = 380; d.P = 750; d.Pp = 750; d.ntc = 6; d.nmt = 12;
    d.dt = 1.0; d.rate_floor = 1e-9; d.nu = 0.28;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t cells = (size_t)B * d.Mp * d.Tp;
    Consts c{}; Work w{};
    std::vector<double> hF(cells), hW(d.Tp, 1.0), hea(B * d.Tp), heb(B * d.Mp), hsc(B * NSCAL, 0.5);
    std::vector<int> hK(cells), hS(cells), hI(cells);
    srand(1);
    for (size_t i = 0; i < cells; ++i) { hF[i] = (rand() % 1000) * 1e-3; hK[i] = rand() % 50; hS[i] = 100000 + rand() % 1000; hI[i] = rand() % 500; }
    for (auto &x : hea) x = 0.25; for (auto &x : heb) x = 1e-5;
    std::vector<double2> lt(LOGTAB_N);
    for (int i = 0; i < LOGTAB_N; ++i) { long double cc = 1.0L + (i + 0.5L) / LOGTAB_N; lt[i].x = (double)(1.0L / cc); lt[i].y = (double)(-logl((long double)lt[i].x)); }
    double2 *dlt; CK(hipMalloc(&dlt, sizeof(double2) * LOGTAB_N)); CK(hipMemcpy(dlt, lt.data(), sizeof(double2) * LOGTAB_N, hipMemcpyHostToDevice)); c.logtab = dlt;
    double *dW; CK(hipMalloc(&dW, d.Tp * 8)); CK(hipMemcpy(dW, hW.data(), d.Tp * 8, hipMemcpyHostToDevice)); c.W = dW;
    CK(hipMalloc(&w.F, cells * 8)); CK(hipMemcpy(w.F, hF.data(), cells * 8, hipMemcpyHostToDevice));
    for (int x = 0; x < 3; ++x) { CK(hipMalloc(&w.K[x], cells * 4)); CK(hipMalloc(&w.St[x], cells * 4)); }
    CK(hipMemcpy(w.K[0], hK.data(), cells * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w.St[0], hS.data(), cells * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w.St[2], hI.data(), cells * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&w.ea, B * d.Tp * 8)); CK(hipMemcpy(w.ea, hea.data(), B * d.Tp * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&w.eb, B * d.Mp * 8)); CK(hipMemcpy(w.eb, heb.data(), B * d.Mp * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&w.scal, B * NSCAL * 8)); CK(hipMemcpy(w.scal, hsc.data(), B * NSCAL * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&w.Lpart, B * 1024 * 8)); CK(hipMalloc(&w.Ppart, B * 1024 * 8));
    CK(hipMalloc(&w.Kpart, (size_t)B * 96 * d.Tp * 8)); CK(hipMalloc(&w.Rpart, (size_t)B * 6 * d.Mp * 8));
    printf("cells=%zu bytes/launch=%.1f MB\n", cells, cells * 20 / 1e6);
#define ROW(RW) printf("RW=%2d  full %.2f  loads-only %.2f  math-only %.2f  no-rowcol-reduce %.2f us\n", RW, \
        run<0, RW>(d, c, w, B, st, e0, e1), run<1, RW>(d, c, w, B, st, e0, e1), run<2, RW>(d, c, w, B, st, e0, e1), run<3, RW>(d, c, w, B, st, e0, e1))
    ROW(2); ROW(4); ROW(8); ROW(16);
    return 0;
}
