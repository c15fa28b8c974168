// libseirhip.so -- C-ABI (include/seir_hip.h) over the gfx950 kernels.
// Host side of the drop-in boundary for joint_log_prob
// (covid19uk/inference/inference.py:537-557).  No CPU fallback anywhere in
// this file: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/seir_hip.h"
#include "sampler_kernels.h"
#include "sim_kernels.h"
#include "moves_kernel.h"
#include "rt_kernels.h"

using namespace seir;

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(SEIR_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                              \
    } while (0)

struct seir_ctx {
    Dims d{};
    Consts c{};
    Work w{};
    int Bmax = 0, device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<void *> allocs;
    // staging for the host-pointer entry points
    double *u_stage = nullptr, *ev_stage = nullptr, *logp_stage = nullptr, *grad_stage = nullptr;
    // arguments of the last evaluation (seir_time_kernel replays them)
    const double *last_u = nullptr, *last_events = nullptr;
    double *last_logp = nullptr, *last_grad = nullptr;
    bool prepared = false;
    int opt_skew = 0, opt_affinity = 3;     // seir_set_option
    int opt_gemm_f32 = 0;
    int opt_eval_form = 0;            // 0 auto (one launch where a chain's blocks share an XCD, else three), 1 four launches, 2 three
    int xcd_local = -1;               // -1 not probed yet; 1: blocks with the same id mod 8 share an XCD, eight different ones
    unsigned long long *eval_cnt = nullptr;   // [8][EVC_STRIDE] k_eval_all's counters
    int *eval_err = nullptr;          // k_eval_all: waits that timed out
    unsigned long long eval_a = 0, eval_b = 0;   // what a chain's counters A and B show after the launches so far
    int eval_nb = -1;                            // batch size of those launches (another size: counters and targets start over)
    int eval_slots[2][2] = {{-1, -1}, {-1, -1}};  // workgroups of k_eval_all<GRAD, TN> the chip holds at once (occupancy API; [TN == 96][GRAD])
    std::vector<float> cstar32_host;        // fp32 copy of the padded Cstar, uploaded when the option is first set
};

static inline int ceil_to(int x, int q) { return (x + q - 1) / q * q; }

// hipFuncSetAttribute applies to the current device's copy of a kernel: "done once" is kept per device (a bit per
// ordinal), not per process
static bool first_on_device(std::atomic<unsigned long long> &mask, int device) {
    const unsigned long long bit = 1ull << (device & 63);
    return (mask.fetch_or(bit) & bit) == 0ull;
}

template <typename T>
static int dev_alloc(seir_ctx *ctx, T **p, size_t count, bool zero = true) {
    void *q = nullptr;
    const size_t bytes = count * sizeof(T);
    HIP_TRY(hipMalloc(&q, bytes ? bytes : sizeof(T)));
    ctx->allocs.push_back(q);
    if (zero) HIP_TRY(hipMemset(q, 0, bytes ? bytes : sizeof(T)));
    *p = (T *)q;
    return 0;
}

template <typename T>
static int dev_upload(seir_ctx *ctx, const T **p, const std::vector<T> &h) {
    T *q = nullptr;
    int rc = dev_alloc(ctx, &q, h.size(), false);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(q, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *p = q;
    return 0;
}

extern "C" int seir_abi_version(void) { return SEIR_ABI_VERSION; }
extern "C" const char *seir_last_error(void) { return g_err; }

static void release_eval_all(seir_ctx *ctx);
extern "C" void seir_destroy(seir_ctx *ctx) {
    if (!ctx) return;
    release_eval_all(ctx);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (void *p : ctx->allocs) (void)hipFree(p);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int create_impl(const seir_desc *ds, seir_ctx *ctx) {
    const int M = ds->M, T = ds->T, B = ds->max_chains;
    Dims &d = ctx->d;
    d.M = M; d.T = T;
    d.Mp = ceil_to(M, 64);
    d.Tp = ceil_to(T, 64);
    d.Kp = ceil_to(M, 4);
    d.Kp0 = d.Mp;
    d.P = 6 + (T - 1) + M;
    d.Pp = d.P;
    d.b0 = 0;
    d.nrb_scan = (M + SCAN_ROWS - 1) / SCAN_ROWS;
    d.nmt = d.Mp / SE_TM;
    d.ntc = d.Tp / 64;
    d.nu = ds->nu; d.dt = ds->time_delta; d.rate_floor = ds->rate_floor;
    d.car_half_logdet = ds->car_half_logdet;
    {   // constants of the prior log-densities, model_spec.py:140-198 (TFP formulas)
        const double L2PI = 1.8378770664093453;
        d.prior_const = (-std::log(10.0) - 0.5 * L2PI)                       // alpha_0
                        + (-0.5 * L2PI)                                        // beta_area
                        + (3.0 * std::log(10.0) - std::lgamma(3.0))            // psi
                        - (T - 1) * (std::log(0.005) + 0.5 * L2PI)             // alpha_t
                        + (0.5 * std::log(2.0 / M_PI) - std::log(0.1))         // sigma_space
                        + (ds->car_half_logdet - 0.5 * M * L2PI)               // spatial_effect
                        + 2.0 * (-std::log(100.0) - 0.5 * L2PI);               // gamma0, gamma1
    }
    d.L_ei = std::log(-std::expm1(-ds->nu * ds->time_delta));
    ctx->Bmax = B;
    ctx->device = ds->device;

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ds->device < 0 || ds->device >= ndev)
        return fail(SEIR_ERR_DEVICE, "device %d not present (%d HIP devices visible)", ds->device, ndev);
    HIP_TRY(hipSetDevice(ds->device));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&ctx->ev0));
    HIP_TRY(hipEventCreate(&ctx->ev1));

    // log-factorial table
    double lf[LFACT_TABLE];
    for (int i = 0; i < LFACT_TABLE; ++i) lf[i] = std::lgamma((double)i + 1.0);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_lfact), lf, sizeof(lf)));

    // padded constants
    std::vector<double> Cs((size_t)d.Mp * d.Kp0, 0.0), N(d.Mp, 1.0), invN(d.Mp, 0.0), la(d.Mp, 0.0),
        W(d.Tp, 0.0), wd(d.Tp, 0.0), init((size_t)d.Mp * 4, 0.0);
    for (int m = 0; m < M; ++m) {
        for (int j = 0; j < M; ++j) Cs[(size_t)m * d.Kp0 + j] = ds->Cstar[(size_t)m * M + j];
        if (!(ds->N[m] > 0.0)) return fail(SEIR_ERR_INVALID, "N[%d] must be positive", m);
        N[m] = ds->N[m];
        invN[m] = 1.0 / ds->N[m];
        la[m] = ds->log_area_c[m];
        for (int s = 0; s < 4; ++s) init[(size_t)m * 4 + s] = ds->init_state[(size_t)m * 4 + s];
    }
    for (int m = 0; m < M; ++m)          // C + C^T with a diagonal is symmetric (model_spec.py:216-219); k_gemm relies on it
        for (int j = 0; j < m; ++j)
            if (ds->Cstar[(size_t)m * M + j] != ds->Cstar[(size_t)j * M + m])
                return fail(SEIR_ERR_INVALID, "Cstar must be symmetric (entry %d,%d)", m, j);
    for (int t = 0; t < T; ++t) { W[t] = ds->W[t]; wd[t] = ds->weekday_c[t]; }
    std::vector<int> qrow(M + 1, 0), qcol;
    std::vector<double> qval;
    for (int m = 0; m < M; ++m) {
        for (int j = 0; j < M; ++j) {
            const double v = ds->car_Q[(size_t)m * M + j];
            if (v != 0.0) { qcol.push_back(j); qval.push_back(v); }
        }
        qrow[m + 1] = (int)qcol.size();
    }
    if (qcol.empty()) { qcol.push_back(0); qval.push_back(0.0); }
    // table of device_math.h fast_log: c_i = 1 + (i + 1/2)/128, (fl(1/c_i), -log(fl(1/c_i)))
    std::vector<double2> ltab(LDSTAB_N);
    for (int i = 0; i < LFACT_TABLE / 2; ++i) {
        ltab[LOGTAB_N + i].x = lf[2 * i];
        ltab[LOGTAB_N + i].y = lf[2 * i + 1];
    }
    for (int i = 0; i < LOGTAB_N; ++i) {
        const long double cc = 1.0L + ((long double)i + 0.5L) / (long double)LOGTAB_N;
        const double invc = (double)(1.0L / cc);
        ltab[i].x = invc;
        ltab[i].y = (double)(-logl((long double)invc));
    }
    int rc;
    if ((rc = dev_upload(ctx, &ctx->c.logtab, ltab))) return rc;
    {
        std::vector<double> lfb(SCAN_LFT);
        for (int i = 0; i < SCAN_LFT; ++i) lfb[i] = std::lgamma((double)i + 1.0);
        if ((rc = dev_upload(ctx, &ctx->c.lfact_big, lfb))) return rc;
    }
    {   // ELL copy of car_Q (adjacency rows are short): [k][m] so that a wave reads coalesced
        int qw = 0;
        for (int m = 0; m < M; ++m) qw = std::max(qw, qrow[m + 1] - qrow[m]);
        if (qw <= 32) {
            std::vector<int> ec((size_t)std::max(qw, 1) * d.Mp, 0);
            std::vector<double> ev((size_t)std::max(qw, 1) * d.Mp, 0.0);
            for (int m = 0; m < M; ++m)
                for (int e = qrow[m], k = 0; e < qrow[m + 1]; ++e, ++k) {
                    ec[(size_t)k * d.Mp + m] = qcol[e];
                    ev[(size_t)k * d.Mp + m] = qval[e];
                }
            ctx->c.qw = qw;
            if ((rc = dev_upload(ctx, &ctx->c.Qell_col, ec))) return rc;
            if ((rc = dev_upload(ctx, &ctx->c.Qell_val, ev))) return rc;
        } else {
            ctx->c.qw = 0;
        }
    }
    if ((rc = dev_upload(ctx, &ctx->c.Cstar, Cs))) return rc;
    ctx->cstar32_host.assign(Cs.begin(), Cs.end());             // rounded to fp32; goes to the device only if asked for
    if ((rc = dev_upload(ctx, &ctx->c.N, N))) return rc;
    if ((rc = dev_upload(ctx, &ctx->c.invN, invN))) return rc;
    if ((rc = dev_upload(ctx, &ctx->c.la, la))) return rc;
    if ((rc = dev_upload(ctx, &ctx->c.W, W))) return rc;
    if ((rc = dev_upload(ctx, &ctx->c.wd, wd))) return rc;
    if ((rc = dev_upload(ctx, &ctx->c.init, init))) return rc;
    if ((rc = dev_upload(ctx, &ctx->c.Qrow, qrow))) return rc;
    if ((rc = dev_upload(ctx, &ctx->c.Qcol, qcol))) return rc;
    if ((rc = dev_upload(ctx, &ctx->c.Qval, qval))) return rc;

    Work &w = ctx->w;
    const size_t cells = (size_t)B * d.Mp * d.Tp;
    if ((rc = dev_alloc(ctx, &w.Xn, cells))) return rc;
    if ((rc = dev_alloc(ctx, &w.F, cells))) return rc;
    if ((rc = dev_alloc(ctx, &w.KS, cells))) return rc;
    if ((rc = dev_alloc(ctx, &w.rowconst, (size_t)B * d.Mp))) return rc;
    if ((rc = dev_alloc(ctx, &w.colIR, (size_t)B * d.nrb_scan * d.Tp * 2))) return rc;
    if ((rc = dev_alloc(ctx, &w.ea, (size_t)B * d.Tp))) return rc;
    if ((rc = dev_alloc(ctx, &w.eb, (size_t)B * d.Mp))) return rc;
    if ((rc = dev_alloc(ctx, &w.rir, (size_t)B * d.Tp))) return rc;
    if ((rc = dev_alloc(ctx, &w.scal, (size_t)B * NSCAL))) return rc;
    if ((rc = dev_alloc(ctx, &w.Qs, (size_t)B * d.Mp))) return rc;
    if ((rc = dev_alloc(ctx, &w.Kir, (size_t)B * d.Tp))) return rc;
    if ((rc = dev_alloc(ctx, &w.Dir, (size_t)B * d.Tp))) return rc;
    if ((rc = dev_alloc(ctx, &w.constsum, (size_t)B))) return rc;
    if ((rc = dev_alloc(ctx, &w.Lpart, (size_t)B * d.nmt * d.ntc))) return rc;
    if ((rc = dev_alloc(ctx, &w.Ppart, (size_t)B * d.nmt * d.ntc))) return rc;
    if ((rc = dev_alloc(ctx, &w.Kpart, (size_t)B * d.nmt * d.Tp))) return rc;
    if ((rc = dev_alloc(ctx, &w.Rpart, (size_t)B * d.ntc * d.Mp))) return rc;

    if ((rc = dev_alloc(ctx, &ctx->u_stage, (size_t)B * d.P))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->ev_stage, (size_t)B * M * T * 3))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->logp_stage, (size_t)B))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->grad_stage, (size_t)B * d.P))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" int seir_create(const seir_desc *ds, seir_ctx **out) {
    if (!ds || !out) return fail(SEIR_ERR_INVALID, "null argument");
    *out = nullptr;
    if (ds->M < 1 || ds->T < 1 || ds->max_chains < 1)
        return fail(SEIR_ERR_INVALID, "M, T and max_chains must be >= 1 (got %d, %d, %d)", ds->M, ds->T,
                    ds->max_chains);
    if (ds->T > 2048 || ds->M > 2048)
        return fail(SEIR_ERR_INVALID, "M=%d, T=%d exceed the supported 2048 x 2048", ds->M, ds->T);
    if (!ds->Cstar || !ds->N || !ds->W || !ds->weekday_c || !ds->log_area_c || !ds->car_Q || !ds->init_state)
        return fail(SEIR_ERR_INVALID, "null covariate pointer");
    if (!(ds->time_delta > 0.0) || !(ds->nu > 0.0))
        return fail(SEIR_ERR_INVALID, "nu and time_delta must be positive");
    seir_ctx *ctx = new (std::nothrow) seir_ctx();
    if (!ctx) return fail(SEIR_ERR_DEVICE, "out of host memory");
    int rc = create_impl(ds, ctx);
    if (rc) { seir_destroy(ctx); return rc; }
    *out = ctx;
    return 0;
}

extern "C" int seir_num_params(const seir_ctx *ctx) { return ctx ? ctx->d.P : SEIR_ERR_INVALID; }

extern "C" int seir_set_initial_state(seir_ctx *ctx, const double *init_state) {
    if (!ctx || !init_state) return fail(SEIR_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int M = ctx->d.M;
    for (int i = 0; i < 4 * M; ++i)
        if (!(init_state[i] >= 0.0) || init_state[i] != std::floor(init_state[i]))
            return fail(SEIR_ERR_INVALID, "init_state[%d]=%g is not a non-negative integer count", i, init_state[i]);
    // the padded rows [M, Mp) stay zero; blocking copy: the caller's buffer is not retained
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(const_cast<double *>(ctx->c.init), init_state, sizeof(double) * 4 * M, hipMemcpyHostToDevice));
    ctx->prepared = false;
    return 0;
}

static int check_batch(seir_ctx *ctx, int B) {
    if (!ctx) return fail(SEIR_ERR_INVALID, "null context");
    if (B < 1 || B > ctx->Bmax) return fail(SEIR_ERR_INVALID, "B=%d outside [1, max_chains=%d]", B, ctx->Bmax);
    HIP_TRY(hipSetDevice(ctx->device));
    return 0;
}

// --- individual launches ---------------------------------------------------
// `d.b0` selects the first chain, `nb` the number of chains, `st` the stream.
// `affinity`: give every block of a chain the same (block id % 8), see xcd_affine()
struct LaunchCfg { Dims d; hipStream_t st; int nb; int affinity; };
// affinity: bit 0 = gradient kernel, bit 1 = event-move kernels (seir_set_option, default 3)
static LaunchCfg whole(seir_ctx *ctx, int B) {
    LaunchCfg l{ctx->d, ctx->stream, B, ctx->opt_affinity};
    l.d.skew = ctx->opt_skew;                           // test hook, see debug_skew()
    return l;
}

extern "C" int seir_set_option(seir_ctx *ctx, int32_t option, int32_t value) {
    if (!ctx) return fail(SEIR_ERR_INVALID, "null context");
    switch (option) {
        case SEIR_OPT_DEBUG_SKEW:
            if (value < 0 || value > 3) return fail(SEIR_ERR_INVALID, "debug skew must be 0..3");
            ctx->opt_skew = value;
            return 0;
        case SEIR_OPT_XCD_AFFINITY:
            if (value < 0 || value > 3) return fail(SEIR_ERR_INVALID, "xcd affinity is a 2-bit mask");
            ctx->opt_affinity = value;
            return 0;
        case SEIR_OPT_EVAL_FORM:
            if (value < 0 || value > 2) return fail(SEIR_ERR_INVALID, "eval form is 0 (auto), 1 (four launches) or 2 (three launches)");
            ctx->opt_eval_form = value;
            return 0;
        case SEIR_OPT_GEMM_F32: {
            if (value < 0 || value > 1) return fail(SEIR_ERR_INVALID, "gemm_f32 is 0 or 1");
            if (value && !ctx->c.Cstar32) {
                HIP_TRY(hipSetDevice(ctx->device));
                int rc = dev_upload(ctx, &ctx->c.Cstar32, ctx->cstar32_host);
                if (rc) return rc;
                if ((rc = dev_alloc(ctx, &ctx->w.Xn32, (size_t)ctx->Bmax * ctx->d.Mp * ctx->d.Tp))) return rc;
            }
            if (value && (ctx->d.Mp % GF_T != 0 || ctx->d.Tp % GF_T != 0))
                return fail(SEIR_ERR_INVALID, "the fp32 contraction needs ceil64(M) and ceil64(T) to be multiples of %d (M=%d, T=%d)",
                            GF_T, ctx->d.M, ctx->d.T);
            ctx->opt_gemm_f32 = value;
            ctx->prepared = false;
            return 0;
        }
        default:
            return fail(SEIR_ERR_INVALID, "unknown option %d", option);
    }
}

template <int SRC>
static void launch_scan(seir_ctx *ctx, const LaunchCfg &l, const double *events) {
    const Dims &d = l.d;
    const size_t lds = ((size_t)SCAN_WAVES * d.Tp * 2 + SCAN_LFT) * sizeof(double);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)k_scan<SRC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_scan<SRC>, dim3(d.nrb_scan, l.nb), dim3(SCAN_WAVES * WAVE), lds,
                       l.st, d, ctx->c, ctx->w, events);
}
static void launch_colreduce(seir_ctx *ctx, const LaunchCfg &l) {
    hipLaunchKernelGGL(k_colreduce, dim3(l.d.Tp / WAVE, l.nb), dim3(256), 0, l.st, l.d, ctx->w);
}
template <int TN>
static void launch_gemm_t(seir_ctx *ctx, const LaunchCfg &l) {
    const Dims &d = l.d;
    const size_t lds = gemm_lds_bytes<TN>();
    static std::atomic<unsigned long long> attr_set{0ull};
    if (lds > 64 * 1024 && first_on_device(attr_set, ctx->device))
        (void)hipFuncSetAttribute((const void *)k_gemm<TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_gemm<TN>), dim3(d.Tp / TN, d.Mp / GEMM_TM, l.nb), dim3(gemm_threads<TN>()), lds, l.st, d, ctx->c,
                       ctx->w);
}
#ifndef GEMM_NCG
#define GEMM_NCG 3
#endif
static void launch_gemm(seir_ctx *ctx, const LaunchCfg &l) {
    if (ctx->opt_gemm_f32 && ctx->c.Cstar32 && ctx->w.Xn32 && l.d.Mp % GF_T == 0 && l.d.Tp % GF_T == 0) {
        hipLaunchKernelGGL(k_gemm_f32, dim3(l.d.Tp / GF_T, l.d.Mp / GF_T, l.nb), dim3(256), 0, l.st, l.d, ctx->c, ctx->w);
        return;
    }
    // 64 x 96 tiles only where they turn two rounds of workgroups into one (UK-380, 8 chains: 288 -> 192 on
    // 256 CUs, 34.6 -> 32.9 us); on large grids the 6-wave tile loses to the 4-wave one (SYN-2048: 38.6 vs 51.7 TF)
    const Dims &d = l.d;
    const long t64 = (long)(d.Tp / 64) * (d.Mp / GEMM_TM) * l.nb, t96 = (long)(d.Tp / 96) * (d.Mp / GEMM_TM) * l.nb;
    if (d.Tp % 96 == 0 && t64 > 256 && t96 <= 256) {
        // the eight-wave form of the 64 x 96 tile: every SIMD of a CU carries two waves (k_gemm<96>'s six waves: 2,2,1,1)
        const size_t lds = gemm_lds_bytes<96>();
        static std::atomic<unsigned long long> attr_set{0ull};
        if (lds > 64 * 1024 && first_on_device(attr_set, ctx->device))
            (void)hipFuncSetAttribute((const void *)k_gemm_w8<GEMM_NCG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_gemm_w8<GEMM_NCG>, dim3(d.Tp / 96, d.Mp / GEMM_TM, l.nb), dim3(256 * GEMM_NCG), lds, l.st, d, ctx->c, ctx->w);
    } else {
        launch_gemm_t<64>(ctx, l);
    }
}
static void launch_params(seir_ctx *ctx, const LaunchCfg &l, const double *u) {
    hipLaunchKernelGGL(k_params, dim3(l.nb), dim3(256), 0, l.st, l.d, ctx->c, ctx->w, u);
}
template <int SRC>
static void launch_se(seir_ctx *ctx, const LaunchCfg &l, bool grad) {
    Dims d = l.d;
    const bool affinity = (l.affinity & 1) && xcd_affinity_applies(d.ntc * d.nmt, l.nb);
    d.aff_nb = affinity ? l.nb : 0;
    const dim3 grid = affinity ? dim3(d.ntc * d.nmt * l.nb) : dim3(d.ntc, d.nmt, l.nb);
    if (grad && SRC == 1 && d.chunked == 1)
        hipLaunchKernelGGL((k_se<true, SRC, 1>), grid, dim3(256), 0, l.st, d, ctx->c, ctx->w);
    else if (grad && SRC == 1 && d.chunked == 2)
        hipLaunchKernelGGL((k_se<true, SRC, 2>), grid, dim3(256), 0, l.st, d, ctx->c, ctx->w);
    else if (grad)
        hipLaunchKernelGGL((k_se<true, SRC>), grid, dim3(256), 0, l.st, d, ctx->c, ctx->w);
    else
        hipLaunchKernelGGL((k_se<false, SRC>), grid, dim3(256), 0, l.st, d, ctx->c, ctx->w);
}
static void launch_finish(seir_ctx *ctx, const LaunchCfg &l, const double *u, double *logp, double *grad) {
    const size_t lds = (size_t)l.d.Tp * sizeof(double);
    if (grad)
        hipLaunchKernelGGL(k_finish<true>, dim3(l.nb), dim3(256), lds, l.st, l.d, ctx->c, ctx->w, u, logp, grad, 0);
    else
        hipLaunchKernelGGL(k_finish<false>, dim3(l.nb), dim3(256), lds, l.st, l.d, ctx->c, ctx->w, u, logp, grad, 0);
}

extern "C" int seir_prepare_events_dev(seir_ctx *ctx, int32_t B, const double *events_dev) {
    int rc = check_batch(ctx, B);
    if (rc) return rc;
    if (!events_dev) return fail(SEIR_ERR_INVALID, "null events pointer");
    launch_scan<0>(ctx, whole(ctx, B), events_dev);
    launch_colreduce(ctx, whole(ctx, B));
    launch_gemm(ctx, whole(ctx, B));
    HIP_TRY(hipGetLastError());
    ctx->last_events = events_dev;
    ctx->prepared = true;
    return 0;
}

extern "C" int seir_eval_prepared_dev(seir_ctx *ctx, int32_t B, const double *u_dev, double *logp_dev,
                                      double *grad_dev) {
    int rc = check_batch(ctx, B);
    if (rc) return rc;
    if (!ctx->prepared) return fail(SEIR_ERR_STATE, "seir_prepare_events_dev has not been called");
    if (!u_dev || !logp_dev) return fail(SEIR_ERR_INVALID, "null u/logp pointer");
    launch_params(ctx, whole(ctx, B), u_dev);
    launch_se<0>(ctx, whole(ctx, B), grad_dev != nullptr);
    launch_finish(ctx, whole(ctx, B), u_dev, logp_dev, grad_dev);
    HIP_TRY(hipGetLastError());
    ctx->last_u = u_dev; ctx->last_logp = logp_dev; ctx->last_grad = grad_dev;
    return 0;
}

// The fused form of the full evaluation (default): [state part of the scan | parameter tables],
// [contraction tiles with the S->E term as epilogue | row constants | fold of the I->R partials], reduction.
static void launch_state_params(seir_ctx *ctx, const LaunchCfg &l, const double *u_dev, const double *events_dev) {
    const Dims &d = l.d;
    const size_t lds_a = (size_t)SCAN_WAVES * d.Tp * 2 * sizeof(double);
    static std::atomic<unsigned long long> attr_a{0ull};
    if (lds_a > 64 * 1024 && first_on_device(attr_a, ctx->device))
        (void)hipFuncSetAttribute((const void *)k_state_params, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
    hipLaunchKernelGGL(k_state_params, dim3(d.nrb_scan + 1, l.nb), dim3(SCAN_WAVES * WAVE), lds_a, l.st, d, ctx->c, ctx->w,
                       events_dev, u_dev);
}
// the Dims of the tile launch and of the reduction after it: partial sums per 64 x TN tile
template <int TN>
static Dims fused_dims(const LaunchCfg &l) {
    Dims d = l.d;
    d.nmt = d.Mp / GEMM_TM;
    d.ntc = d.Tp / TN;
    return d;
}
// Do blocks with the same id mod 8 share an XCD on this GPU, eight different ones for the eight classes?  (XCC_ID of a
// probe grid; the XCD-local hand-offs inside a launch -- k_eval_all here, k_se_chunk and the band workgroups of
// k_move_pair in the sampler -- are used only then.)
static bool probe_xcd_local(hipStream_t st) {
    const int nblk = 8 * 144;
    unsigned *xcc = nullptr;
    std::vector<unsigned> host(nblk, 99u);
    hipError_t e = hipMalloc((void **)&xcc, nblk * sizeof(unsigned));
    if (e != hipSuccess) return false;
    hipLaunchKernelGGL(k_xcc_probe, dim3(nblk), dim3(256), 0, st, xcc);
    e = hipMemcpyAsync(host.data(), xcc, nblk * sizeof(unsigned), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(xcc);
    bool ok = e == hipSuccess;
    for (int L = 8; L < nblk && ok; ++L) ok = host[L] == host[L & 7] && host[L] < 16u;
    for (int a = 0; a < 8 && ok; ++a)
        for (int b2 = a + 1; b2 < 8; ++b2) ok = ok && host[a] != host[b2];
    return ok;
}

// which context may use the one-launch evaluation on each device (see seir_log_prob_dev)
static std::atomic<seir_ctx *> g_eval_all_owner[64];
static bool claim_eval_all(seir_ctx *ctx) {
    std::atomic<seir_ctx *> &slot = g_eval_all_owner[ctx->device & 63];
    seir_ctx *cur = slot.load();
    if (cur == ctx) return true;
    if (cur != nullptr) return false;
    return slot.compare_exchange_strong(cur, ctx) || cur == ctx;
}
static void release_eval_all(seir_ctx *ctx) {
    seir_ctx *me = ctx;
    (void)g_eval_all_owner[ctx->device & 63].compare_exchange_strong(me, nullptr);
}

template <int TN>
static void launch_finish_fused(seir_ctx *ctx, const LaunchCfg &l, const double *u_dev, double *logp_dev, double *grad_dev);
// The whole evaluation in one launch (k_eval_all): 8 chains, XCD-affine block ids, the GPU's XCD placement checked.
template <int TN>
static int launch_eval_all(seir_ctx *ctx, const LaunchCfg &l, const double *u_dev, const double *events_dev,
                           double *logp_dev, double *grad_dev) {
    const int nbv = (l.nb + 7) / 8 * 8;
    if (!ctx->eval_cnt) {
        int rc = dev_alloc(ctx, &ctx->eval_cnt, (size_t)((ctx->Bmax + 7) / 8 * 8) * EVC_STRIDE + 16);
        if (!rc) rc = dev_alloc(ctx, &ctx->eval_err, 1);
        if (rc) return rc;
    }
    Dims d = fused_dims<TN>(l);
    d.aff_nb = nbv;
    d.nlive = nbv != l.nb ? l.nb : 0;
    const int per = d.ntc * d.nmt, ncb = d.Tp / WAVE;
    if (ctx->eval_nb != l.nb) {
        // the counters are per chain and the targets one running total per context: a batch of another size leaves the
        // chains that sat out behind the target (they would wait for the time-out and go on without their producers'
        // data) -- start all of them from zero again, in stream order
        if (hipMemsetAsync(ctx->eval_cnt, 0, sizeof(unsigned long long) * ((size_t)((ctx->Bmax + 7) / 8 * 8) * EVC_STRIDE + 16),
                           l.st) != hipSuccess)
            return fail(SEIR_ERR_DEVICE, "resetting the evaluation hand-off counters failed");
        ctx->eval_a = ctx->eval_b = 0;
        ctx->eval_nb = l.nb;
    }
    ctx->eval_a += (unsigned long long)(per + 1);                       // per chain: the tiles' state parts + the parameter block
    ctx->eval_b += (unsigned long long)(per + d.nrb_scan + ncb);        //            tiles + row-constant blocks + I->R fold blocks
    const size_t lds = eval_all_lds_bytes<TN>(d);
    static std::atomic<unsigned long long> attr{0ull};
    if (lds > 64 * 1024 && first_on_device(attr, ctx->device)) {
        (void)hipFuncSetAttribute((const void *)k_eval_all<true, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)k_eval_all<false, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const dim3 grid((unsigned)((1 + per + d.nrb_scan + ncb + 1) * nbv));
    // the reduction block runs inside the launch for value-only calls; with the gradient it is its own launch (measured:
    // the gradient assembly takes 12.7 us as the last block of this large kernel against 4.8 us in k_finish -- 59.2 us
    // per batch against 56.9)
#ifndef EVAL_FIN_GRAD
#define EVAL_FIN_GRAD 0
#endif
    const int fin = grad_dev ? EVAL_FIN_GRAD : 1;
    if (grad_dev)
        hipLaunchKernelGGL((k_eval_all<true, TN>), grid, dim3(512), lds, l.st, d, ctx->c, ctx->w, events_dev, u_dev, logp_dev,
                           grad_dev, ctx->eval_cnt, ctx->eval_a, ctx->eval_b, ctx->eval_err, fin);
    else
        hipLaunchKernelGGL((k_eval_all<false, TN>), grid, dim3(512), lds, l.st, d, ctx->c, ctx->w, events_dev, u_dev, logp_dev,
                           grad_dev, ctx->eval_cnt, ctx->eval_a, ctx->eval_b, ctx->eval_err, fin);
    if (!fin) launch_finish_fused<TN>(ctx, l, u_dev, logp_dev, grad_dev);
    return 0;
}

template <int TN>
static void launch_eval_tiles(seir_ctx *ctx, const LaunchCfg &l, const double *events_dev, bool grad) {
    Dims d = fused_dims<TN>(l);
    const int B = l.nb;
    const bool affinity = (l.affinity & 1) && xcd_affinity_applies(d.ntc * d.nmt, B);
    d.aff_nb = affinity ? B : 0;
    const size_t lds_b = eval_tiles_lds_bytes<TN>();
    static std::atomic<unsigned long long> attr_b{0ull};
    if (lds_b > 64 * 1024 && first_on_device(attr_b, ctx->device)) {
        (void)hipFuncSetAttribute((const void *)k_eval_tiles<true, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        (void)hipFuncSetAttribute((const void *)k_eval_tiles<false, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    }
    const dim3 grid((unsigned)((d.ntc * d.nmt + d.nrb_scan + d.Tp / WAVE) * B));
    if (grad) hipLaunchKernelGGL((k_eval_tiles<true, TN>), grid, dim3(512), lds_b, l.st, d, ctx->c, ctx->w, events_dev, B);
    else hipLaunchKernelGGL((k_eval_tiles<false, TN>), grid, dim3(512), lds_b, l.st, d, ctx->c, ctx->w, events_dev, B);
}
template <int TN>
static void launch_finish_fused(seir_ctx *ctx, const LaunchCfg &l, const double *u_dev, double *logp_dev, double *grad_dev) {
    const Dims d = fused_dims<TN>(l);
    const size_t lds_f = (size_t)d.Tp * sizeof(double);
    if (grad_dev) hipLaunchKernelGGL(k_finish<true>, dim3(l.nb), dim3(256), lds_f, l.st, d, ctx->c, ctx->w, u_dev, logp_dev, grad_dev, 1);
    else hipLaunchKernelGGL(k_finish<false>, dim3(l.nb), dim3(256), lds_f, l.st, d, ctx->c, ctx->w, u_dev, logp_dev, grad_dev, 1);
}
template <int TN>
static void launch_eval_fused(seir_ctx *ctx, const LaunchCfg &l, const double *u_dev, const double *events_dev,
                              double *logp_dev, double *grad_dev) {
    launch_state_params(ctx, l, u_dev, events_dev);
    launch_eval_tiles<TN>(ctx, l, events_dev, grad_dev != nullptr);
    launch_finish_fused<TN>(ctx, l, u_dev, logp_dev, grad_dev);
}

// The full evaluation.  Default: the fused form above.  SEIR_OPT_EVAL_FORM = 1 (and the fp32 contraction option)
// select the four-launch form: [state scan | parameter tables], mobility contraction,
// [S->E tiles | fold of the scan's I->R partials], reduction -- the tables depend on u only and the fold
// feeds the last launch only, so each rides along with the wide kernel next to it.
extern "C" int seir_log_prob_dev(seir_ctx *ctx, int32_t B, const double *u_dev, const double *events_dev,
                                 double *logp_dev, double *grad_dev) {
    int rc = check_batch(ctx, B);
    if (rc) return rc;
    if (!events_dev || !u_dev || !logp_dev) return fail(SEIR_ERR_INVALID, "null u/events/logp pointer");
    const LaunchCfg l = whole(ctx, B);
    Dims d = l.d;
    const bool f32 = ctx->opt_gemm_f32 && ctx->c.Cstar32 && ctx->w.Xn32;
    if (ctx->opt_eval_form != 1 && !f32) {
        bool one = false;
        if (ctx->opt_eval_form == 0 && B % 8 == 0 && (l.affinity & 1)) {
            // one launch for multiples of 8 chains (a single chain confined to one XCD loses more than the launches cost:
            // 46 / 54 us against 38 / 45, level at 4; 12 chains in the layout of 16 cost nearly what 16 do) while the parameter blocks and every tile workgroup
            // fit the chip at once (two workgroups of k_eval_all per CU: 128 VGPRs x 8 waves, <= 54 KB of LDS): UK-380
            // up to 16 chains (83 / 92 us against 87 / 93 in three launches)
            const int tn = d.Tp % 96 == 0 ? 96 : 64;
            const int per1 = (d.Tp / tn) * (d.Mp / GEMM_TM);
            const int nbv = (B + 7) / 8 * 8;
            // what the chip holds of this kernel at once: asked of the occupancy API with the launch's real dynamic LDS
            // (cached per instance), not assumed
            const int gi = grad_dev ? 1 : 0, ti = tn == 96 ? 1 : 0;
            if (ctx->eval_slots[ti][gi] < 0) {
                int occ = 0, cus = 0;
                const void *fn = tn == 96 ? (grad_dev ? (const void *)k_eval_all<true, 96> : (const void *)k_eval_all<false, 96>)
                                          : (grad_dev ? (const void *)k_eval_all<true, 64> : (const void *)k_eval_all<false, 64>);
                Dims dq = d;
                dq.nmt = d.Mp / GEMM_TM; dq.ntc = d.Tp / tn;
                const size_t lds = tn == 96 ? eval_all_lds_bytes<96>(dq) : eval_all_lds_bytes<64>(dq);
                if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 512, lds) != hipSuccess) occ = 0;
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
                ctx->eval_slots[ti][gi] = occ * cus;
            }
            // ... and ONE context per device: the launch's tiles wait, holding their CUs, for workgroups of the same launch; two
            // such launches from two contexts' streams can each be placed in part and wait for the other's CUs (three
            // contexts used in turn: 45 - 152 us per launch, sigma 32 us, profiles/r03_bench_kernel_stats.csv).  The
            // first context to get here owns the one-launch form on its device until it is destroyed; the others run the
            // three launches, which wait for nothing and overlap freely.
            if ((1 + per1) * nbv <= ctx->eval_slots[ti][gi] && claim_eval_all(ctx)) {
                if (ctx->xcd_local < 0) ctx->xcd_local = probe_xcd_local(ctx->stream) ? 1 : 0;
                one = ctx->xcd_local == 1;
            }
        }
        if (one) {
            rc = d.Tp % 96 == 0 ? launch_eval_all<96>(ctx, l, u_dev, events_dev, logp_dev, grad_dev)
                                : launch_eval_all<64>(ctx, l, u_dev, events_dev, logp_dev, grad_dev);
            if (rc) return rc;
        } else if (d.Tp % 96 == 0) {
            launch_eval_fused<96>(ctx, l, u_dev, events_dev, logp_dev, grad_dev);
        } else {
            launch_eval_fused<64>(ctx, l, u_dev, events_dev, logp_dev, grad_dev);
        }
    } else {
        const size_t lds = ((size_t)SCAN_WAVES * d.Tp * 2 + SCAN_LFT) * sizeof(double);
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)k_scan_params, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_scan_params, dim3(d.nrb_scan + 1, B), dim3(SCAN_WAVES * WAVE), lds, l.st, d, ctx->c, ctx->w,
                           events_dev, u_dev);
        launch_gemm(ctx, l);
        const bool affinity = (l.affinity & 1) && xcd_affinity_applies(d.ntc * d.nmt, B) && (d.ntc * B) % 8 == 0;
        d.aff_nb = affinity ? B : 0;
        const dim3 grid((unsigned)(d.ntc * d.nmt * B + d.ntc * B));
        if (grad_dev) hipLaunchKernelGGL(k_se_colreduce<true>, grid, dim3(256), 0, l.st, d, ctx->c, ctx->w, B);
        else hipLaunchKernelGGL(k_se_colreduce<false>, grid, dim3(256), 0, l.st, d, ctx->c, ctx->w, B);
        launch_finish(ctx, l, u_dev, logp_dev, grad_dev);
    }
    HIP_TRY(hipGetLastError());
    ctx->last_events = events_dev;
    ctx->prepared = true;
    ctx->last_u = u_dev; ctx->last_logp = logp_dev; ctx->last_grad = grad_dev;
    return 0;
}

static int check_eval_handoffs(seir_ctx *ctx);
static int host_eval(seir_ctx *ctx, int B, const double *u, const double *events, double *logp, double *grad) {
    int rc = check_batch(ctx, B);
    if (rc) return rc;
    if (!u || !events || !logp) return fail(SEIR_ERR_INVALID, "null host pointer");
    const Dims &d = ctx->d;
    HIP_TRY(hipMemcpyAsync(ctx->u_stage, u, sizeof(double) * B * d.P, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->ev_stage, events, sizeof(double) * B * d.M * d.T * 3, hipMemcpyHostToDevice,
                           ctx->stream));
    rc = seir_log_prob_dev(ctx, B, ctx->u_stage, ctx->ev_stage, ctx->logp_stage, grad ? ctx->grad_stage : nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(logp, ctx->logp_stage, sizeof(double) * B, hipMemcpyDeviceToHost, ctx->stream));
    if (grad)
        HIP_TRY(hipMemcpyAsync(grad, ctx->grad_stage, sizeof(double) * B * d.P, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return check_eval_handoffs(ctx);
}

extern "C" int seir_log_prob(seir_ctx *ctx, int32_t B, const double *u, const double *events, double *logp) {
    return host_eval(ctx, B, u, events, logp, nullptr);
}

extern "C" int seir_log_prob_grad(seir_ctx *ctx, int32_t B, const double *u, const double *events, double *logp,
                                  double *grad) {
    if (!grad) return fail(SEIR_ERR_INVALID, "null grad pointer");
    return host_eval(ctx, B, u, events, logp, grad);
}

#ifdef EVAL_STAMPS
extern "C" int seir_debug_eval_stamps(seir_ctx *ctx, unsigned long long *out) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out, ctx->eval_cnt + 8 * EVC_STRIDE, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif
// k_eval_all's bounded waits: a time-out means a consumer block went on without its producers' data
static int check_eval_handoffs(seir_ctx *ctx) {
    if (!ctx->eval_err) return 0;
    int n = 0;
    HIP_TRY(hipMemcpy(&n, ctx->eval_err, sizeof(int), hipMemcpyDeviceToHost));
    if (n) {
        (void)hipMemset(ctx->eval_err, 0, sizeof(int));
        return fail(SEIR_ERR_STATE, "%d in-launch hand-off(s) of the one-launch evaluation timed out: results since the last "
                    "synchronisation are unreliable (SEIR_OPT_EVAL_FORM 2 selects the three-launch form)", n);
    }
    return 0;
}

extern "C" int seir_sync(seir_ctx *ctx) {
    if (!ctx) return fail(SEIR_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (int rc = check_eval_handoffs(ctx)) return rc;
    return 0;
}

extern "C" void *seir_stream(seir_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int seir_malloc(void **p, uint64_t bytes) {
    if (!p) return fail(SEIR_ERR_INVALID, "null pointer");
    HIP_TRY(hipMalloc(p, bytes ? bytes : 8));
    return 0;
}
extern "C" int seir_free(void *p) {
    HIP_TRY(hipFree(p));
    return 0;
}
extern "C" int seir_memcpy_h2d(void *dst, const void *src, uint64_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int seir_memcpy_d2h(void *dst, const void *src, uint64_t bytes) {
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int seir_timer_start(seir_ctx *ctx) {
    if (!ctx) return fail(SEIR_ERR_INVALID, "null context");
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    return 0;
}
extern "C" int seir_timer_stop(seir_ctx *ctx, float *ms) {
    if (!ctx || !ms) return fail(SEIR_ERR_INVALID, "null argument");
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return 0;
}

extern "C" int seir_time_kernel(seir_ctx *ctx, int32_t which, int32_t B, int32_t iters, float *mean_ms) {
    int rc = check_batch(ctx, B);
    if (rc) return rc;
    if (!mean_ms || iters < 1) return fail(SEIR_ERR_INVALID, "bad iters/mean_ms");
    if (!ctx->last_events || !ctx->last_u || !ctx->last_logp)
        return fail(SEIR_ERR_STATE, "run one evaluation before timing a kernel");
    if ((which == SEIR_K_SE_GRAD || which == SEIR_K_TILES_GRAD) && !ctx->last_grad)
        return fail(SEIR_ERR_STATE, "last evaluation had no gradient buffer");
    auto once = [&]() {
        switch (which) {
            case SEIR_K_SCAN: launch_scan<0>(ctx, whole(ctx, B), ctx->last_events); break;
            case SEIR_K_GEMM: launch_gemm(ctx, whole(ctx, B)); break;
            case SEIR_K_SE_VALUE: launch_se<0>(ctx, whole(ctx, B), false); break;
            case SEIR_K_SE_GRAD: launch_se<0>(ctx, whole(ctx, B), true); break;
            case SEIR_K_STATE: launch_state_params(ctx, whole(ctx, B), ctx->last_u, ctx->last_events); break;
            case SEIR_K_TILES_VALUE:
            case SEIR_K_TILES_GRAD:
                if (ctx->d.Tp % 96 == 0) launch_eval_tiles<96>(ctx, whole(ctx, B), ctx->last_events, which == SEIR_K_TILES_GRAD);
                else launch_eval_tiles<64>(ctx, whole(ctx, B), ctx->last_events, which == SEIR_K_TILES_GRAD);
                break;
            case SEIR_K_FINISH_FUSED:
                if (ctx->d.Tp % 96 == 0) launch_finish_fused<96>(ctx, whole(ctx, B), ctx->last_u, ctx->last_logp, ctx->last_grad);
                else launch_finish_fused<64>(ctx, whole(ctx, B), ctx->last_u, ctx->last_logp, ctx->last_grad);
                break;
            default: launch_finish(ctx, whole(ctx, B), ctx->last_u, ctx->last_logp, ctx->last_grad); break;
        }
    };
    if (which < SEIR_K_SCAN || which > SEIR_K_FINISH_FUSED) return fail(SEIR_ERR_INVALID, "unknown kernel id %d", which);
    once();                                    // warm
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; ++i) once();
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    HIP_TRY(hipGetLastError());
    *mean_ms = ms / iters;
    return 0;
}

__global__ void k_selftest_math(Consts c, int n, const double *x, double *L, double *inv, double *lf) {
    __shared__ double2 ltab[LDSTAB_N];
    log_table_to_lds(ltab, c.logtab);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double a, b2;
        l1me_inv(x[i], a, b2, ltab);
        L[i] = a; inv[i] = b2;
        lf[i] = lfact(floor(x[i]), ltab);
    }
}

extern "C" int seir_selftest_math(seir_ctx *ctx, int32_t n, const double *x, double *L, double *inv, double *lf) {
    int rc = check_batch(ctx, 1);
    if (rc) return rc;
    if (n < 1 || !x || !L || !inv || !lf) return fail(SEIR_ERR_INVALID, "bad arguments");
    double *dx = nullptr;
    HIP_TRY(hipMalloc((void **)&dx, sizeof(double) * 4 * n));
    HIP_TRY(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest_math, dim3(64), dim3(256), 0, ctx->stream, ctx->c, n, dx, dx + n, dx + 2 * n, dx + 3 * n);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(L, dx + n, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(inv, dx + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lf, dx + 3 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(dx));
    return 0;
}

extern "C" int seir_reproduction_number(seir_ctx *ctx, int32_t n, const double *theta, const double *events,
                                        double *R_it) {
    int rc = check_batch(ctx, 1);
    if (rc) return rc;
    if (n < 1 || !theta || !events || !R_it) return fail(SEIR_ERR_INVALID, "bad arguments");
    const Dims &d = ctx->d;
    const int Bm = ctx->Bmax;
    double *rit_dev = nullptr;
    HIP_TRY(hipMalloc((void **)&rit_dev, sizeof(double) * Bm * d.T * d.M));
    for (int s0 = 0; s0 < n; s0 += Bm) {
        const int nb = std::min(Bm, n - s0);
        HIP_TRY(hipMemcpyAsync(ctx->u_stage, theta + (size_t)s0 * d.P, sizeof(double) * nb * d.P,
                               hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->ev_stage, events + (size_t)s0 * d.M * d.T * 3, sizeof(double) * nb * d.M * d.T * 3,
                               hipMemcpyHostToDevice, ctx->stream));
        launch_scan<0>(ctx, whole(ctx, nb), ctx->ev_stage);               // KS = (k_se, S - k_se): S_it
        hipLaunchKernelGGL(k_rt_tables, dim3(nb), dim3(256), 0, ctx->stream, d, ctx->w, ctx->u_stage);
        hipLaunchKernelGGL(k_rt, dim3((d.M + 63) / 64, (d.T + RT_TT - 1) / RT_TT, nb), dim3(256), k_rt_lds_bytes(d),
                           ctx->stream, d, ctx->c, ctx->w, ctx->u_stage, rit_dev);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(R_it + (size_t)s0 * d.T * d.M, rit_dev, sizeof(double) * nb * d.T * d.M,
                               hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    HIP_TRY(hipFree(rit_dev));
    ctx->prepared = false;                     // the scan overwrote the prepared-events workspace
    return 0;
}

// ===========================================================================
// Forward simulation (see include/seir_hip.h, "Chain-binomial forward simulation")
// ===========================================================================
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { HIP_TRY(hipMalloc(&p, bytes ? bytes : 8)); return 0; }
    template <typename T> T *as() const { return (T *)p; }
};
}  // namespace

__global__ void k_selftest_math_wide(Consts c, int n, const double *x, double *L, double *inv) {
    __shared__ double2 ltab[LDSTAB_N];
    log_table_to_lds(ltab, c.logtab);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double a, b2;
        l1me_inv_wide(x[i], a, b2, ltab);
        L[i] = a; inv[i] = b2;
    }
}

extern "C" int seir_selftest_math_wide(seir_ctx *ctx, int32_t n, const double *x, double *L, double *inv) {
    int rc = check_batch(ctx, 1);
    if (rc) return rc;
    if (n < 1 || !x || !L || !inv) return fail(SEIR_ERR_INVALID, "bad arguments");
    DevBuf dx, dL, di;
    if ((rc = dx.alloc(sizeof(double) * n)) || (rc = dL.alloc(sizeof(double) * n)) || (rc = di.alloc(sizeof(double) * n)))
        return rc;
    HIP_TRY(hipMemcpyAsync(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_selftest_math_wide, dim3(64), dim3(256), 0, ctx->stream, ctx->c, n, dx.as<double>(), dL.as<double>(),
                       di.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(L, dL.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(inv, di.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int seir_within_between(seir_ctx *ctx, int32_t n, const double *psi, const double *I_last, double W,
                                   double *within, double *between) {
    int rc = check_batch(ctx, 1);
    if (rc) return rc;
    if (n < 1 || !psi || !I_last || !within || !between) return fail(SEIR_ERR_INVALID, "bad arguments");
    const Dims &d = ctx->d;
    DevBuf dpsi, dI, dw, db;
    const size_t nm = sizeof(double) * (size_t)n * d.M;
    if ((rc = dpsi.alloc(sizeof(double) * n)) || (rc = dI.alloc(nm)) || (rc = dw.alloc(nm)) || (rc = db.alloc(nm)))
        return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(dpsi.p, psi, sizeof(double) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dI.p, I_last, nm, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_within_between, dim3(n), dim3(256), sizeof(double) * d.Mp, st, d, ctx->c, n, dpsi.as<double>(),
                       dI.as<double>(), W, dw.as<double>(), db.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(within, dw.p, nm, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(between, db.p, nm, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

extern "C" int seir_simulate(seir_ctx *ctx, const seir_sim_desc *sd) {
    int rc = check_batch(ctx, 1);
    if (rc) return rc;
    if (!sd || sd->num_draws < 1 || sd->num_steps < 1 || !sd->par || !sd->log_baseline || !sd->spatial || !sd->W ||
        !sd->weekday_c || !sd->init_state || !sd->events)
        return fail(SEIR_ERR_INVALID, "bad arguments");
    const Dims &d = ctx->d;
    const int M = d.M, S = sd->num_steps;
    if ((long long)S * M > 0x7fffffffLL) return fail(SEIR_ERR_INVALID, "num_steps * M overflows the cell counter");
    const size_t lds = k_simulate_lds_bytes(d);
    if (lds > 160 * 1024) return fail(SEIR_ERR_INVALID, "M=%d needs %zu B of LDS for the simulator", M, lds);
    HIP_TRY(hipFuncSetAttribute((const void *)k_simulate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // draws per batch: bound the device output buffer to ~512 MB
    const size_t per_draw = (size_t)M * S * 3 * sizeof(double);
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)sd->num_draws, ((size_t)512 << 20) / per_draw));
    DevBuf par, ap, sp, W, wd, init, ev;
    if ((rc = par.alloc(sizeof(double) * chunk * 5)) || (rc = ap.alloc(sizeof(double) * chunk * S)) ||
        (rc = sp.alloc(sizeof(double) * chunk * M)) || (rc = W.alloc(sizeof(double) * S)) ||
        (rc = wd.alloc(sizeof(double) * S)) || (rc = init.alloc(sizeof(double) * chunk * M * 4)) ||
        (rc = ev.alloc(per_draw * chunk)))
        return rc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(W.p, sd->W, sizeof(double) * S, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(wd.p, sd->weekday_c, sizeof(double) * S, hipMemcpyHostToDevice, st));
    for (int s0 = 0; s0 < sd->num_draws; s0 += chunk) {
        const int nb = std::min(chunk, sd->num_draws - s0);
        HIP_TRY(hipMemcpyAsync(par.p, sd->par + (size_t)s0 * 5, sizeof(double) * nb * 5, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(ap.p, sd->log_baseline + (size_t)s0 * S, sizeof(double) * nb * S, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(sp.p, sd->spatial + (size_t)s0 * M, sizeof(double) * nb * M, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(init.p, sd->init_state + (size_t)s0 * M * 4, sizeof(double) * nb * M * 4,
                               hipMemcpyHostToDevice, st));
        SimArgs a{};
        a.n = nb; a.S = S; a.first_draw = sd->first_draw_id + s0;
        a.k0 = (uint32_t)(sd->seed & 0xffffffffu); a.k1 = (uint32_t)(sd->seed >> 32);
        a.par = par.as<double>(); a.a_path = ap.as<double>(); a.spatial = sp.as<double>();
        a.W = W.as<double>(); a.wd = wd.as<double>(); a.init = init.as<double>(); a.events = ev.as<double>();
        hipLaunchKernelGGL(k_simulate, dim3(nb), dim3(SIM_THREADS), lds, st, d, ctx->c, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(sd->events + (size_t)s0 * M * S * 3, ev.p, per_draw * nb, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

__global__ void k_selftest_binomial(int count, const int *n, const double *p, uint32_t k0, uint32_t k1, int *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    RngKey key{k0, k1, 0u, (uint32_t)i};
    out[i] = sim_binomial(n[i], p[i], key, RS_SIM_BASE);
}

extern "C" int seir_selftest_binomial(seir_ctx *ctx, int32_t count, const int32_t *n, const double *p, uint64_t seed,
                                      int32_t *out) {
    int rc = check_batch(ctx, 1);
    if (rc) return rc;
    if (count < 1 || !n || !p || !out) return fail(SEIR_ERR_INVALID, "bad arguments");
    DevBuf dn, dp, dout;
    if ((rc = dn.alloc(sizeof(int) * count)) || (rc = dp.alloc(sizeof(double) * count)) ||
        (rc = dout.alloc(sizeof(int) * count)))
        return rc;
    HIP_TRY(hipMemcpyAsync(dn.p, n, sizeof(int) * count, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dp.p, p, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_selftest_binomial, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, count, dn.as<int>(),
                       dp.as<double>(), (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), dout.as<int>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, dout.p, sizeof(int) * count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ===========================================================================
// Sampler (see include/seir_hip.h, "Device-resident Metropolis-within-Gibbs")
// ===========================================================================
struct seir_sampler {
    seir_ctx *ctx = nullptr;
    SamplerCfg cfg{};
    Chains ch{};
    int record_events = 1;
    bool move_lds_attr = false;       // the event-update kernels were allowed more than 64 KB of dynamic LDS
    int pairs_lds_attr = 0;           // k_move_pairs was allowed its LDS request (one workgroup per CU): 1 yes, -1 refused
    std::vector<void *> allocs;
    // chains are independent: they are split into groups that run on their own streams
    // so that one group's single-workgroup-per-chain kernels overlap another group's wide ones
    int ngroups = 1;
    std::vector<hipStream_t> gstream;
    std::vector<hipGraph_t> graph;
    std::vector<hipGraphExec_t> gexec;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    hipStream_t copy_stream = nullptr;    // overlapped egress (seir_sampler_read_trace_async)
    hipEvent_t ev_burst = nullptr, ev_copy = nullptr;
    bool copy_pending = false;
    bool use_graph = false;       // seir_sampler_desc::use_graph
    bool hmc_chunked = true;      // hmc_mode 1: every leapfrog step by the single-workgroup kernel
    bool hmc_tail = true;         // hmc_mode 0 / 3: chunk roles inside the gradient launch (k_se_chunk) where xcd_local holds
    bool hmc_leap = true;         // hmc_mode 0: all inner steps in one persistent launch (k_leap) where every workgroup fits the chip
    int leap_occ[2][3][2];        // workgroups of k_leap<TSM, NTC, NST> the chip holds at once (occupancy query, cached; -1: not asked yet)
    int leap_rows = 0;            // seir_sampler_desc::leap_rows: 0 auto, 24 / 32: only that tile shape (else the per-step form)
    // seir_sampler_time_leapfrog: HIP events around the inner leapfrog steps of each sweep while it is on
    std::vector<hipEvent_t> prof_ev;     // pairs (before, after)
    int prof_i = -1, prof_launches = 0, prof_evals = 0;  // next pair to record (-1: off); launches / gradient evaluations of the section in the last sweep
    bool hmc_fold = true;         // hmc_mode 0 / 5: the trajectory's first step and both end-point gradients inside k_leap as well
    bool hmc_end = true;          // hmc_mode 0: ... and its last half kick, accept test, adaptation and trace (5: k_hmc_step<2> does those)
    bool hmc_tailfold = true;     // hmc_mode 0 (where k_leap does not fit) / 6: the trajectory's first and last step by the chunk roles of
                                  // the per-step launches as well (L + 1 k_se_chunk launches and k_hmc_final instead of k_se, k_hmc_step<0>,
                                  // L - 1 k_se_chunk, k_se, k_hmc_step<2>)
    bool vt_dirty = true;         // Work::Vt does not match Chains::var (set_kernel / set_adaptation / creation)
    unsigned long long leap_rsteps = 0;  // steps the ROLES of k_leap have done over all launches (the tiles do one more per folded launch)
    unsigned long long leap_steps = 0;   // leapfrog steps done by all k_leap launches so far (what Chains::leap's flags show)
    bool xcd_local = false;       // blocks with the same id mod 8 share an XCD (k_xcc_probe at creation)
    unsigned long long tail_count = 0;   // tiles per chain counted in by all k_se_chunk launches so far (Chains::tail)
    unsigned pbar_count = 0;      // workgroup arrivals every chain's step counter (Chains::pbar) has seen over all k_move_pairs launches
    int pair_debug = 0;           // debug_pair: test hooks of k_move_pair's handshake (1 late, 2 absent role 1)
    int moves_mode = 0;           // 0 = paired updates (k_move_pair) with the S->E-type proposal pre-drawn one pair ahead -- every pair of a
                                  //     sweep in one launch (k_move_pairs) where band workgroups can be part of it, else one launch per
                                  //     pair (4: always one launch per pair; 3: the same, never with band workgroups in the pair launch);
                                  // 1 = one proposal kernel per update (k_move_pa2); 2 = paired launches without the pre-draw
    int graph_skew = 0, graph_aff = 3;   // context options the captured graph was built with
    bool have_state = false;
    double *ev_stage = nullptr;       // [B][M][T][3] fp64 staging for set/get_state
    // --- recovery from a failed in-launch hand-off (seir_sampler_snapshot / _restore) ---
    // Every device buffer of the sampler is one of: chain STATE (what the next sweep's draws are a function of: copied by a
    // snapshot), HAND-OFF scratch (tokens, counters, descriptors in flight inside a sweep: zeroed by a restore, together
    // with the host's running totals of the counters), or neither (trace, staging).
    enum { R_OTHER = 0, R_STATE = 1, R_HANDOFF = 2 };
    struct Region { void *p; size_t bytes; int kind; };
    std::vector<Region> regions;
    void *snap[2] = {nullptr, nullptr};   // shadow copies of the STATE regions, packed
    bool snap_valid[2] = {false, false};
    size_t snap_bytes = 0;
    bool poisoned = false;            // a fatal hand-off time-out was reported: no sweeps until set_state / refresh / restore
    int poison_chain = 0;
    unsigned poison_count = 0;
    int hmc_mode = 0;                 // the launch forms in force (seir_sampler_set_launch_form)
};

template <typename T>
static int s_alloc(seir_sampler *s, T **p, size_t count, int kind = seir_sampler::R_OTHER) {
    void *q = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    HIP_TRY(hipMalloc(&q, bytes));
    s->allocs.push_back(q);
    s->regions.push_back({q, bytes, kind});
    HIP_TRY(hipMemset(q, 0, bytes));
    *p = (T *)q;
    return 0;
}

static void drop_graph(seir_sampler *s) {
    for (auto &g : s->gexec) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    for (auto &g : s->graph) if (g) { (void)hipGraphDestroy(g); g = nullptr; }
}

extern "C" void seir_sampler_destroy(seir_sampler *s) {
    if (s) for (hipEvent_t e : s->prof_ev) (void)hipEventDestroy(e);
    if (s) s->prof_ev.clear();
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    drop_graph(s);
    for (auto st : s->gstream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (auto e : s->ev_join) if (e) (void)hipEventDestroy(e);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->copy_stream) { (void)hipStreamSynchronize(s->copy_stream); (void)hipStreamDestroy(s->copy_stream); }
    if (s->ev_burst) (void)hipEventDestroy(s->ev_burst);
    if (s->ev_copy) (void)hipEventDestroy(s->ev_copy);
    for (void *p : s->allocs) (void)hipFree(p);
    for (void *p : s->snap) if (p) (void)hipFree(p);
    Work &w = s->ctx->w;
    for (int x = 0; x < 3; ++x) { w.K[x] = nullptr; w.St[x] = nullptr; }
    w.rowtot = w.rngtot = nullptr;
    w.TS = w.sp = w.gst = w.Vt = w.acur = w.rirc = w.CT = w.CG = w.Lpart0 = nullptr;
    delete s;
}

// seir_sampler_desc::hmc_mode / moves_mode -> the switches enqueue_sweep reads
static void apply_launch_form(seir_sampler *s, int hmc_mode, int moves_mode) {
    s->hmc_mode = hmc_mode;
    s->hmc_chunked = hmc_mode != 1;
    s->hmc_tail = hmc_mode == 0 || hmc_mode == 3 || hmc_mode == 4 || hmc_mode == 5;
    s->hmc_leap = hmc_mode == 0 || hmc_mode == 4 || hmc_mode == 5;
    s->hmc_fold = hmc_mode == 0 || hmc_mode == 5;
    s->hmc_end = hmc_mode == 0;
    s->hmc_tailfold = hmc_mode == 0 || hmc_mode == 6;
    if (hmc_mode == 6) s->hmc_tail = true;
    s->moves_mode = moves_mode;
}

extern "C" int seir_sampler_create(seir_ctx *ctx, const seir_sampler_desc *ds, seir_sampler **out) {
    if (!ctx || !ds || !out) return fail(SEIR_ERR_INVALID, "null argument");
    *out = nullptr;
    const Dims &d = ctx->d;
    const int B = ds->num_chains;
    if (B < 1 || B > ctx->Bmax) return fail(SEIR_ERR_INVALID, "num_chains=%d outside [1, %d]", B, ctx->Bmax);
    if (ctx->w.K[0]) return fail(SEIR_ERR_STATE, "this context already has a sampler");
    if (d.Tp > 2 * HB || d.M > 4 * HB)
        return fail(SEIR_ERR_INVALID, "sampler supports T <= %d and M <= %d", 2 * HB, 4 * HB);
    if (ds->m < 1 || ds->m > MMAX) return fail(SEIR_ERR_INVALID, "m=%d outside [1, %d]", ds->m, MMAX);
    if (ds->dmax < 1 || ds->nmax < 0 || ds->occult_nmax < 0 || ds->num_event_time_updates < 0)
        return fail(SEIR_ERR_INVALID, "bad dmax/nmax/occult_nmax/num_event_time_updates");
    if (ds->t_range_lo < 0 || ds->t_range_hi > d.T || ds->t_range_lo >= ds->t_range_hi)
        return fail(SEIR_ERR_INVALID, "occult t_range [%d,%d) outside [0,%d)", ds->t_range_lo, ds->t_range_hi, d.T);
    if (ds->num_leapfrog_steps < 1) return fail(SEIR_ERR_INVALID, "num_leapfrog_steps must be >= 1");
    if (ds->trace_capacity < 1) return fail(SEIR_ERR_INVALID, "trace_capacity must be >= 1");
    if (ds->record_events < 0 || ds->record_events > 2) return fail(SEIR_ERR_INVALID, "record_events is 0, 1 or 2");
    if (ds->moves_mode < 0 || ds->moves_mode > 4 || ds->hmc_mode < 0 || ds->hmc_mode > 6)
        return fail(SEIR_ERR_INVALID, "moves_mode is 0..4, hmc_mode 0..6");
    if (ds->disable_mask < 0 || ds->disable_mask > 31) return fail(SEIR_ERR_INVALID, "disable_mask is a 5-bit mask");
    if (ds->leap_rows != 0 && ds->leap_rows != 24 && ds->leap_rows != 32) return fail(SEIR_ERR_INVALID, "leap_rows is 0 (auto), 24 or 32");
    HIP_TRY(hipSetDevice(ctx->device));
    seir_sampler *s = new (std::nothrow) seir_sampler();
    if (!s) return fail(SEIR_ERR_DEVICE, "out of host memory");
    s->ctx = ctx;
    SamplerCfg &c = s->cfg;
    c.B = B; c.dmax = ds->dmax; c.nmax = ds->nmax; c.mmax = ds->m; c.occult_nmax = ds->occult_nmax;
    c.n_scans = ds->num_event_time_updates; c.tr_lo = ds->t_range_lo; c.tr_hi = ds->t_range_hi;
    c.L = ds->num_leapfrog_steps;
    c.k0 = (uint32_t)(ds->seed & 0xffffffffu); c.k1 = (uint32_t)(ds->seed >> 32);
    c.chain0 = ds->first_chain_id;
    c.adapt_step = 0; c.adapt_mass = 0; c.n_adapt = 0; c.target_accept = 0.75;
    c.cap = ds->trace_capacity;
    c.nrb_d = (d.M + 7) / 8;        // row blocks of k_move_delta (4-row blocks measured slower, twice)
    s->record_events = ds->record_events;
    c.ev16 = ds->record_events == 2 ? 1 : 0;
    // Launch mode of a sweep's ~76 dependent kernels.  Measured on MI355X / ROCm 7.2 (UK-380, 8 chains):
    // stream launches 0.815 ms per sweep, replay of the captured hipGraph 0.872 ms -- the graph
    // executor costs ~0.75 us more per node than the stream path while the host (3-4 us per launch,
    // kernels of ~10 us) stays ahead either way.  Default: stream launches; seir_sampler_desc::use_graph selects the graph.
    s->use_graph = ds->use_graph != 0;
    s->pair_debug = ds->debug_pair;
    s->leap_rows = ds->leap_rows;
    apply_launch_form(s, ds->hmc_mode, ds->moves_mode);
    for (auto &a : s->leap_occ) for (auto &b2 : a) for (int &v : b2) v = -1;
    c.disable_mask = ds->disable_mask;
    {
        int g = ds->chain_groups;    // measured: concurrent chain groups on several streams do not overlap profitably
        if (g < 1) g = 1;
        if (g > B) g = B;
        s->ngroups = g;
        s->gstream.assign(g, nullptr); s->graph.assign(g, nullptr); s->gexec.assign(g, nullptr);
        s->ev_join.assign(g, nullptr);
    }

    int rc = 0;
    Work &w = ctx->w;
    const size_t cells = (size_t)ctx->Bmax * d.Mp * d.Tp;
    Chains &ch = s->ch;
#define S_ALLOC(ptr, n) if (!rc) rc = s_alloc(s, &(ptr), (n))
#define S_STATE(ptr, n) if (!rc) rc = s_alloc(s, &(ptr), (n), seir_sampler::R_STATE)
#define S_HAND(ptr, n) if (!rc) rc = s_alloc(s, &(ptr), (n), seir_sampler::R_HANDOFF)
    for (int x = 0; x < 3; ++x) { S_STATE(w.K[x], cells); S_STATE(w.St[x], cells); }
    S_STATE(w.rowtot, (size_t)ctx->Bmax * 2 * d.Mp);
    S_STATE(w.rngtot, (size_t)ctx->Bmax * 2 * d.Mp);
    S_ALLOC(w.TS, (size_t)ctx->Bmax * d.nmt * d.ntc * 4);
    S_ALLOC(w.Lpart0, (size_t)ctx->Bmax * d.nmt * d.ntc);
    S_STATE(w.sp, (size_t)ctx->Bmax * 2 * d.Mp);
    S_STATE(w.gst, (size_t)ctx->Bmax * 2 * GST_N);
    S_STATE(w.Vt, (size_t)ctx->Bmax * d.Tp);
    S_STATE(w.acur, (size_t)ctx->Bmax * d.Tp);
    S_ALLOC(w.rirc, (size_t)ctx->Bmax * 2 * d.Tp);
    S_STATE(w.CT, (size_t)ctx->Bmax * 2 * CT_MAXC * 4);
    S_STATE(w.CG, (size_t)ctx->Bmax * 2 * CT_MAXC * 2);
    S_STATE(ch.q, (size_t)B * d.Pp); S_STATE(ch.p, (size_t)B * d.Pp); S_STATE(ch.q0, (size_t)B * d.Pp);
    S_STATE(ch.grad, (size_t)B * d.Pp); S_STATE(ch.var, (size_t)B * d.Pp);
    S_STATE(ch.rv_mean, (size_t)B * d.Pp); S_STATE(ch.rv_m2, (size_t)B * d.Pp);
    S_STATE(ch.hs, (size_t)B * NHS);
    S_HAND(ch.mv, (size_t)2 * B);
    S_HAND(ch.fpend, (size_t)B);
    S_HAND(ch.mvfix, (size_t)2 * B);
    S_HAND(ch.mvsel, (size_t)2 * B);
    S_HAND(ch.hand, (size_t)B);
    S_HAND(ch.late, (size_t)2 * B);
    ch.late_fatal = B;
    S_HAND(ch.tail, (size_t)B * TAIL_STRIDE + (size_t)B * TAIL_FLAG_STRIDE);
    S_HAND(ch.leap, (size_t)B * LEAP_CH);
    // k_leap's hand-off words (16 bytes per value, two step parities): the tiles' partial sums and the roles' tables
    S_HAND(ch.llK, (size_t)B * 2 * (d.Mp / 16) * d.Tp);
    S_HAND(ch.llR, (size_t)B * 2 * d.ntc * d.Mp);
    S_HAND(ch.llP, (size_t)B * 2 * d.ntc * (d.Mp / 16));
    S_HAND(ch.llTS, (size_t)B * 2 * d.ntc * (d.Mp / 16) * 4);
    S_HAND(ch.llT, (size_t)B * ((size_t)d.Tp + 2 * (size_t)d.Mp + 8));
    S_HAND(ch.llmv, (size_t)2 * B * 32);
    S_HAND(ch.k0part, (size_t)B * ROLE_SLOTS);
    S_HAND(ch.irl0, (size_t)B);
    S_ALLOC(ch.leap_st, (size_t)(B + 2) * 16 * 8 + 4096);
    S_HAND(ch.done, (size_t)B * 2 * TAIL_STRIDE);
    S_HAND(ch.pbar, (size_t)B * PBAR_STRIDE);
    S_HAND(ch.finpart, (size_t)B * ROLE_SLOTS * 4);
    S_HAND(ch.hand2, (size_t)B);
    S_HAND(ch.mvs, (size_t)2 * B);
    S_HAND(ch.DownS, (size_t)2 * B * 2);
    S_HAND(ch.prev, (size_t)2 * B);
    S_HAND(ch.Dpart, (size_t)B * c.nrb_d * 2);
    S_HAND(ch.Down, (size_t)2 * 2 * B * 2);
    S_STATE(ch.sweep, (size_t)B); S_STATE(ch.slot0, 1);
    S_ALLOC(ch.tr_theta, (size_t)c.cap * B * d.P);
    {
        char *tre = nullptr;                               // bytes: int32 or uint16 per count
        S_ALLOC(tre, s->record_events ? (size_t)c.cap * B * d.M * d.T * 3 * (c.ev16 ? 2 : 4) : 4);
        ch.tr_events = tre;
    }
    S_ALLOC(ch.ev_overflow, 1);
    S_ALLOC(ch.tr_hmc, (size_t)c.cap * B * 3);
    S_ALLOC(ch.tr_mv, (size_t)c.cap * B * 4 * NMVTR);
    S_ALLOC(s->ev_stage, (size_t)B * d.M * d.T * 3);
#undef S_STATE
#undef S_HAND
    if (!rc) {
        // chain state held in the context's own work arrays (written by accepted event updates and by the HMC roles):
        // part of a snapshot too.  Sizes as in create_impl (max_chains chains).
        const size_t Bm = (size_t)ctx->Bmax;
        auto st_ = [&](void *p_, size_t bytes_) { s->regions.push_back({p_, bytes_, seir_sampler::R_STATE}); };
        st_(w.F, cells * sizeof(double));
        st_(w.rowconst, Bm * d.Mp * sizeof(double));
        st_(w.ea, Bm * d.Tp * sizeof(double)); st_(w.eb, Bm * d.Mp * sizeof(double)); st_(w.rir, Bm * d.Tp * sizeof(double));
        st_(w.scal, Bm * NSCAL * sizeof(double)); st_(w.Qs, Bm * d.Mp * sizeof(double));
        st_(w.Kir, Bm * d.Tp * sizeof(double)); st_(w.Dir, Bm * d.Tp * sizeof(double));
        st_(w.constsum, Bm * sizeof(double));
    }
#undef S_ALLOC
    if (!rc) {
        std::vector<double> ones((size_t)B * d.Pp, 1.0), hs((size_t)B * NHS, 0.0);
        for (int b = 0; b < B; ++b) hs[(size_t)b * NHS + HS_EPS] = 0.1;     // inference.py:325
        hipError_t e = hipMemcpy(ch.var, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ch.hs, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(SEIR_ERR_DEVICE, "sampler init copy failed: %s", hipGetErrorString(e));
    }
    if (!rc) {
        hipError_t e = hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_burst, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_copy, hipEventDisableTiming);
        for (int g = 0; g < s->ngroups && e == hipSuccess; ++g) {
            e = hipStreamCreateWithFlags(&s->gstream[g], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_join[g], hipEventDisableTiming);
        }
        if (e != hipSuccess) rc = fail(SEIR_ERR_DEVICE, "sampler stream setup failed: %s", hipGetErrorString(e));
    }
    if (!rc && (s->hmc_tail || s->moves_mode == 0 || s->moves_mode == 2 || s->moves_mode == 4)) {
        // Do blocks with the same id mod 8 share an XCD here?  (probe_xcd_local)
        if (ctx->xcd_local < 0) ctx->xcd_local = probe_xcd_local(ctx->stream) ? 1 : 0;
        const bool ok = ctx->xcd_local == 1;
        s->xcd_local = ok;
    }
    if (rc) { seir_sampler_destroy(s); return rc; }
    *out = s;
    return 0;
}

extern "C" int seir_sampler_xcd_local(seir_sampler *s) { return (s && s->xcd_local) ? 1 : 0; }

static int sampler_check(seir_sampler *s) {
    if (!s) return fail(SEIR_ERR_INVALID, "null sampler");
    HIP_TRY(hipSetDevice(s->ctx->device));
    return 0;
}

// Hand-off scratch back to its initial state, in stream order: every token / counter / in-flight descriptor zero (0 is no
// launch's token) and the host's running totals of the counters with them.  Between two sweeps nothing of it is live
// (a sweep's first launch starts with have_prev = have_pre = 0), so this is always allowed there.
static int reset_handoffs(seir_sampler *s) {
    hipStream_t st = s->ctx->stream;
    for (const auto &r : s->regions)
        if (r.kind == seir_sampler::R_HANDOFF) HIP_TRY(hipMemsetAsync(r.p, 0, r.bytes, st));
    s->leap_rsteps = s->leap_steps = 0;
    s->tail_count = 0;
    s->pbar_count = 0;
    s->poisoned = false;
    return 0;
}

extern "C" int seir_sampler_set_launch_form(seir_sampler *s, int32_t hmc_mode, int32_t moves_mode) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (moves_mode < 0 || moves_mode > 4 || hmc_mode < 0 || hmc_mode > 6)
        return fail(SEIR_ERR_INVALID, "moves_mode is 0..4, hmc_mode 0..6");
    if (hmc_mode != s->hmc_mode || moves_mode != s->moves_mode) {
        HIP_TRY(hipStreamSynchronize(s->ctx->stream));
        drop_graph(s);                               // the captured sweep is one form's launches
        apply_launch_form(s, hmc_mode, moves_mode);
    }
    return 0;
}

extern "C" int seir_sampler_launch_form(seir_sampler *s, int32_t *hmc_mode, int32_t *moves_mode) {
    if (!s) return fail(SEIR_ERR_INVALID, "null sampler");
    if (hmc_mode) *hmc_mode = s->hmc_mode;
    if (moves_mode) *moves_mode = s->moves_mode;
    return 0;
}

// Test hook: what a timed-out wait leaves behind -- chain `chain`'s fatal counter raised, in stream order.  Every wait of
// that chain then gives up at its first look at the counter (a wait that is served within 256 polls still completes) and
// the next read of the trace reports the time-out.
extern "C" int seir_sampler_debug_fail_handoff(seir_sampler *s, int32_t chain) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (chain < 0 || chain >= s->cfg.B) return fail(SEIR_ERR_INVALID, "chain %d outside [0, %d)", chain, s->cfg.B);
    static const uint32_t one = 1u;
    HIP_TRY(hipMemcpyAsync(s->ch.late + s->ch.late_fatal + chain, &one, sizeof(one), hipMemcpyHostToDevice, s->ctx->stream));
    return 0;
}

extern "C" int seir_sampler_snapshot(seir_sampler *s, int32_t slot) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (slot < 0 || slot > 1) return fail(SEIR_ERR_INVALID, "snapshot slot is 0 or 1");
    if (!s->have_state) return fail(SEIR_ERR_STATE, "no chain state set");
    if (s->poisoned) return fail(SEIR_ERR_STATE, "the sampler's state is unreliable (hand-off time-out): nothing to snapshot");
    if (!s->snap_bytes)
        for (const auto &r : s->regions)
            if (r.kind == seir_sampler::R_STATE) s->snap_bytes += (r.bytes + 255) / 256 * 256;
    if (!s->snap[slot]) HIP_TRY(hipMalloc(&s->snap[slot], s->snap_bytes));
    // in stream order: the snapshot is the state after everything queued so far
    size_t off = 0;
    for (const auto &r : s->regions)
        if (r.kind == seir_sampler::R_STATE) {
            HIP_TRY(hipMemcpyAsync((char *)s->snap[slot] + off, r.p, r.bytes, hipMemcpyDeviceToDevice, s->ctx->stream));
            off += (r.bytes + 255) / 256 * 256;
        }
    s->snap_valid[slot] = true;
    return 0;
}

extern "C" int seir_sampler_restore(seir_sampler *s, int32_t slot) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (slot < 0 || slot > 1) return fail(SEIR_ERR_INVALID, "snapshot slot is 0 or 1");
    if (!s->snap_valid[slot] || !s->snap[slot]) return fail(SEIR_ERR_STATE, "no snapshot in slot %d", slot);
    hipStream_t st = s->ctx->stream;
    // whatever is still queued (the rest of a failed burst: its waits give up at their first look at the chain's time-out
    // counter, so it drains quickly) and the copy of a burst that nobody wants any more
    HIP_TRY(hipStreamSynchronize(st));
    if (s->copy_pending) { (void)hipEventSynchronize(s->ev_copy); s->copy_pending = false; }
    size_t off = 0;
    for (const auto &r : s->regions)
        if (r.kind == seir_sampler::R_STATE) {
            HIP_TRY(hipMemcpyAsync(r.p, (const char *)s->snap[slot] + off, r.bytes, hipMemcpyDeviceToDevice, st));
            off += (r.bytes + 255) / 256 * 256;
        }
    if ((rc = reset_handoffs(s))) return rc;
    s->vt_dirty = true;                              // Work::Vt came back with the snapshot, the flag did not
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

static void enqueue_refresh(seir_sampler *s) {
    seir_ctx *ctx = s->ctx;
    const Dims &d = ctx->d;
    const int B = s->cfg.B;
    launch_scan<1>(ctx, whole(ctx, B), nullptr);
    hipLaunchKernelGGL(k_range_totals, dim3((d.M + 3) / 4, B), dim3(256), 0, ctx->stream, d, ctx->w, s->cfg);
    launch_colreduce(ctx, whole(ctx, B));
    launch_gemm(ctx, whole(ctx, B));
    hipLaunchKernelGGL(k_chain_tables, dim3(B), dim3(256), 0, ctx->stream, d, ctx->c, ctx->w, s->ch);
    launch_se<1>(ctx, whole(ctx, B), false);
    hipLaunchKernelGGL(k_chain_refresh, dim3(B), dim3(256), (size_t)d.Tp * sizeof(double), ctx->stream, d, ctx->c,
                       ctx->w, s->ch);
}

extern "C" int seir_sampler_refresh(seir_sampler *s) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (!s->have_state) return fail(SEIR_ERR_STATE, "no chain state set");
    if (s->poisoned) {
        HIP_TRY(hipStreamSynchronize(s->ctx->stream));
        if ((rc = reset_handoffs(s))) return rc;
    }
    enqueue_refresh(s);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int seir_sampler_set_state(seir_sampler *s, const double *u, const double *events) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (!u || !events) return fail(SEIR_ERR_INVALID, "null pointer");
    seir_ctx *ctx = s->ctx;
    const Dims &d = ctx->d;
    const int B = s->cfg.B;
    for (size_t i = 0, n = (size_t)B * d.M * d.T * 3; i < n; ++i)
        if (!(events[i] >= 0.0 && events[i] < 2147483648.0 && events[i] == std::floor(events[i])))
            return fail(SEIR_ERR_INVALID, "events[%zu]=%g is not a non-negative integer count", i, events[i]);
    HIP_TRY(hipMemcpyAsync(s->ch.q, u, sizeof(double) * B * d.P, hipMemcpyHostToDevice, ctx->stream));
    // Chains::q0 = the position at the start of the next trajectory, kept equal to q between trajectories (k_hmc_step<2>
    // leaves it so): the folded first step reads the start point from it while its roles already write the next one to q
    HIP_TRY(hipMemcpyAsync(s->ch.q0, u, sizeof(double) * B * d.P, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(s->ev_stage, events, sizeof(double) * B * d.M * d.T * 3, hipMemcpyHostToDevice,
                           ctx->stream));
    hipLaunchKernelGGL(k_import_events, dim3(1024), dim3(256), 0, ctx->stream, d, ctx->w, s->ev_stage, B);
    s->have_state = true;
    if (s->poisoned && (rc = reset_handoffs(s))) return rc;
    enqueue_refresh(s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int seir_sampler_get_state(seir_sampler *s, double *u, double *events, double *logp) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (!s->have_state) return fail(SEIR_ERR_STATE, "no chain state set");
    seir_ctx *ctx = s->ctx;
    const Dims &d = ctx->d;
    const int B = s->cfg.B;
    if (u) HIP_TRY(hipMemcpyAsync(u, s->ch.q, sizeof(double) * B * d.P, hipMemcpyDeviceToHost, ctx->stream));
    if (events) {
        hipLaunchKernelGGL(k_export_events, dim3(1024), dim3(256), 0, ctx->stream, d, ctx->w, s->ev_stage, B);
        HIP_TRY(hipMemcpyAsync(events, s->ev_stage, sizeof(double) * B * d.M * d.T * 3, hipMemcpyDeviceToHost,
                               ctx->stream));
    }
    std::vector<double> hs;
    if (logp) {
        hs.resize((size_t)B * NHS);
        HIP_TRY(hipMemcpyAsync(hs.data(), s->ch.hs, sizeof(double) * B * NHS, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (logp)
        for (int b = 0; b < B; ++b) logp[b] = hs[(size_t)b * NHS + HS_LP_THETA] + hs[(size_t)b * NHS + HS_LP_CONST];
    return 0;
}

static int hs_update(seir_sampler *s, const std::vector<std::pair<int, const double *>> &cols) {
    const int B = s->cfg.B;
    std::vector<double> hs((size_t)B * NHS);
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    HIP_TRY(hipMemcpy(hs.data(), s->ch.hs, sizeof(double) * B * NHS, hipMemcpyDeviceToHost));
    for (auto &c : cols)
        for (int b = 0; b < B; ++b) hs[(size_t)b * NHS + c.first] = c.second[b];
    HIP_TRY(hipMemcpy(s->ch.hs, hs.data(), sizeof(double) * B * NHS, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int seir_sampler_set_kernel(seir_sampler *s, const double *step_size, const double *variance) {
    int rc = sampler_check(s);
    if (rc) return rc;
    const Dims &d = s->ctx->d;
    const int B = s->cfg.B;
    if (step_size) {
        for (int b = 0; b < B; ++b)
            if (!(step_size[b] > 0.0)) return fail(SEIR_ERR_INVALID, "step_size[%d] must be positive", b);
        if ((rc = hs_update(s, {{HS_EPS, step_size}}))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    if (variance) {
        for (size_t i = 0; i < (size_t)B * d.P; ++i)
            if (!(variance[i] > 0.0)) return fail(SEIR_ERR_INVALID, "variance[%zu] must be positive", i);
        HIP_TRY(hipMemcpy(s->ch.var, variance, sizeof(double) * B * d.P, hipMemcpyHostToDevice));
        s->vt_dirty = true;
    } else {
        std::vector<double> ones((size_t)B * d.P, 1.0);
        HIP_TRY(hipMemcpy(s->ch.var, ones.data(), sizeof(double) * B * d.P, hipMemcpyHostToDevice));
        s->vt_dirty = true;
    }
    return 0;
}

extern "C" int seir_sampler_get_kernel(seir_sampler *s, double *step_size, double *variance) {
    int rc = sampler_check(s);
    if (rc) return rc;
    const Dims &d = s->ctx->d;
    const int B = s->cfg.B;
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    if (step_size) {
        std::vector<double> hs((size_t)B * NHS);
        HIP_TRY(hipMemcpy(hs.data(), s->ch.hs, sizeof(double) * B * NHS, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) step_size[b] = hs[(size_t)b * NHS + HS_EPS];
    }
    if (variance) HIP_TRY(hipMemcpy(variance, s->ch.var, sizeof(double) * B * d.P, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int seir_sampler_set_adaptation(seir_sampler *s, int32_t adapt_step, int32_t adapt_mass, int32_t n_adapt,
                                           double target, const double *rv_count, const double *rv_mean,
                                           const double *rv_var) {
    int rc = sampler_check(s);
    if (rc) return rc;
    const Dims &d = s->ctx->d;
    const int B = s->cfg.B;
    if (adapt_mass && (!rv_count || !rv_mean || !rv_var))
        return fail(SEIR_ERR_INVALID, "adapt_mass needs the initial running variance");
    if (adapt_step && !(target > 0.0 && target < 1.0)) return fail(SEIR_ERR_INVALID, "target_accept_prob in (0,1)");
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    std::vector<double> hs((size_t)B * NHS);
    HIP_TRY(hipMemcpy(hs.data(), s->ch.hs, sizeof(double) * B * NHS, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
        double *h = hs.data() + (size_t)b * NHS;
        // a fresh DualAveragingStepSizeAdaptation: error_sum 0, step 0, log_averaging_step 0,
        // shrinkage target log(10 * current step size)
        h[HS_DA_ERR] = 0.0; h[HS_DA_STEP] = 0.0; h[HS_DA_LOGAVG] = 0.0;
        h[HS_DA_MU] = std::log(10.0 * h[HS_EPS]);
        if (adapt_mass) h[HS_RV_N] = rv_count[b];
    }
    HIP_TRY(hipMemcpy(s->ch.hs, hs.data(), sizeof(double) * B * NHS, hipMemcpyHostToDevice));
    if (adapt_mass) {
        std::vector<double> m2((size_t)B * d.P);
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < d.P; ++i) {
                const double v = rv_var[(size_t)b * d.P + i];
                if (!(v > 0.0)) return fail(SEIR_ERR_INVALID, "running variance must be positive");
                m2[(size_t)b * d.P + i] = v * rv_count[b];
            }
        HIP_TRY(hipMemcpy(s->ch.rv_mean, rv_mean, sizeof(double) * B * d.P, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->ch.rv_m2, m2.data(), sizeof(double) * B * d.P, hipMemcpyHostToDevice));
        // momentum distribution at bootstrap: the initial running variance
        HIP_TRY(hipMemcpy(s->ch.var, rv_var, sizeof(double) * B * d.P, hipMemcpyHostToDevice));
    }
    SamplerCfg &c = s->cfg;
    if (c.adapt_step != adapt_step || c.adapt_mass != adapt_mass || c.n_adapt != n_adapt ||
        c.target_accept != target)
        drop_graph(s);                               // kernel arguments are baked into the graph
    c.adapt_step = adapt_step; c.adapt_mass = adapt_mass; c.n_adapt = n_adapt; c.target_accept = target;
    s->vt_dirty = true;                              // the variances may have moved since Work::Vt was last formed (k_vt)
    return 0;
}

extern "C" int seir_sampler_reset_trace(seir_sampler *s) {
    int rc = sampler_check(s);
    if (rc) return rc;
    // slot0 = sweep counter of chain 0 (all chains advance together)
    HIP_TRY(hipMemcpyAsync(s->ch.slot0, s->ch.sweep, sizeof(unsigned), hipMemcpyDeviceToDevice, s->ctx->stream));
    return 0;
}

extern "C" int seir_sampler_reset_trace_at(seir_sampler *s, int32_t first_slot) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (first_slot < 0 || first_slot >= s->cfg.cap)
        return fail(SEIR_ERR_INVALID, "first_slot %d outside [0, %d)", first_slot, s->cfg.cap);
    hipLaunchKernelGGL(k_set_slot0, dim3(1), dim3(1), 0, s->ctx->stream, s->ch, (unsigned)first_slot);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int HT, int HM>
static void launch_hmc_t(seir_ctx *ctx, const LaunchCfg &l, const SamplerCfg &c, const Chains &ch, int stage,
                         int gather_qs) {
    const size_t lds = (size_t)l.d.Mp * sizeof(double);
    const dim3 g(l.nb), blk(HB);
    if (stage == 0)
        hipLaunchKernelGGL((k_hmc_step<0, HT, HM>), g, blk, lds, l.st, l.d, ctx->c, ctx->w, c, ch, gather_qs);
    else if (stage == 1)
        hipLaunchKernelGGL((k_hmc_step<1, HT, HM>), g, blk, lds, l.st, l.d, ctx->c, ctx->w, c, ch, gather_qs);
    else
        hipLaunchKernelGGL((k_hmc_step<2, HT, HM>), g, blk, lds, l.st, l.d, ctx->c, ctx->w, c, ch, gather_qs);
}
static void launch_hmc(seir_ctx *ctx, const LaunchCfg &l, const SamplerCfg &c, const Chains &ch, int stage,
                       int gather_qs = 0) {
    const int ht = (l.d.Tp + HB - 1) / HB, hm = (l.d.M + HB - 1) / HB;    // sampler_create caps T<=1024, M<=2048
    if (ht <= 1) {
        if (hm <= 1) launch_hmc_t<1, 1>(ctx, l, c, ch, stage, gather_qs);
        else if (hm <= 2) launch_hmc_t<1, 2>(ctx, l, c, ch, stage, gather_qs);
        else launch_hmc_t<1, 4>(ctx, l, c, ch, stage, gather_qs);
    } else {
        if (hm <= 1) launch_hmc_t<2, 1>(ctx, l, c, ch, stage, gather_qs);
        else if (hm <= 2) launch_hmc_t<2, 2>(ctx, l, c, ch, stage, gather_qs);
        else launch_hmc_t<2, 4>(ctx, l, c, ch, stage, gather_qs);
    }
}

// chains [b0, b0+nb) of group g
static void group_range(const seir_sampler *s, int g, int &b0, int &nb) {
    const int B = s->cfg.B, G = s->ngroups;
    b0 = (int)((long long)B * g / G);
    nb = (int)((long long)B * (g + 1) / G) - b0;
}

// the instance of the persistent leapfrog kernel for (tile-scalar mode, day chunks, gradient tiles per workgroup)
// nst = 2: two 16-row gradient tiles per workgroup (32 rows); nst = 1: ONE 24-row tile per workgroup, six rows per wave -- the
// shape whose 96 workgroups per chain divide UK-380's XCD evenly (instantiated for that size class: M <= 512, six day chunks)
static const void *leap_fn(int ts_mode, int ntc, int nst) {
    if (nst == 1) return (const void *)k_leap<1, 6, 1, 6>;
#define LEAP_ROW(TSM_, NTC_) ((const void *)k_leap<TSM_, NTC_, 2>)
    if (ts_mode == 1) return ntc == 1 ? LEAP_ROW(1, 1) : ntc == 6 ? LEAP_ROW(1, 6) : LEAP_ROW(1, 12);
    return ntc == 1 ? LEAP_ROW(2, 1) : ntc == 6 ? LEAP_ROW(2, 6) : LEAP_ROW(2, 12);
#undef LEAP_ROW
}

static void enqueue_sweep(seir_sampler *s, int g) {
    seir_ctx *ctx = s->ctx;
    const SamplerCfg &c = s->cfg;
    int b0, nb;
    group_range(s, g, b0, nb);
    LaunchCfg l{ctx->d, s->gstream[g], nb, ctx->opt_affinity};
    l.d.b0 = b0;
    l.d.skew = ctx->opt_skew;
    const Dims d0 = l.d;
    hipStream_t st = l.st;
    // [part 0] HMC on u | events: L+1 gradient evaluations
    const bool chunked = s->hmc_chunked && c.L >= 3 && d0.ntc <= CT_MAXC;
    const int ts_mode = chunked ? (d0.Mp <= 512 ? 1 : 2) : 0;   // 1: the M-chunks sum the row partials themselves
    l.d.sp_par = 0;
    // Is the persistent leapfrog launch (k_leap) usable for this sweep?  (the conditions of k_se_chunk -- XCD placement checked,
    // one stream, the XCD-affine grid -- and every workgroup of the launch resident at once: its tiles wait for the roles)
    const int per_roles = d0.ntc + d0.Mp / WAVE, ntile_all = d0.ntc * d0.nmt, nbv_all = (nb + 7) / 8 * 8;
    bool leap_ok = false;
    // the tile shape: 24-row workgroups (k_leap<1, 6, 1, 6>: nmt24 row tiles, one per workgroup) where they exist for the size
    // and the whole launch -- three of them per CU and the roles beside -- is resident; else 32-row workgroups (two 16-row tiles)
    int leap_nst = 2, leap_nmt = d0.nmt, leap_wgs = ntile_all / 2;
    auto leap_slots = [&](int ti, int ni, int nst) {
        int &slot = s->leap_occ[ti][ni][nst - 1];
        if (slot < 0) {
            int occ = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, leap_fn(ti + 1, ni == 0 ? 1 : ni == 1 ? 6 : 12, nst), 256, 0) != hipSuccess) occ = 0;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
            slot = occ * cus;                                     // workgroups the chip holds at once
        }
        return (long long)slot;
    };
    // chains per launch of k_leap: all of them -- or, for 16 chains, which do not fit the chip at once, two launches of 8 one
    // after the other (2 x 134 us at UK-380 against 335 us for the 18 launches of the per-step form; from 24 chains on the
    // per-step form is the faster one)
    int leap_nbv = nbv_all;
    if (chunked && s->hmc_leap && s->hmc_tail && s->xcd_local && s->ngroups == 1 && (l.affinity & 1) && !s->use_graph &&
        xcd_affinity_applies(ntile_all, nbv_all) && (d0.ntc == 1 || d0.ntc == 6 || d0.ntc == 12) && c.L >= 3 && d0.nmt <= WAVE &&
        d0.nmt % 2 == 0) {
        const int ti = ts_mode - 1, ni = d0.ntc == 1 ? 0 : d0.ntc == 6 ? 1 : 2;
        const int nmt24 = d0.Mp / 24, wgs24 = d0.ntc * nmt24;
        auto fit = [&](int nbv) {
            if (s->leap_rows != 32 && ts_mode == 1 && d0.ntc == 6 && d0.Mp % 24 == 0 && xcd_affinity_applies(wgs24, nbv) &&
                (long long)(wgs24 + per_roles) * nbv <= leap_slots(ti, ni, 1)) {
                leap_nst = 1; leap_nmt = nmt24; leap_wgs = wgs24;
                return true;
            }
            if (s->leap_rows != 24) return (long long)(ntile_all / 2 + per_roles) * nbv <= leap_slots(ti, ni, 2);
            return false;
        };
        leap_ok = fit(nbv_all);
        if (!leap_ok && nb == 16 && fit(8)) { leap_ok = true; leap_nbv = 8; }
    }
    auto launch_leap = [&](int par0, int nsteps, int fold) {
        Dims df = l.d;
        df.aff_nb = leap_nbv;
        df.nlive = leap_nbv == nbv_all && nbv_all != nb ? nb : 0;
        df.sp_par = 0;
        df.chunked = ts_mode;
        df.nmt = leap_nmt;                                       // the partial sums of this launch: one set per row tile of ITS shape
        const dim3 gf((unsigned)((leap_wgs + per_roles) * leap_nbv));
        // (every chain's counters and hand-off words count its OWN launches: the sub-batches of a sweep share the step numbers)
        const unsigned long long step_base = s->leap_steps, role_base = s->leap_rsteps;
        s->leap_steps += (unsigned long long)nsteps;
        s->leap_rsteps += (unsigned long long)(nsteps - (((fold & 2) && !(fold & 4)) ? 1 : 0));
        for (int sub = 0; sub < nbv_all / leap_nbv; ++sub) {
            df.b0 = l.d.b0 + sub * leap_nbv;
            void *args[] = {(void *)&df, (void *)&ctx->c, (void *)&ctx->w, (void *)&c, (void *)&s->ch, (void *)&par0, (void *)&nsteps,
                            (void *)&step_base, (void *)&role_base, (void *)&fold};
            (void)hipLaunchKernel(leap_fn(ts_mode, d0.ntc, leap_nst), gf, dim3(256), args, 0, st);
        }
    };
    const bool prof0 = s->prof_i >= 0 && (size_t)(2 * s->prof_i + 1) < s->prof_ev.size() && g == 0;
    const bool fold = leap_ok && s->hmc_fold;
    if (fold) {
        // The whole trajectory in ONE launch: gradient at the start point, the first step (momentum draw, half kick, drift:
        // k_hmc_step<0>'s work, by the chunk roles), the L-1 inner steps, gradient at the end point, and the end itself (half
        // kick, accept test, adaptation, trace: by the roles as well, or -- hmc_mode 5 -- by k_hmc_step<2> as a launch of its
        // own).  L+1 gradient evaluations, L (+1) role steps.
        if (s->vt_dirty || c.adapt_mass) {
            Dims dv = l.d;
            hipLaunchKernelGGL(k_vt, dim3(nb), dim3(WAVE), 0, st, dv, ctx->w, s->ch);
            s->vt_dirty = false;
        }
        if (prof0) (void)hipEventRecord(s->prof_ev[2 * s->prof_i], st);
        // hmc_end: the last half kick, the accept test, adaptation and trace by the chunk roles of the same launch as well;
        // otherwise (hmc_mode 5) k_hmc_step<2> closes the trajectory as a launch of its own
        launch_leap(1, c.L + 1, s->hmc_end ? 7 : 3);
        if (prof0) {
            (void)hipEventRecord(s->prof_ev[2 * s->prof_i + 1], st);
            s->prof_i += 1;
            s->prof_launches = nbv_all / leap_nbv;
            s->prof_evals = c.L + 1;
        }
        if (!s->hmc_end) {
            l.d.sp_par = c.L & 1 ? 0 : 1;       // the buffer the last role step wrote: steps alternate from buffer 1
            l.d.chunked = 0;
            l.d.nmt = leap_nmt;                 // the end point's partial sums are k_leap's: its row tiles
            launch_hmc(ctx, l, c, s->ch, 2, /*gather_qs=*/3);
            l.d.nmt = d0.nmt;
        }
    } else if (chunked && s->hmc_tailfold && s->hmc_tail && s->xcd_local && nbv_all > 0 && s->ngroups == 1 && (l.affinity & 1) &&
               xcd_affinity_applies(ntile_all, nbv_all) && !s->use_graph && (d0.ntc == 1 || d0.ntc == 6 || d0.ntc == 12) && c.L >= 3 && per_roles <= ROLE_SLOTS) {
        // The whole trajectory as L + 1 launches of k_se_chunk -- where the persistent launch does not fit the chip (16+ chains
        // at UK-380, SYN-2048) -- with the trajectory's first step (momentum draw, half kick: traj 1), the step after it (2) and
        // the last half kick (3) by the chunk roles of those launches, and the accept test, adaptation and trace by the roles'
        // own launch (k_hmc_final): what k_se, k_hmc_step<0>, ..., k_se, k_hmc_step<2> do in the stage form (hmc_mode 3), and
        // what k_leap's roles do inside the persistent launch.  Buffers alternate from 1, as there.
        if (s->vt_dirty || c.adapt_mass) {
            hipLaunchKernelGGL(k_vt, dim3(nb), dim3(WAVE), 0, st, l.d, ctx->w, s->ch);
            s->vt_dirty = false;
        }
        int par = 1;
        if (ts_mode == 2) hipLaunchKernelGGL(k_sp_prep, dim3(nb), dim3(256), 0, st, l.d, ctx->w, s->ch, par);
        if (prof0) (void)hipEventRecord(s->prof_ev[2 * s->prof_i], st);
        const int per = d0.ntc + d0.Mp / WAVE;
        Dims df = l.d;
        df.aff_nb = nbv_all;
        df.nlive = nbv_all != nb ? nb : 0;
        df.chunked = ts_mode;
        const dim3 gf((unsigned)((ntile_all + per) * nbv_all));      // tiles, then the chunk roles
        for (int it = 0; it <= c.L; ++it) {
            const int traj = it == 0 ? 1 : it == 1 ? 2 : it == c.L ? 3 : 0;
            df.sp_par = par;
            Work wf = ctx->w;
            if (it == 0) wf.Lpart = wf.Lpart0;                    // the start point's value of the S->E term: kept for the accept test
            s->tail_count += (unsigned long long)ntile_all;
            const unsigned long long target = s->tail_count;
#define LAUNCH_TAILF(TSM_, NTC_) hipLaunchKernelGGL((k_se_chunk<TSM_, NTC_>), gf, dim3(256), 0, st, df, ctx->c, wf, c, s->ch, par, target, traj)
            if (ts_mode == 1) {
                if (d0.ntc == 1) LAUNCH_TAILF(1, 1); else if (d0.ntc == 6) LAUNCH_TAILF(1, 6); else LAUNCH_TAILF(1, 12);
            } else {
                if (d0.ntc == 1) LAUNCH_TAILF(2, 1); else if (d0.ntc == 6) LAUNCH_TAILF(2, 6); else LAUNCH_TAILF(2, 12);
            }
#undef LAUNCH_TAILF
            if (it < c.L) par ^= 1;
        }
        {
            Dims dz = l.d;
            const bool aff = xcd_affinity_applies(per, nb);
            dz.aff_nb = aff ? nb : 0;
            const dim3 gz = aff ? dim3(per * nb) : dim3(per, nb);
            switch (d0.ntc) {
                case 1: hipLaunchKernelGGL(k_hmc_final<1>, gz, dim3(WAVE), 0, st, dz, ctx->c, ctx->w, c, s->ch, par); break;
                case 6: hipLaunchKernelGGL(k_hmc_final<6>, gz, dim3(WAVE), 0, st, dz, ctx->c, ctx->w, c, s->ch, par); break;
                default: hipLaunchKernelGGL(k_hmc_final<12>, gz, dim3(WAVE), 0, st, dz, ctx->c, ctx->w, c, s->ch, par); break;
            }
        }
        if (prof0) {
            (void)hipEventRecord(s->prof_ev[2 * s->prof_i + 1], st);
            s->prof_i += 1;
            s->prof_launches = c.L + 2;
            s->prof_evals = c.L + 1;
        }
    } else {
    l.d.chunked = 0;                       // k_se writes tile scalars only ahead of a chunked step
    launch_se<1>(ctx, l, true);
    l.d.chunked = ts_mode;                 // stage 0 hands the trajectory over to the chunk kernel
    launch_hmc(ctx, l, c, s->ch, 0);
    s->vt_dirty = false;                   // (k_hmc_step<0> writes Work::Vt on its way)
    if (!chunked) {
        for (int i = 1; i < c.L; ++i) {
            launch_se<1>(ctx, l, true);
            launch_hmc(ctx, l, c, s->ch, 1);
        }
    } else {
        // all inner steps 1..L-1 by independent 64-lane chunks (k_hmc_chunk); stage 2 then gathers (Q s) and
        // computes the priors and the Jacobian of the end point itself (gather_qs = 3)
        const int per = d0.ntc + d0.Mp / WAVE;
        const bool aff = (l.affinity & 1) && xcd_affinity_applies(per, nb);
        int par = 0;
        // chunk roles inside the gradient launch (k_se_chunk): 8 chains, one XCD each (checked at creation), the
        // XCD-affine grid, no graph capture in progress (the ticket counter does not care, but keep the two apart)
        const int ntile_se = d0.ntc * d0.nmt;
        // not a multiple of 8 chains: the layout of the next multiple with the missing chains' blocks retiring at once,
        // so that every chain is still whole on one XCD (nbv = chains of the layout)
        const int nbv = (nb + 7) / 8 * 8;
        const bool tail = s->hmc_tail && s->xcd_local && nbv > 0 && s->ngroups == 1 && (l.affinity & 1) && xcd_affinity_applies(ntile_se, nbv) &&
                          !s->use_graph && (d0.ntc == 1 || d0.ntc == 6 || d0.ntc == 12) && ts_mode != 0;
        // all of them in ONE persistent launch (k_leap: the tiles keep their cells in registers over the steps) when every
        // workgroup of that launch can be resident at once -- its tiles wait for the roles.  (Reached only when the folded
        // form above is switched off: the inner steps alone, between k_hmc_step<0> and k_se + k_hmc_step<2>.)
        const bool prof = prof0;
        if (prof) (void)hipEventRecord(s->prof_ev[2 * s->prof_i], st);
        const bool leap = leap_ok;
        if (leap) {
            launch_leap(par, c.L - 1, 0);
            par = (c.L - 1) & 1;
        }
        for (int i = 1; i < c.L && !leap; ++i) {
            l.d.sp_par = par;
            if (tail) {
                Dims df = l.d;
                df.aff_nb = nbv;
                df.nlive = nbv != nb ? nb : 0;
                const dim3 gf((unsigned)((ntile_se + per) * nbv));     // tiles, then the chunk roles
                s->tail_count += (unsigned long long)ntile_se;         // what a chain's counter shows once this launch's tiles are in
                const unsigned long long target = s->tail_count;
#define LAUNCH_TAIL(TSM_, NTC_) hipLaunchKernelGGL((k_se_chunk<TSM_, NTC_>), gf, dim3(256), 0, st, df, ctx->c, ctx->w, c, s->ch, par, target, 0)
                if (ts_mode == 1) {
                    if (d0.ntc == 1) LAUNCH_TAIL(1, 1); else if (d0.ntc == 6) LAUNCH_TAIL(1, 6); else LAUNCH_TAIL(1, 12);
                } else {
                    if (d0.ntc == 1) LAUNCH_TAIL(2, 1); else if (d0.ntc == 6) LAUNCH_TAIL(2, 6); else LAUNCH_TAIL(2, 12);
                }
#undef LAUNCH_TAIL
                par ^= 1;
                continue;
            }
            launch_se<1>(ctx, l, true);
            Dims dc = l.d;
            dc.aff_nb = aff ? nb : 0;
            const dim3 gc = aff ? dim3(per * nb) : dim3(per, nb);
            switch (d0.ntc) {       // chunk count at compile time for the BASELINE sizes (NI, UK, SYN), rolled loops otherwise
                case 1: hipLaunchKernelGGL(k_hmc_chunk<1>, gc, dim3(WAVE), 0, st, dc, ctx->c, ctx->w, c, s->ch, par); break;
                case 6: hipLaunchKernelGGL(k_hmc_chunk<6>, gc, dim3(WAVE), 0, st, dc, ctx->c, ctx->w, c, s->ch, par); break;
                case 12: hipLaunchKernelGGL(k_hmc_chunk<12>, gc, dim3(WAVE), 0, st, dc, ctx->c, ctx->w, c, s->ch, par); break;
                default: hipLaunchKernelGGL(k_hmc_chunk<0>, gc, dim3(WAVE), 0, st, dc, ctx->c, ctx->w, c, s->ch, par); break;
            }
            par ^= 1;
        }
        l.d.sp_par = par;
        if (prof) {
            (void)hipEventRecord(s->prof_ev[2 * s->prof_i + 1], st);
            s->prof_i += 1;
            s->prof_launches = leap ? 1 : tail ? c.L - 1 : 2 * (c.L - 1);
            s->prof_evals = c.L - 1;
        }
    }
    l.d.chunked = 0;
    launch_se<1>(ctx, l, true);
    launch_hmc(ctx, l, c, s->ch, 2, /*gather_qs=*/chunked ? 3 : 0);
    }   // !fold
    // [part 1] MultiScan(n_scans, Gibbs[move S->E, move E->I, occult S->E, occult E->I]):
    // per update [finalize previous | propose] then the log-ratio over the touched cells
    Dims d = l.d;
    int advanced = 0, fpend_in_record = 0, recorded = 0;
    {
        const bool aff = (l.affinity & 2) && xcd_affinity_applies(c.nrb_d, nb);
        d.aff_nb = aff ? nb : 0;
        const dim3 gm = aff ? dim3(c.nrb_d * nb) : dim3(c.nrb_d, nb);
        const size_t plds = k_move_pa2_lds_bytes(d);
        // 64-day chunks a row of the series is held in by the proposing wave (moves_kernel.h; sampler_create caps T at 1024)
        const int nch = d.Tp <= 6 * WAVE ? 6 : d.Tp <= 12 * WAVE ? 12 : 16;
        auto pair_fn = nch == 6 ? k_move_pair<6> : nch == 12 ? k_move_pair<12> : k_move_pair<16>;
        auto pa2_fn = nch == 6 ? k_move_pa2<6> : nch == 12 ? k_move_pa2<12> : k_move_pa2<16>;
        auto pairs_fn = nch == 6 ? k_move_pairs<6> : nch == 12 ? k_move_pairs<12> : k_move_pairs<16>;
        const size_t plds_pairs = k_move_pairs_lds_bytes(d);
        if (s->pairs_lds_attr == 0)
            s->pairs_lds_attr = hipFuncSetAttribute((const void *)pairs_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)plds_pairs) == hipSuccess ? 1 : -1;
        if (plds > 64 * 1024 && !s->move_lds_attr) {
            (void)hipFuncSetAttribute((const void *)pair_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
            (void)hipFuncSetAttribute((const void *)pa2_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
            s->move_lds_attr = true;
        }
        int have_prev = 0, pbuf = 0;
        if (s->moves_mode != 1 && c.n_scans > 30) s->moves_mode = 1;   // k_move_pair's launch tokens cover 62 launches per sweep
        if (s->moves_mode != 1) {
            // paired form: [finalize pending E->I-type | whole S->E-type update | propose E->I-type], then
            // the log-ratio of the E->I-type proposal over its band: 4 launches per scan -- or 2, with the band
            // evaluated by more workgroups of the pair launch itself (XCD-local hand-off, see pair_band_block): 8
            // chains on one stream, one XCD each (checked at creation), every workgroup resident at once
            const int npairs = 2 * c.n_scans;
            int have_pre = 0;
            int cus = 0;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
            const int nbv = (nb + 7) / 8 * 8;                            // as for k_se_chunk: the layout's chains
            // 16 rows per band workgroup, two per wave; 32 (four per wave) where that is what lets every workgroup of the launch
            // hold a CU: sixteen chains at UK-380 are (3 + 12) x 16 = 240 workgroups
            int nband_fit = (d.M + 15) / 16;
            if ((3 + nband_fit) * nbv > cus && (3 + (d.M + 31) / 32) * nbv <= cus) nband_fit = (d.M + 31) / 32;
            const bool band_in_pair = s->moves_mode != 3 && s->xcd_local && nbv > 0 && s->ngroups == 1 && !s->use_graph &&
                                      (3 + nband_fit) * nbv <= cus;
            const int nbk = band_in_pair ? nbv : nb;
            Dims dp = d;
            dp.nlive = band_in_pair && nbv != nb ? nb : 0;
            const int nband = band_in_pair ? nband_fit : 0;
            SamplerCfg cp = c;
            if (band_in_pair) cp.nrb_d = nband;                          // the band's partial sums: one pair per band workgroup
            // every pair of the sweep and the closing step in ONE launch (k_move_pairs): the grid of a pair launch with band
            // workgroups, resident for the whole sweep -- under the same conditions
            const bool persistent = band_in_pair && s->moves_mode == 0 && npairs > 0 && s->pairs_lds_attr == 1;
            if (persistent) {
                // (its closing step also does what k_apply_fpend / k_record are launched for in the other forms)
                hipLaunchKernelGGL(pairs_fn, dim3((3 + nband) * nbk), dim3(MVB), plds_pairs, st, dp, ctx->c, ctx->w, cp, s->ch, npairs, 1, nbk,
                                   s->pair_debug, nband, s->pbar_count, s->record_events ? 3 : 1);
                s->pbar_count += (unsigned)(npairs * (3 + nband));           // what every live chain's counter shows after this launch
                advanced = 1;
                recorded = 1;
            }
            for (int scan = 0; scan < (persistent ? 0 : c.n_scans); ++scan)
                for (int half = 0; half < 2; ++half) {
                    const int pair = 2 * scan + half;
                    const MoveSpec se{half, 0, 2 * half, scan}, nx{half, 1, 2 * half + 1, scan};
                    // a third role pre-draws the S->E-type proposal of the next pair (same sweep)
                    const bool pre = (s->moves_mode == 0 || s->moves_mode == 3 || s->moves_mode == 4) && pair + 1 < npairs;
                    const int nh = (half + 1) & 1, nscan = scan + (half == 1 ? 1 : 0);
                    const MoveSpec se_next = pre ? MoveSpec{nh, 0, 2 * nh, nscan} : MoveSpec{-1, 0, 0, 0};
                    hipLaunchKernelGGL(pair_fn, dim3(((pre ? 3 : 2) + nband) * nbk), dim3(MVB), plds, st, dp, ctx->c, ctx->w, cp, s->ch,
                                       se, nx, se_next, have_prev, have_pre, pbuf, nbk, pair, s->pair_debug, nband);
                    have_pre = pre ? 1 : 0;
                    pbuf ^= 1;
                    if (!band_in_pair)
                        hipLaunchKernelGGL((k_move_delta<false>), gm, dim3(DELTA_THREADS), 0, st, d, ctx->c, ctx->w, c, s->ch, pbuf, 1);
                    have_prev = 1;
                }
            if (have_prev && !persistent) {
                const MoveSpec none{-1, 0, 0, 0}, close{-2, 0, 0, 0};
                hipLaunchKernelGGL(pair_fn, dim3(nb), dim3(MVB), plds, st, d, ctx->c, ctx->w, cp, s->ch, none, close, none, 1,
                                   0, pbuf, nb, 62, 0, 0);
                // the F band of the last accepted E->I update: by k_record's waves when it runs anyway
                if (s->record_events) fpend_in_record = 1;
                else hipLaunchKernelGGL(k_apply_fpend, gm, dim3(256), 0, st, d, ctx->c, ctx->w, c, s->ch);
                advanced = 1;
            }
        } else {
            for (int scan = 0; scan < c.n_scans; ++scan)
                for (int slot = 0; slot < 4; ++slot) {
                    const MoveSpec spec{slot >= 2 ? 1 : 0, slot & 1, slot, scan};
                    hipLaunchKernelGGL(pa2_fn, gm, dim3(MVB), plds, st, d, ctx->c, ctx->w, c, s->ch, spec,
                                       have_prev, pbuf);
                    pbuf ^= 1;
                    hipLaunchKernelGGL((k_move_delta<true>), gm, dim3(DELTA_THREADS), 0, st, d, ctx->c, ctx->w, c, s->ch, pbuf, 0);
                    have_prev = 1;
                }
            if (have_prev) {
                // closing launch: finalize the last proposal and advance the sweep counter
                const MoveSpec none{-2, 0, 0, 0};
                hipLaunchKernelGGL(pa2_fn, gm, dim3(MVB), plds, st, d, ctx->c, ctx->w, c, s->ch, none, 1, pbuf);
                advanced = 1;
            }
        }
    }
    d.aff_nb = 0;
    if (s->record_events && !recorded)
        hipLaunchKernelGGL(k_record, dim3((d.M + 3) / 4, nb), dim3(256), 0, st, d, ctx->c, ctx->w, c, s->ch, advanced,
                           fpend_in_record);
    if (!advanced) hipLaunchKernelGGL(k_advance, dim3((nb + 63) / 64), dim3(64), 0, st, s->ch, b0, nb);
}

static int check_handoffs(seir_sampler *s);
extern "C" int seir_sampler_run(seir_sampler *s, int32_t n) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (!s->have_state) return fail(SEIR_ERR_STATE, "no chain state set");
    if (n < 0) return fail(SEIR_ERR_INVALID, "num_sweeps must be >= 0");
    if (s->poisoned) return check_handoffs(s);       // sticky: see there
    hipStream_t main_st = s->ctx->stream;
    // fork: every group stream starts after what is already queued on the context stream
    HIP_TRY(hipEventRecord(s->ev_fork, main_st));
    for (int g = 0; g < s->ngroups; ++g) {
        hipStream_t st = s->gstream[g];
        HIP_TRY(hipStreamWaitEvent(st, s->ev_fork, 0));
        if (s->use_graph && (s->graph_skew != s->ctx->opt_skew || s->graph_aff != s->ctx->opt_affinity)) {
            drop_graph(s);                           // launch options are baked into the captured kernels
            s->graph_skew = s->ctx->opt_skew; s->graph_aff = s->ctx->opt_affinity;
        }
        if (s->use_graph && !s->gexec[g]) {
            HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            enqueue_sweep(s, g);
            HIP_TRY(hipStreamEndCapture(st, &s->graph[g]));
            HIP_TRY(hipGraphInstantiate(&s->gexec[g], s->graph[g], nullptr, nullptr, 0));
        }
    }
    for (int i = 0; i < n; ++i)
        for (int g = 0; g < s->ngroups; ++g) {
            if (s->use_graph) HIP_TRY(hipGraphLaunch(s->gexec[g], s->gstream[g]));
            else enqueue_sweep(s, g);
        }
    // join: the context stream continues after every group has finished
    for (int g = 0; g < s->ngroups; ++g) {
        HIP_TRY(hipEventRecord(s->ev_join[g], s->gstream[g]));
        HIP_TRY(hipStreamWaitEvent(main_st, s->ev_join[g], 0));
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// 16-bit event trace: a count that did not fit was truncated on the device -- fail loudly
// A hand-off inside a launch that timed out (k_move_pair's handshakes and band tokens, k_se_chunk's tile counter) means
// a workgroup went on without what it waited for: never seen outside the test hooks, and then the draws are not to be
// trusted -- the next read of the trace fails loudly instead of delivering them.
static int check_handoffs(seir_sampler *s) {
    if (s->pair_debug != 0) return 0;                     // the hooks make roles late on purpose
    // only the waits a workgroup cannot recover from (band tokens, k_se_chunk's tile flag, k_leap's flags, k_move_pairs'
    // step barrier); a late speculative role of k_move_pair is benign -- role 0 draws the proposal itself, the traces
    // are the same -- and only counted
    if (!s->poisoned) {
        std::vector<uint32_t> late((size_t)s->cfg.B, 0u);
        uint32_t *fatal = s->ch.late + s->ch.late_fatal;
        HIP_TRY(hipMemcpy(late.data(), fatal, sizeof(uint32_t) * late.size(), hipMemcpyDeviceToHost));
        for (size_t b = 0; b < late.size(); ++b)
            if (late[b]) { s->poisoned = true; s->poison_chain = (int)b; s->poison_count = late[b]; break; }
    }
    if (s->poisoned)
        // STICKY: a workgroup that gave up a wait went on with stale data (a band workgroup with an old F-band descriptor:
        // Work::F is only ever updated incrementally and would stay out of step with the event planes), so nothing this
        // sampler produces is to be trusted until its state is rebuilt -- seir_sampler_restore (back to the last snapshot:
        // the failed burst can be run again, e.g. in the per-step launch forms), seir_sampler_set_state or
        // seir_sampler_refresh (F and every table recomputed from the planes as they are; the failed burst's draws are lost)
        return fail(SEIR_ERR_STATE, "chain %d: %u in-launch hand-off(s) timed out -- the persistent launches could not get all "
                    "their workgroups on the GPU at once (another sampler or process holds part of it?).  Draws since the last "
                    "check are unreliable and the sampler refuses to go on until seir_sampler_restore / _set_state / _refresh; "
                    "hmc_mode 3 + moves_mode 4 (seir_sampler_set_launch_form) are the launch forms for a shared GPU",
                    s->poison_chain, s->poison_count);
    return 0;
}

static int check_ev_overflow(seir_sampler *s) {
    int rc = check_handoffs(s);
    if (rc) return rc;
    if (!s->cfg.ev16) return 0;
    unsigned flag = 0;
    HIP_TRY(hipMemcpy(&flag, s->ch.ev_overflow, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) return fail(SEIR_ERR_STATE, "an event count exceeded 65535: record_events=2 (uint16 trace) cannot hold this chain");
    return 0;
}

extern "C" int seir_sampler_read_trace(seir_sampler *s, int32_t first, int32_t count, double *theta,
                                       void *events, double *hmc, double *moves) {
    int rc = sampler_check(s);
    if (rc) return rc;
    const Dims &d = s->ctx->d;
    const SamplerCfg &c = s->cfg;
    if (first < 0 || count < 0 || first + count > c.cap)
        return fail(SEIR_ERR_INVALID, "trace range [%d,%d) outside capacity %d", first, first + count, c.cap);
    if (events && !s->record_events) return fail(SEIR_ERR_STATE, "sampler was created with record_events=0");
    hipStream_t st = s->ctx->stream;
    const size_t B = c.B;
    if (theta)
        HIP_TRY(hipMemcpyAsync(theta, s->ch.tr_theta + (size_t)first * B * d.P, sizeof(double) * count * B * d.P,
                               hipMemcpyDeviceToHost, st));
    if (events)
        HIP_TRY(hipMemcpyAsync(events, (const char *)s->ch.tr_events + (size_t)first * B * d.M * d.T * 3 * (c.ev16 ? 2 : 4),
                               (size_t)(c.ev16 ? 2 : 4) * count * B * d.M * d.T * 3, hipMemcpyDeviceToHost, st));
    if (hmc)
        HIP_TRY(hipMemcpyAsync(hmc, s->ch.tr_hmc + (size_t)first * B * 3, sizeof(double) * count * B * 3,
                               hipMemcpyDeviceToHost, st));
    if (moves)
        HIP_TRY(hipMemcpyAsync(moves, s->ch.tr_mv + (size_t)first * B * 4 * NMVTR,
                               sizeof(double) * count * B * 4 * NMVTR, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return check_ev_overflow(s);
}

extern "C" int seir_sampler_read_trace_async(seir_sampler *s, int32_t first, int32_t count, double *theta,
                                             void *events, double *hmc, double *moves) {
    int rc = sampler_check(s);
    if (rc) return rc;
    const Dims &d = s->ctx->d;
    const SamplerCfg &c = s->cfg;
    if (first < 0 || count < 0 || first + count > c.cap)
        return fail(SEIR_ERR_INVALID, "trace range [%d,%d) outside capacity %d", first, first + count, c.cap);
    if (events && !s->record_events) return fail(SEIR_ERR_STATE, "sampler was created with record_events=0");
    // the copies start when everything queued on the context stream so far (the burst) is done; what is
    // queued there afterwards overlaps them
    HIP_TRY(hipEventRecord(s->ev_burst, s->ctx->stream));
    hipStream_t st = s->copy_stream;
    HIP_TRY(hipStreamWaitEvent(st, s->ev_burst, 0));
    const size_t B = c.B;
    if (theta)
        HIP_TRY(hipMemcpyAsync(theta, s->ch.tr_theta + (size_t)first * B * d.P, sizeof(double) * count * B * d.P,
                               hipMemcpyDeviceToHost, st));
    if (events)
        HIP_TRY(hipMemcpyAsync(events, (const char *)s->ch.tr_events + (size_t)first * B * d.M * d.T * 3 * (c.ev16 ? 2 : 4),
                               (size_t)(c.ev16 ? 2 : 4) * count * B * d.M * d.T * 3, hipMemcpyDeviceToHost, st));
    if (hmc)
        HIP_TRY(hipMemcpyAsync(hmc, s->ch.tr_hmc + (size_t)first * B * 3, sizeof(double) * count * B * 3,
                               hipMemcpyDeviceToHost, st));
    if (moves)
        HIP_TRY(hipMemcpyAsync(moves, s->ch.tr_mv + (size_t)first * B * 4 * NMVTR,
                               sizeof(double) * count * B * 4 * NMVTR, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(s->ev_copy, st));
    s->copy_pending = true;
    return 0;
}

extern "C" int seir_sampler_trace_wait(seir_sampler *s) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (s->copy_pending) {
        HIP_TRY(hipEventSynchronize(s->ev_copy));
        s->copy_pending = false;
        return check_ev_overflow(s);
    }
    return 0;
}

extern "C" int seir_host_alloc(void **p, uint64_t bytes) {
    if (!p) return fail(SEIR_ERR_INVALID, "null pointer");
    HIP_TRY(hipHostMalloc(p, bytes ? bytes : 8, hipHostMallocDefault));
    return 0;
}
extern "C" int seir_host_free(void *p) {
    HIP_TRY(hipHostFree(p));
    return 0;
}

extern "C" int seir_sampler_time_grad_kernel(seir_sampler *s, int32_t iters, float *mean_ms) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (!s->have_state) return fail(SEIR_ERR_STATE, "no chain state set");
    if (!mean_ms || iters < 1) return fail(SEIR_ERR_INVALID, "bad iters/mean_ms");
    seir_ctx *ctx = s->ctx;
    LaunchCfg l = whole(ctx, s->cfg.B);
    l.d.chunked = (s->hmc_chunked && s->cfg.L >= 3 && l.d.ntc <= CT_MAXC) ? (l.d.Mp <= 512 ? 1 : 2) : 0;   // as in the sweep
    launch_se<1>(ctx, l, true);
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; ++i) launch_se<1>(ctx, l, true);
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    HIP_TRY(hipGetLastError());
    *mean_ms = ms / iters;
    return 0;
}

extern "C" int seir_sampler_time_leapfrog(seir_sampler *s, int32_t sweeps, float *mean_ms, int32_t *launches, int32_t *evals) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (!s->have_state) return fail(SEIR_ERR_STATE, "no chain state set");
    if (!mean_ms || sweeps < 1 || sweeps > 4096) return fail(SEIR_ERR_INVALID, "bad sweeps/mean_ms");
    if (s->use_graph || s->ngroups != 1) return fail(SEIR_ERR_STATE, "timing of the leapfrog section needs stream launches on one stream");
    while (s->prof_ev.size() < (size_t)2 * sweeps) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        s->prof_ev.push_back(e);
    }
    s->prof_i = 0;
    s->prof_launches = 0;
    rc = seir_sampler_run(s, sweeps);
    const int recorded = s->prof_i;
    s->prof_i = -1;
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    if (recorded < 1) return fail(SEIR_ERR_STATE, "this sampler's sweep has no chunked leapfrog section (hmc_mode 1 or fewer than 3 leapfrog steps)");
    double sum = 0.0;
    for (int i = 0; i < recorded; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->prof_ev[2 * i], s->prof_ev[2 * i + 1]));
        sum += ms;
    }
    *mean_ms = (float)(sum / recorded);
    if (launches) *launches = s->prof_launches;
    if (evals) *evals = s->prof_evals;
    return 0;
}

#ifdef SE_STAMPS
extern "C" int seir_debug_read_ts(seir_ctx *ctx, double *out, int64_t n) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out, ctx->w.TS, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}
#endif

#if defined(LEAP_STAMPS) || defined(PAIR_STAMPS)
// [B][16][8] stamps of k_leap; reset = 1: min-words to ~0, max-words to 0 (call before the sweep to look at)
extern "C" int seir_debug_leap_stamps(seir_sampler *s, unsigned long long *out, int reset) {
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    const size_t n = (size_t)(s->cfg.B + 2) * 16 * 8 + 4096;
    HIP_TRY(hipMemcpy(out, s->ch.leap_st, n * 8, hipMemcpyDeviceToHost));
    if (reset) {
        std::vector<unsigned long long> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (i & 1) ? 0ull : ~0ull;
        HIP_TRY(hipMemcpy(s->ch.leap_st, h.data(), n * 8, hipMemcpyHostToDevice));
    }
    return 0;
}
#endif

#ifdef TAIL_STAMPS
// words 8..15 of every chain's counter line; reset = 1: min-words to ~0, max-words to 0 (call before the launch to look at)
extern "C" int seir_debug_tail_stamps(seir_sampler *s, unsigned long long *out, int reset) {
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    const int B = s->cfg.B;
    std::vector<unsigned long long> h((size_t)B * TAIL_STRIDE);
    HIP_TRY(hipMemcpy(h.data(), s->ch.tail, h.size() * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < 8; ++k) out[b * 8 + k] = h[(size_t)b * TAIL_STRIDE + 8 + k];
    if (reset) {
        for (int b = 0; b < B; ++b) {
            unsigned long long *p = h.data() + (size_t)b * TAIL_STRIDE + 8;
            p[0] = p[2] = p[5] = ~0ull; p[1] = p[3] = p[4] = 0;
        }
        HIP_TRY(hipMemcpy(s->ch.tail, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    }
    return 0;
}
#endif

extern "C" int seir_sampler_pair_timeouts(seir_sampler *s, uint32_t *out) {
    int rc = sampler_check(s);
    if (rc) return rc;
    if (!out) return fail(SEIR_ERR_INVALID, "null pointer");
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    std::vector<uint32_t> both((size_t)2 * s->cfg.B, 0u);
    HIP_TRY(hipMemcpy(both.data(), s->ch.late, sizeof(uint32_t) * both.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < s->cfg.B; ++b) out[b] = both[b] + both[(size_t)s->cfg.B + b];
    return 0;
}

#ifdef SEIR_STAMPS
extern "C" int seir_sampler_debug_hs(seir_sampler *s, double *out) {
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    HIP_TRY(hipMemcpy(out, s->ch.hs, sizeof(double) * s->cfg.B * NHS, hipMemcpyDeviceToHost));
    return 0;
}
#endif
