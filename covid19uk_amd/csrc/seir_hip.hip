// libseirhip.so -- C-ABI (include/seir_hip.h) over the gfx950 kernels.
// Host side of the drop-in boundary for joint_log_prob
// (covid19uk/inference/inference.py:537-557).  No CPU fallback anywhere in
// this file: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/seir_hip.h"
#include "logprob_kernels.h"

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
};

static inline int ceil_to(int x, int q) { return (x + q - 1) / q * q; }

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

extern "C" void seir_destroy(seir_ctx *ctx) {
    if (!ctx) return;
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
    d.Mp = ceil_to(M, 16);
    d.Tp = ceil_to(T, 64);
    d.Kp = ceil_to(M, 4);
    d.P = 6 + (T - 1) + M;
    d.nrb_scan = (M + SCAN_ROWS - 1) / SCAN_ROWS;
    d.nrb_se = (M + SE_ROWS - 1) / SE_ROWS;
    d.nu = ds->nu; d.dt = ds->time_delta; d.rate_floor = ds->rate_floor;
    d.car_half_logdet = ds->car_half_logdet;
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
    std::vector<double> Cs((size_t)d.Mp * d.Kp, 0.0), N(d.Mp, 1.0), invN(d.Mp, 0.0), la(d.Mp, 0.0),
        W(d.Tp, 0.0), wd(d.Tp, 0.0), init((size_t)d.Mp * 4, 0.0);
    for (int m = 0; m < M; ++m) {
        for (int j = 0; j < M; ++j) Cs[(size_t)m * d.Kp + j] = ds->Cstar[(size_t)m * M + j];
        if (!(ds->N[m] > 0.0)) return fail(SEIR_ERR_INVALID, "N[%d] must be positive", m);
        N[m] = ds->N[m];
        invN[m] = 1.0 / ds->N[m];
        la[m] = ds->log_area_c[m];
        for (int s = 0; s < 4; ++s) init[(size_t)m * 4 + s] = ds->init_state[(size_t)m * 4 + s];
    }
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
    int rc;
    if ((rc = dev_upload(ctx, &ctx->c.Cstar, Cs))) return rc;
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
    if ((rc = dev_alloc(ctx, &w.Lpart, (size_t)B * d.nrb_se))) return rc;
    if ((rc = dev_alloc(ctx, &w.Ppart, (size_t)B * d.nrb_se))) return rc;
    if ((rc = dev_alloc(ctx, &w.Kpart, (size_t)B * d.nrb_se * d.Tp))) return rc;
    if ((rc = dev_alloc(ctx, &w.Rsum, (size_t)B * d.Mp))) return rc;

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
    if (ds->T > 2048) return fail(SEIR_ERR_INVALID, "T=%d exceeds the supported 2048 days", ds->T);
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

static int check_batch(seir_ctx *ctx, int B) {
    if (!ctx) return fail(SEIR_ERR_INVALID, "null context");
    if (B < 1 || B > ctx->Bmax) return fail(SEIR_ERR_INVALID, "B=%d outside [1, max_chains=%d]", B, ctx->Bmax);
    HIP_TRY(hipSetDevice(ctx->device));
    return 0;
}

// --- individual launches ---------------------------------------------------
static void launch_scan(seir_ctx *ctx, int B, const double *events) {
    const Dims &d = ctx->d;
    hipLaunchKernelGGL(k_scan, dim3(d.nrb_scan, B), dim3(256), (size_t)4 * d.Tp * 2 * sizeof(double),
                       ctx->stream, d, ctx->c, ctx->w, events);
}
static void launch_gemm(seir_ctx *ctx, int B) {
    const Dims &d = ctx->d;
    const int ntt = d.Tp / 16, per_wg = 4 * GEMM_TT;
    hipLaunchKernelGGL(k_gemm, dim3((ntt + per_wg - 1) / per_wg, d.Mp / 16, B), dim3(256),
                       (size_t)16 * gemm_lda(d.Kp) * sizeof(double), ctx->stream, d, ctx->c, ctx->w);
}
static void launch_params(seir_ctx *ctx, int B, const double *u) {
    hipLaunchKernelGGL(k_params, dim3(B), dim3(256), 0, ctx->stream, ctx->d, ctx->c, ctx->w, u);
}
static void launch_se(seir_ctx *ctx, int B, bool grad) {
    const Dims &d = ctx->d;
    if (grad)
        hipLaunchKernelGGL(k_se<true>, dim3(d.nrb_se, B), dim3(256), (size_t)4 * d.Tp * sizeof(double),
                           ctx->stream, d, ctx->c, ctx->w);
    else
        hipLaunchKernelGGL(k_se<false>, dim3(d.nrb_se, B), dim3(256), 0, ctx->stream, d, ctx->c, ctx->w);
}
static void launch_finish(seir_ctx *ctx, int B, const double *u, double *logp, double *grad) {
    if (grad)
        hipLaunchKernelGGL(k_finish<true>, dim3(B), dim3(256), 0, ctx->stream, ctx->d, ctx->c, ctx->w, u, logp,
                           grad);
    else
        hipLaunchKernelGGL(k_finish<false>, dim3(B), dim3(256), 0, ctx->stream, ctx->d, ctx->c, ctx->w, u, logp,
                           grad);
}

extern "C" int seir_prepare_events_dev(seir_ctx *ctx, int32_t B, const double *events_dev) {
    int rc = check_batch(ctx, B);
    if (rc) return rc;
    if (!events_dev) return fail(SEIR_ERR_INVALID, "null events pointer");
    launch_scan(ctx, B, events_dev);
    launch_gemm(ctx, B);
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
    launch_params(ctx, B, u_dev);
    launch_se(ctx, B, grad_dev != nullptr);
    launch_finish(ctx, B, u_dev, logp_dev, grad_dev);
    HIP_TRY(hipGetLastError());
    ctx->last_u = u_dev; ctx->last_logp = logp_dev; ctx->last_grad = grad_dev;
    return 0;
}

extern "C" int seir_log_prob_dev(seir_ctx *ctx, int32_t B, const double *u_dev, const double *events_dev,
                                 double *logp_dev, double *grad_dev) {
    int rc = seir_prepare_events_dev(ctx, B, events_dev);
    if (rc) return rc;
    return seir_eval_prepared_dev(ctx, B, u_dev, logp_dev, grad_dev);
}

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
    return 0;
}

extern "C" int seir_log_prob(seir_ctx *ctx, int32_t B, const double *u, const double *events, double *logp) {
    return host_eval(ctx, B, u, events, logp, nullptr);
}

extern "C" int seir_log_prob_grad(seir_ctx *ctx, int32_t B, const double *u, const double *events, double *logp,
                                  double *grad) {
    if (!grad) return fail(SEIR_ERR_INVALID, "null grad pointer");
    return host_eval(ctx, B, u, events, logp, grad);
}

extern "C" int seir_sync(seir_ctx *ctx) {
    if (!ctx) return fail(SEIR_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
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
    if (which == SEIR_K_SE_GRAD && !ctx->last_grad)
        return fail(SEIR_ERR_STATE, "last evaluation had no gradient buffer");
    auto once = [&]() {
        switch (which) {
            case SEIR_K_SCAN: launch_scan(ctx, B, ctx->last_events); break;
            case SEIR_K_GEMM: launch_gemm(ctx, B); break;
            case SEIR_K_SE_VALUE: launch_se(ctx, B, false); break;
            case SEIR_K_SE_GRAD: launch_se(ctx, B, true); break;
            default: launch_finish(ctx, B, ctx->last_u, ctx->last_logp, ctx->last_grad); break;
        }
    };
    if (which < SEIR_K_SCAN || which > SEIR_K_FINISH) return fail(SEIR_ERR_INVALID, "unknown kernel id %d", which);
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
