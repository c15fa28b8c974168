/* seir_hip.h -- C-ABI of libseirhip.so, the MI355X (gfx950) implementation of the
 * covid19uk spatial SEIR posterior hot path.
 *
 * The reference has no FFI for this path: the seam is the Python callable
 *     joint_log_prob(unconstrained_params[P], events[M,T,3]) -> scalar
 * (covid19uk/inference/inference.py:537-557) handed to the Gibbs/HMC/MH kernels
 * (inference.py:97-101, mcmc_kernel_factory.py:14-168), and below it
 *     DiscreteTimeStateTransitionModel(...).log_prob(events)
 * (covid19uk/model_spec.py:278-285).  Each entry point below names the
 * reference interface it stands in for.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success and a negative seir_status on
 *     failure; seir_last_error() then returns a thread-local message;
 *   - the caller owns every host buffer, nothing is retained past the call;
 *   - a context owns its device buffers and one HIP stream, is bound to one
 *     device and is not thread-safe;
 *   - no exceptions, no global mutable state, no torch types;
 *   - layouts are the reference's: events[B][M][T][3] (M-major, fp64 counts),
 *     u[B][P] with P = 6 + (T-1) + M ordered psi, sigma_space, beta_area,
 *     gamma0, gamma1, alpha_0, alpha_t[T-1], spatial_effect[M]
 *     (inference.py:541-552); psi and sigma_space are unconstrained by
 *     softplus + eps (inference.py:525-535).
 *   - there is NO CPU fallback: without a HIP device seir_create fails.
 */
#ifndef SEIR_HIP_H
#define SEIR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEIR_ABI_VERSION 3

typedef enum {
    SEIR_OK = 0,
    SEIR_ERR_INVALID = -1,   /* bad argument / shape */
    SEIR_ERR_DEVICE = -2,    /* HIP runtime failure (no device, OOM, launch error) */
    SEIR_ERR_STATE = -3      /* call made in the wrong state */
} seir_status;

typedef struct seir_ctx seir_ctx;
typedef struct seir_sampler seir_sampler;

/* Everything `seir()`'s closure holds that does not depend on parameters
 * (model_spec.py:216-230), the CAR precision of spatial_effect() (:171-175),
 * the initial state handed to CovidUK (:139) and the constants of :22-26.
 * All pointers are host pointers, copied during seir_create. */
typedef struct {
    int32_t M;                 /* metapopulations (LADs) */
    int32_t T;                 /* time steps (days) */
    int32_t max_chains;        /* largest batch B any later call will use */
    int32_t device;            /* HIP device ordinal */
    const double *Cstar;       /* [M*M] row-major: C+C^T, diag = -colsum(C)  (:216-219) */
    const double *N;           /* [M] population */
    const double *W;           /* [T] commute volume (:221) */
    const double *weekday_c;   /* [T] centred weekday (:224-225) */
    const double *log_area_c;  /* [M] centred log(area/1e8) (:228-230) */
    const double *car_Q;       /* [M*M] D_w - 0.25 W_adj (:172-175) */
    double car_half_logdet;    /* 0.5*logdet(car_Q) */
    const double *init_state;  /* [M*4] S,E,I,R at step 0 (inference.py:511) */
    double nu;                 /* E->I rate, 0.28 (:26) */
    double time_delta;         /* 1.0 (:25) */
    double rate_floor;         /* 1e-9 (:264-266) */
} seir_desc;

int seir_abi_version(void);
const char *seir_last_error(void);

/* CovidUK(covariates, initial_state, initial_step=0, num_steps=T)  (model_spec.py:139) */
int seir_create(const seir_desc *desc, seir_ctx **out);
void seir_destroy(seir_ctx *ctx);
int seir_num_params(const seir_ctx *ctx);          /* P */

/* Replace the initial state S,E,I,R [M*4] of the context (CovidUK's `initial_state`, model_spec.py:139;
 * inference.py:511).  Host pointer; ordered on the context stream.  Used by the T-sharded evaluation
 * (SURVEY.md 8e): a shard's state at its first day follows from the events of the shards before it.
 * With a sampler attached, call seir_sampler_set_state / _refresh afterwards. */
int seir_set_initial_state(seir_ctx *ctx, const double *init_state);

/* joint_log_prob(unconstrained_params, events) for a batch of B chains
 * (inference.py:537-557).  Host pointers; blocking. */
int seir_log_prob(seir_ctx *ctx, int32_t B, const double *u, const double *events,
                  double *logp /* [B] */);

/* The same value plus d/du, which the reference obtains by TF autodiff inside
 * PreconditionedHamiltonianMonteCarlo (mcmc_kernel_factory.py:21-27). */
int seir_log_prob_grad(seir_ctx *ctx, int32_t B, const double *u, const double *events,
                       double *logp /* [B] */, double *grad /* [B*P] */);

/* Device-pointer form of the two calls above: asynchronous on the context's
 * stream, no host copies.  grad_dev may be NULL (value only). */
int seir_log_prob_dev(seir_ctx *ctx, int32_t B, const double *u_dev, const double *events_dev,
                      double *logp_dev, double *grad_dev);

/* Stage split used when many parameter vectors are evaluated against the same
 * events (the 16 leapfrogs of one HMC step): `prepare` runs the parameter-free
 * part (state scan, binomial coefficients, mobility contraction F = Cstar.I/N),
 * `eval_prepared` only the parameter-dependent part. */
int seir_prepare_events_dev(seir_ctx *ctx, int32_t B, const double *events_dev);
int seir_eval_prepared_dev(seir_ctx *ctx, int32_t B, const double *u_dev,
                           double *logp_dev, double *grad_dev);

int seir_sync(seir_ctx *ctx);
void *seir_stream(seir_ctx *ctx);                  /* hipStream_t of the context */

/* Launch options of a context (no reference counterpart; the first two do not change a result):
 *   SEIR_OPT_DEBUG_SKEW      0 off; 1..3: test hook, a pseudo-random third of the workgroups of every
 *                            launch starts ~30 us late (results must not depend on workgroup timing)
 *   SEIR_OPT_XCD_AFFINITY    bit 0 gradient kernel, bit 1 event-update kernels: chain <-> XCD affine block
 *                            mapping (default 3); speed only
 *   SEIR_OPT_GEMM_F32        1: the mobility contraction F = Cstar . I/N (model_spec.py:262 for all days) with fp32
 *                            operands on v_mfma_f32_32x32x2_f32 instead of the fp64 matrix instruction (BASELINE
 *                            config 5).  THIS ONE CHANGES RESULTS: F carries ~1e-7 relative error, the log-prob
 *                            ~1e-8 -- outside the 1e-9 the fp64 path is held to; off by default.  Needs
 *                            ceil64(M) and ceil64(T) to be multiples of 128.
 *   SEIR_OPT_EVAL_FORM       launch form of seir_log_prob_dev (speed only): 0 (default) = the S->E term evaluated on the
 *                            contraction's accumulators and the row constants beside the matrix-core tiles, as ONE
 *                            launch for a batch of 8 or 16 chains (a multiple of 8 whose tile workgroups all fit the chip) on a
 *                            GPU that places block ids congruent mod 8 on one XCD each (state, tiles and reduction hand
 *                            over through that XCD's L2), as three launches otherwise; 1 = the four-launch form (scan, contraction, S->E tiles, reduction); 2 = always
 *                            three launches.  0 and 2 give the same bits
 * Options are read when a launch is enqueued (for a sampler using graph replay: at capture). */
enum { SEIR_OPT_DEBUG_SKEW = 0, SEIR_OPT_XCD_AFFINITY = 1, SEIR_OPT_GEMM_F32 = 2, SEIR_OPT_EVAL_FORM = 3 };
int seir_set_option(seir_ctx *ctx, int32_t option, int32_t value);

/* Device memory helpers so that a ctypes host can keep inputs resident
 * without torch (torch tensors' data_ptr() work just as well). */
int seir_malloc(void **dev_ptr, uint64_t bytes);
int seir_free(void *dev_ptr);
int seir_memcpy_h2d(void *dst_dev, const void *src_host, uint64_t bytes);
int seir_memcpy_d2h(void *dst_host, const void *src_dev, uint64_t bytes);

/* HIP-event timing on the context's stream (torch.cuda.Event only sees
 * torch's own streams).  start/stop bracket whatever was enqueued between
 * them; stop blocks and returns milliseconds. */
int seir_timer_start(seir_ctx *ctx);
int seir_timer_stop(seir_ctx *ctx, float *ms);

/* Per-kernel timing: launches kernel `which` `iters` times back to back on the
 * context stream with the arguments of the last evaluation and returns the
 * mean launch duration in milliseconds (HIP events). */
enum { SEIR_K_SCAN = 0, SEIR_K_GEMM = 1, SEIR_K_SE_VALUE = 2, SEIR_K_SE_GRAD = 3,
       SEIR_K_FINISH = 4,
       /* the launches of the fused form (SEIR_OPT_EVAL_FORM 0) */
       SEIR_K_STATE = 5, SEIR_K_TILES_VALUE = 6, SEIR_K_TILES_GRAD = 7, SEIR_K_FINISH_FUSED = 8 };
int seir_time_kernel(seir_ctx *ctx, int32_t which, int32_t B, int32_t iters, float *mean_ms);

/* Self-test hook for the device math the kernels share (csrc/device_math.h):
 * for each x[i] > 0 returns L[i] = log(1-exp(-x)), inv[i] = 1/expm1(x) and
 * lfact[i] = log Gamma(floor(x)+1) as the kernels evaluate them.  Host pointers. */
int seir_selftest_math(seir_ctx *ctx, int32_t n, const double *x, double *L, double *inv, double *lfact);
/* the 8-term variant used where rates are not small (the I->R terms of the HMC kernels): L, inv as above */
int seir_selftest_math_wide(seir_ctx *ctx, int32_t n, const double *x, double *L, double *inv);


/* ------------------------------------------------------------------------
 * Device-resident Metropolis-within-Gibbs sampler.
 *
 * Stands in for the kernel stack the reference assembles per window in
 * inference.py:86-101 / :151-167 / :219-228 and mcmc_kernel_factory.py:14-168:
 *   GibbsKernel[ (0, HMC [+DualAveraging [+DiagonalMassMatrixAdaptation]]),
 *                (1, MultiScanKernel(num_event_time_updates,
 *                      GibbsKernel[ MH(EventTimesUpdate S->E), MH(EventTimesUpdate E->I),
 *                                   MH(OccultUpdate S->E),     MH(OccultUpdate E->I) ])) ]
 * and for tfp.mcmc.sample_chain(num_results, ..., trace_fn=trace_results_fn)
 * (inference.py:107-115,232-240,245-282).  One "sweep" = one posterior draw of
 * every chain.  All state stays in HBM between calls; draws and traces are
 * written to a device-side burst buffer and read back with
 * seir_sampler_read_trace (the reference's per-burst posterior.write_samples,
 * inference.py:453-468).
 * ------------------------------------------------------------------------ */
#define SEIR_MMAX 4           /* upper bound on config["m"] */
#define SEIR_MOVE_TRACE (2 + 4 * SEIR_MMAX)   /* is_accepted, target_log_prob, m[], t[], delta_t[], x_star[] */

typedef struct {
    int32_t num_chains;             /* B <= ctx max_chains */
    int32_t dmax, nmax, m;          /* config["dmax"], ["nmax"], ["m"]   (mcmc_kernel_factory.py:79-81) */
    int32_t occult_nmax;            /* config["occult_nmax"]             (:106) */
    int32_t num_event_time_updates; /* config["num_event_time_updates"]  (:123) */
    int32_t t_range_lo, t_range_hi; /* occult window [lo, hi)            (inference.py:336-339) */
    int32_t num_leapfrog_steps;     /* 16                                (inference.py:326) */
    int32_t trace_capacity;         /* sweeps the burst buffer holds */
    int32_t first_chain_id;         /* global id of chain 0: selects the RNG stream (multi-GPU sharding) */
    int32_t record_events;          /* 0: no samples/seir; 1: int32 counts for every draw; 2: uint16 counts (half the burst
                                       buffer and half the bytes over PCIe; a count > 65535 makes the read fail) */
    uint64_t seed;
    /* ---- ABI v2: launch form and test hooks; all-zero = the defaults ---------------------------- */
    int32_t moves_mode;             /* 0: paired event updates (k_move_pair's steps) with the S->E-type proposal
                                       pre-drawn one pair ahead -- every pair of a sweep and the closing step in ONE
                                       launch (k_move_pairs: the pair launch's grid with band workgroups, resident for
                                       the sweep; between two steps a chain's workgroups meet at a counter and drop
                                       their L1) where all of a chain's workgroups share an XCD
                                       (seir_sampler_xcd_local; any number of chains in the layout of the next multiple
                                       of 8, while every workgroup of the launch fits the chip), else one launch per
                                       pair; 4: always one launch per pair (k_move_pair, band workgroups in the launch
                                       under the same condition); 1: one proposal kernel per update (k_move_pa2) --
                                       kept as a cross-check; 2: as 4 without the pre-draw; 3: as 4 with the band part
                                       of the E->I-type log-ratio always as its own launch (k_move_delta).
                                       Same draws in all five.  The persistent forms (this one and hmc_mode 0) need
                                       every workgroup of their launch resident at once: one sampler at a time per GPU
                                       -- two such launches from different streams or processes can each hold part of
                                       the chip and wait for the rest; their waits are bounded, the sampler then fails
                                       loudly at the next read of the trace (seir_sampler_pair_timeouts), and modes 4 /
                                       hmc_mode 3 are the forms for a shared GPU */
    int32_t hmc_mode;               /* 0: the leapfrog steps by 64-lane chunk roles and the WHOLE trajectory in ONE
                                       persistent launch (k_leap: the gradient tiles keep their cells in registers over the
                                       L+1 gradient evaluations, tiles and chunk roles hand each other the partial sums / the
                                       next tables through the XCD's L2; the roles also draw the momentum, and at the end
                                       make the accept test -- each from the roles' parts, all alike -- restore the start
                                       point on rejection, and do the adaptation and the trace) when all of a chain's
                                       workgroups share an XCD (chain b's block ids are congruent to b mod 8; checked
                                       through XCC_ID at creation) and every workgroup of that launch fits the chip at
                                       once; else one launch per step with the chunk roles inside the gradient launch
                                       (k_se_chunk; needs the XCD placement only) between the stage kernels
                                       (k_hmc_step<0>, <2>); else the chunks as their own launch (k_hmc_chunk).
                                       Sixteen chains run it as two launches of eight, one after the other.
                                       Where the persistent launch does not fit (24+ chains at UK-380, SYN-2048) mode 0 runs
                                       as 6: the whole trajectory as L + 1 per-step launches (k_se_chunk) whose chunk roles also
                                       do the trajectory's first step and last half kick, and k_hmc_final (the accept test,
                                       adaptation and trace by the roles) -- no single-workgroup stage kernel.
                                       5: as 0 with the last half kick, accept test, adaptation and trace by
                                       k_hmc_step<2> as a launch of its own; 4: the persistent launch for the inner steps
                                       only; 3: one launch per step (k_se_chunk), never the persistent one; 2: chunks always
                                       as their own launch (2 and 3: same bits; 0, 4, 5 against them: same draws up to
                                       summation order); 1: every step by the single-workgroup kernel */
    int32_t use_graph;              /* 1: replay the sweep as a captured hipGraph (default: stream launches) */
    int32_t chain_groups;           /* chains split over this many streams (0 or 1: one stream) */
    int32_t disable_mask;           /* bit 0: HMC update, bits 1..4: S->E move, E->I move, S->E occult, E->I
                                       occult.  A disabled sub-kernel still draws its proposal (the random
                                       streams do not shift) but is always rejected -- used by the
                                       invariant-distribution tests to run each MH kernel alone */
    int32_t debug_pair;             /* test hooks of k_move_pair's handshake: 1 late, 2 absent speculative role */
    int32_t leap_rows;              /* tile shape of the persistent leapfrog launch (hmc_mode 0, 4, 5; speed only, same draws up
                                       to the order of summation): 0 = auto -- workgroups of 24 rows x 64 days (six rows per
                                       wave) where ceil64(M) is a multiple of 24 with M <= 512 and T in six 64-day chunks
                                       (UK-380: 96 tile workgroups per chain, three on every CU of the chain's XCD) and the
                                       launch is resident, else 32 rows (two 16-row tiles per workgroup); 24 / 32: that shape
                                       only (the per-step form where it cannot be used) */
    int32_t reserved[1];
} seir_sampler_desc;

int seir_sampler_create(seir_ctx *ctx, const seir_sampler_desc *desc, seir_sampler **out);
void seir_sampler_destroy(seir_sampler *s);

/* current_state = [unconstrained params u[B][P], events[B][M][T][3]] (inference.py:563-576); host pointers */
int seir_sampler_set_state(seir_sampler *s, const double *u, const double *events);
int seir_sampler_get_state(seir_sampler *s, double *u, double *events, double *logp /* [B] running target_log_prob */);

/* HMC step size per chain and diagonal of the momentum precision ("variance",
 * i.e. M = diag(1/variance); NULL = identity = momentum_distribution None)
 * (hmc_kernel_kwargs, inference.py:324-329,384,405-406) */
int seir_sampler_set_kernel(seir_sampler *s, const double *step_size /* [B] */, const double *variance /* [B][P] */);
int seir_sampler_get_kernel(seir_sampler *s, double *step_size, double *variance);

/* Window mode (inference.py:60-121 fast, :125-196 slow, :199-242 fixed):
 * adapt_step_size -> DualAveragingStepSizeAdaptation(target_accept_prob,
 * num_adaptation_steps) restarted at the current step size;
 * adapt_mass -> DiagonalMassMatrixAdaptation seeded with the running
 * variance (count[B], mean[B][P], variance[B][P]) of get_weighted_running_variance
 * (inference.py:36-47).  Pointers may be NULL when adapt_mass == 0. */
int seir_sampler_set_adaptation(seir_sampler *s, int32_t adapt_step_size, int32_t adapt_mass,
                                int32_t num_adaptation_steps, double target_accept_prob,
                                const double *rv_count, const double *rv_mean, const double *rv_variance);

/* Recompute every cache (state planes, F, tables, running log-prob) from the
 * event planes and u; called implicitly by set_state. */
int seir_sampler_refresh(seir_sampler *s);

/* Start a new burst: trace slot 0 = the next sweep. */
int seir_sampler_reset_trace(seir_sampler *s);
/* The same with the next sweep recorded in slot `first_slot`: a burst buffer of 2 n slots used as two
 * halves lets burst k+1 run while burst k leaves the device (seir_sampler_read_trace_async). */
int seir_sampler_reset_trace_at(seir_sampler *s, int32_t first_slot);
/* Enqueue num_sweeps sweeps on the context stream (asynchronous). */
int seir_sampler_run(seir_sampler *s, int32_t num_sweeps);
/* Blocking read of trace slots [first, first+count):
 *   theta  [count][B][P]           constrained draws (param_bijector.inverse, inference.py:375)
 *   events [count][B][M][T][3]     int32 counts -- uint16 if record_events == 2 -- (NULL to skip)
 *   hmc    [count][B][3]           is_accepted, target_log_prob, step_size (inference.py:255-261)
 *   moves  [count][B][4][SEIR_MOVE_TRACE]  per sub-kernel S->E move, E->I move, S->E occult,
 *          E->I occult of the LAST inner scan (MultiScanKernel returns the last results,
 *          inference.py:262-280) */
int seir_sampler_read_trace(seir_sampler *s, int32_t first, int32_t count, double *theta, void *events,
                            double *hmc, double *moves);

/* Overlapped egress of a burst (the reference's per-burst posterior.write_samples, inference.py:453-468,
 * without stalling the sampler): the copies are enqueued on a dedicated copy stream behind everything
 * already queued on the context stream and the call returns at once; sweeps enqueued afterwards run
 * concurrently with the transfer.  Arguments as seir_sampler_read_trace; the host buffers should be
 * page-locked (seir_host_alloc) -- pageable memory works but serialises.  seir_sampler_trace_wait
 * blocks until the last async read has landed; call it before touching the host buffers and before
 * re-using the trace slots being read. */
int seir_sampler_read_trace_async(seir_sampler *s, int32_t first, int32_t count, double *theta, void *events,
                                  double *hmc, double *moves);
int seir_sampler_trace_wait(seir_sampler *s);
/* Page-locked host memory for the calls above. */
int seir_host_alloc(void **host_ptr, uint64_t bytes);
int seir_host_free(void *host_ptr);

/* Mean launch duration (ms, HIP events on the context stream) of the sweep's
 * gradient kernel -- the S->E term + d/d eta sums over all B chains that runs
 * num_leapfrog_steps+1 times per sweep -- replayed `iters` times on the current
 * chain state (it only writes its partial-sum buffers). */
int seir_sampler_time_grad_kernel(seir_sampler *s, int32_t iters, float *mean_ms);

/* 1 if the context's GPU places workgroups whose ids are congruent mod 8 on one XCD each (XCC_ID of a probe grid at
 * creation): the condition under which hmc_mode 0 runs the chunk roles inside the gradient launch. */
int seir_sampler_xcd_local(seir_sampler *s);

/* Time-outs of waits inside a launch, per chain: out [B] = (a) k_move_pair launches in which the
 * authoritative workgroup gave up waiting for a speculative one (it then draws the proposal itself:
 * results are unaffected, throughput is not) + (b) waits that cannot be recovered from (band tokens,
 * k_se_chunk's tile flag, k_leap's flags, k_move_pairs' step barrier).  (a) is benign and only counted
 * here.  (b) makes the next seir_sampler_read_trace / seir_sampler_trace_wait fail with SEIR_ERR_STATE
 * and the error is STICKY: a workgroup that gave up went on with stale data (the incrementally updated
 * F = Cstar . I/N can be out of step with the event planes afterwards), so seir_sampler_run and every read
 * of the trace keep failing until seir_sampler_restore, seir_sampler_set_state or seir_sampler_refresh has
 * rebuilt the state.  Both counts stay 0 outside the debug_pair test hooks and a GPU shared with
 * another persistent launch. */
int seir_sampler_pair_timeouts(seir_sampler *s, uint32_t *out);

/* Measurement hook (no reference counterpart): runs `sweeps` ordinary sweeps with a pair of HIP events around the
 * leapfrog section of each, on the stream the kernels are launched on.  What the section is depends on the launch form
 * in force:
 *   hmc_mode 0 (default, where it fits)  ONE launch, the persistent k_leap: the whole trajectory -- all L+1 gradient
 *                                        evaluations, the L leapfrog steps, accept test, adaptation and trace:
 *                                        launches = 1, evals = L+1;
 *   hmc_mode 5                           the same launch without the trajectory's end: launches = 1, evals = L+1;
 *   hmc_mode 4                           k_leap for the inner steps 1..L-1 only: launches = 1, evals = L-1;
 *   hmc_mode 6 (and 0 where k_leap       the whole trajectory as L+1 k_se_chunk launches and k_hmc_final:
 *     does not fit)                      launches = L+2, evals = L+1;
 *   hmc_mode 3 / 2                       the inner steps 1..L-1 as one k_se_chunk launch each (launches = L-1) or as
 *                                        k_se + k_hmc_chunk (launches = 2(L-1)): evals = L-1.
 * mean_ms = mean duration of that section.  The chains advance as in seir_sampler_run. */
int seir_sampler_time_leapfrog(seir_sampler *s, int32_t sweeps, float *mean_ms, int32_t *launches, int32_t *evals);

/* ------------------------------------------------------------------------
 * Surviving a placement failure of the persistent launches (no reference counterpart: the reference carries
 * `current_state` and the kernel results in host memory from burst to burst, inference.py:457-458, and cannot lose
 * a run to what else is on the GPU).
 *
 * hmc_mode 0 and moves_mode 0 need every workgroup of their launch resident at once; a second sampler, another
 * process or a profiler's replay pass can leave part of a grid unplaced, the bounded waits then time out and the
 * sampler reports SEIR_ERR_STATE (see seir_sampler_pair_timeouts).  The burst loop of inference.py:453-468 is kept
 * alive like this:
 *     seir_sampler_snapshot(s, k & 1)            at the start of burst k (in stream order, a few device copies)
 *     seir_sampler_run(s, n); read the trace
 *     on SEIR_ERR_STATE:  seir_sampler_restore(s, k & 1);  seir_sampler_set_launch_form(s, 3, 4);  run burst k again
 * A snapshot holds everything the next sweep's draws are a function of (event planes, state planes, F, the
 * tables, position, step size, adaptation state, sweep counter) -- not the trace.  After a restore the same
 * launch form reproduces the burst bit for bit, the per-step forms (hmc_mode 3, moves_mode 4) draw for draw with
 * continuous quantities equal up to the order of summation.  Two slots, so that a burst that is still being
 * copied out can be re-run as well.
 * ------------------------------------------------------------------------ */
int seir_sampler_snapshot(seir_sampler *s, int32_t slot /* 0 or 1 */);
int seir_sampler_restore(seir_sampler *s, int32_t slot);
/* Change the launch form (seir_sampler_desc::hmc_mode / moves_mode) of an existing sampler; what is sampled does not
 * change.  seir_sampler_launch_form reads the form in force. */
int seir_sampler_set_launch_form(seir_sampler *s, int32_t hmc_mode, int32_t moves_mode);
int seir_sampler_launch_form(seir_sampler *s, int32_t *hmc_mode, int32_t *moves_mode);
/* Test hook: raises chain `chain`'s fatal time-out counter in stream order, i.e. leaves what a timed-out wait leaves.
 * The sweeps queued behind it run with every long wait of that chain cut short and the next read of the trace fails. */
int seir_sampler_debug_fail_handoff(seir_sampler *s, int32_t chain);

/* ------------------------------------------------------------------------
 * Reproduction number R_it.
 *
 * calc_posterior_rit (covid19uk/posterior/reproduction_number.py:13-44): for each
 * posterior draw and day, the column sums of next_generation_matrix_fn
 * (covid19uk/model_spec.py:302-368).  theta: n constrained draws [n][P] in the order of
 * inference.py:541-552; events: [n][M][T][3] fp64 counts (samples/seir); R_it: [n][T][M].
 * Host pointers, blocking; n is processed in batches of the context's max_chains.
 * ------------------------------------------------------------------------ */
int seir_reproduction_number(seir_ctx *ctx, int32_t n, const double *theta, const double *events, double *R_it);

/* Within- / between-location infection pressure (second half):
 * calc_pressure_components (covid19uk/posterior/within_between.py:13-57).  For each draw,
 * with I the infectives of the last state and x = I/N:
 *   within  = I - psi x W colsum(C),  between = psi W (C + C^T) x,
 * returned as fractions within/(within+between), between/(within+between), each [n][M].
 * psi [n]; I_last [n][M]; W: the commute volume the reference gathers (W[T-1]). Host pointers. */
int seir_within_between(seir_ctx *ctx, int32_t n, const double *psi, const double *I_last, double W,
                        double *within, double *between);

/* ------------------------------------------------------------------------
 * Chain-binomial forward simulation.
 *
 * Replaces `CovidUK(covar_data, initial_state=init_, initial_step, num_steps).sample(**par)["seir"]`
 * mapped over posterior draws in covid19uk/posterior/predict.py:50-70, i.e. gemlib's
 * DiscreteTimeStateTransitionModel.sample with the rates of covid19uk/model_spec.py:232-276:
 * day by day, y_x[m] ~ Binomial(source_x[m], 1 - exp(-rate_x[m] time_delta)).
 *
 * The caller resolves the model's time indexing on the host (model_spec.py:234-256: W, weekday
 * and b_t are gathered at clipped absolute day t = initial_step + s) and passes per-day arrays:
 *   par          [n][5]     psi, sigma_space, beta_area, gamma0, gamma1 (constrained)
 *   log_baseline [n][S]     a_t of every simulated day
 *   spatial      [n][M]     spatial_effect
 *   W, weekday_c [S]        commute volume and centred weekday of every simulated day
 *   init_state   [n][M][4]  S,E,I,R at the first simulated day (integer-valued)
 *   events       [n][M][S][3] out, fp64 counts in the reference's layout
 * Cstar, N, log-area, nu, time_delta and the rate floor come from the context; the context's T
 * does not limit S.  Random stream: Philox4x32-10 keyed by (seed; attempt, 64 + transition,
 * s*M + m, first_draw_id + draw) -- independent of batching and of the number of GPUs.
 * Host pointers, blocking.
 * ------------------------------------------------------------------------ */
typedef struct seir_sim_desc {
    int32_t num_draws;
    int32_t num_steps;
    int32_t first_draw_id;
    int32_t reserved;
    uint64_t seed;
    const double *par;
    const double *log_baseline;
    const double *spatial;
    const double *W;
    const double *weekday_c;
    const double *init_state;
    double *events;
} seir_sim_desc;

int seir_simulate(seir_ctx *ctx, const seir_sim_desc *sim);

/* One Binomial(n, p) variate per element with the simulator's sampler and stream
 * (cell = element index, transition 0, draw id 0): self-test hook for the distribution tests. */
int seir_selftest_binomial(seir_ctx *ctx, int32_t count, const int32_t *n, const double *p, uint64_t seed,
                           int32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* SEIR_HIP_H */
