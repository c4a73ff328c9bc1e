/*
 * ngp.h — C-ABI of the MI355X-native GP inference core ("libngp").
 *
 * This is the drop-in boundary for the ONE hot path NowcastAutoGP delegates to
 * AutoGP.jl: per-particle covariance assembly from a kernel tree -> Cholesky ->
 * log-marginal-likelihood -> posterior-predictive solves, batched over SMC
 * particles x nowcast scenarios.  Reference call sites (paths relative to the
 * reference checkout):
 *     src/make_and_fit_model.jl:84-91   GPModel(...), linear_schedule, fit_smc!
 *     src/forecasting.jl:46-47,65-67  predict_mvn + rand
 *     src/forecasting.jl:133-149          GPModel(dict), add_data!, maybe_resample!,
 *                                         mcmc_structure!, mcmc_parameters!
 * The reference has no FFI of its own (it is pure Julia on AutoGP.jl); the
 * entry points below are what a Julia `ccall` / Python `ctypes` shim binds in
 * place of AutoGP's internal covariance/Cholesky/logpdf arithmetic.  See
 * INTEGRATION.md for the binding stubs.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns an
 *     ngp_status (0 = ok, <0 = bad argument, >0 = HIP runtime error code).
 *   - all pointers are caller-owned HOST memory, valid for the duration of the
 *     call (or, for staged jobs, until ngp_job_run returns for inputs and
 *     ngp_job_fetch returns for outputs); the library retains none of them.
 *   - Float64 everywhere (src/forecasting.jl:62); dense outputs are row-major
 *     and symmetric where that applies, so Julia's column-major view is the same
 *     matrix.
 *   - numerical failure is reported PER ITEM in info[] with LAPACK potrf
 *     semantics (k > 0: leading minor k is not positive definite), which the
 *     shim rethrows as PosDefException(k) (src/make_and_fit_model.jl:6-8).
 *   - re-entrant: may be entered concurrently from many host threads
 *     (Threads.@spawn per scenario, src/forecasting.jl:131-132); device work of one
 *     ctx is serialised by a blocking mutex, never a spin, and concurrent one-shot
 *     calls on the same dates are combined into one launch sequence ("concurrent
 *     callers" below).
 */
#ifndef NGP_H
#define NGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kernel grammar: opcode numbering follows AutoGP.GP.GPConfig ---------
 * (docs/src/vignettes/setting-priors.md:229-236 in the reference).            */
enum {
    NGP_OP_CONSTANT     = 1, /* params: value                                 */
    NGP_OP_LINEAR       = 2, /* params: intercept, bias, amplitude            */
    NGP_OP_SQEXP        = 3, /* params: lengthscale, amplitude                */
    NGP_OP_GAMMAEXP     = 4, /* params: lengthscale, gamma, amplitude         */
    NGP_OP_PERIODIC     = 5, /* params: lengthscale, period, amplitude        */
    NGP_OP_PLUS         = 6, /* binary, no params                             */
    NGP_OP_TIMES        = 7, /* binary, no params                             */
    NGP_OP_CHANGEPOINT  = 8  /* binary, params: location, scale               */
};

#define NGP_MAX_OPS    64  /* nodes per kernel tree                            */
#define NGP_MAX_PARAMS 96  /* continuous parameters per kernel tree            */
#define NGP_MAX_STACK  16  /* RPN evaluation stack depth                       */
#define NGP_MAX_AUX    192 /* appended + forecast rows per item (d_tail+d+m+1) */
/* Size limits beyond these (NGP_ERR_TOO_LARGE): a series of more than 16,256 points (11,520 for
 * ngp_logml_grad_batch) — an item's factor storage is addressed with 32-bit byte offsets; a
 * resident factor of more than 65,535 particles.  Batches of any size are cut into chunks that
 * fit the device memory; series longer than 8,319 points run NGP_PREC_MIXED jobs in fp64.        */

/* Formula variants.  AutoGP.jl's source is not available in the build
 * container, so the per-node formulas are restated from memory (SURVEY.md
 * Appendix B); each doubtful choice is a data-driven flag so that correcting
 * it is a one-line spec change + fixture regeneration, not a kernel rewrite. */
typedef struct ngp_spec {
    int32_t se_form;       /* 0: a*exp(-0.5*d^2/l^2)       1: a*exp(-0.5*d^2/l)        */
    int32_t periodic_form; /* 0: a*exp(-(2/l^2)*sin^2(pi*d/p))  1: a*exp(-(2/l)*sin^2(pi*d/p)) */
    int32_t cp_form;       /* 0: sigma(x)=.5*(1+tanh((loc-x)/scale))  1: tanh((x-loc)/scale) */
    int32_t precision;     /* NGP_PREC_F64 (default) or NGP_PREC_MIXED, see below          */
    double  jitter;        /* added to the diagonal next to the noise variance */
    /* ---- NGP_PREC_MIXED only (BASELINE config C5: long histories) ---------------------
     * The trailing updates of the blocked Cholesky run on the fp32 matrix cores wherever
     * that is provably harmless and in fp64 elsewhere: the product of two 64x64 tiles of L
     * goes through v_mfma_f32_32x32x2_f32 (operands rounded to fp32) iff
     *     64 * 2^-24 * max|tile A| * max|tile B|  <=  mixed_tau * (noise + jitter),
     * i.e. iff its rounding error is below mixed_tau of the smallest pivot the matrix can
     * have; everything else (the diagonal blocks, the panel solves, the accumulators and
     * the stored factor) stays fp64.  The Gram matrix X K^-1 X' of the appended / forecast
     * / data rows is then refined against the fp64 covariance (G <- A X' + (X - A K) A',
     * A += (X - A K) (L L')^-1) until the predicted remaining relative error is below
     * refine_tol or refine_max steps were taken; an item that does not get there is
     * reported with info = NGP_INFO_NOT_REFINED.  log det comes from the factor itself.
     * Series of fewer than 128 or more than 8,319 points, gradient jobs and resident
     * factors run in fp64 whatever this field says (ngp_job_mixed_stats: frac_f32 = 0).  */
    double  mixed_tau;     /* default 1e-6                                              */
    double  refine_tol;    /* default 1e-9                                              */
    int32_t refine_max;    /* default 3 (0: no refinement)                              */
    int32_t reserved;
} ngp_spec;
enum { NGP_PREC_F64 = 0, NGP_PREC_MIXED = 1 };
/* info[] < 0: not a pivot index */
#define NGP_INFO_NOT_REFINED (-2)

/* One particle's covariance kernel: the tree in postfix (RPN) order
 * (left subtree, right subtree, operator); params are consumed in RPN order. */
typedef struct ngp_kernel {
    int32_t        n_ops;
    int32_t        n_params;
    const int32_t *ops;     /* [n_ops] opcodes 1..8                            */
    const double  *params;  /* [n_params]                                      */
    double         noise;   /* observation-noise variance on the diagonal      */
} ngp_kernel;

typedef int32_t ngp_status;
enum {
    NGP_OK               =  0,
    NGP_ERR_ARG          = -1, /* null pointer / negative size                 */
    NGP_ERR_PROGRAM      = -2, /* malformed kernel program                     */
    NGP_ERR_TOO_LARGE    = -3, /* exceeds NGP_MAX_* or device memory           */
    NGP_ERR_NO_DEVICE    = -4, /* no HIP device / extension not usable         */
    NGP_ERR_STATE        = -5, /* job used out of order                        */
    NGP_ERR_UNAVAILABLE  = -6  /* optional component missing (librccl.so)      */
};

typedef struct ngp_ctx ngp_ctx;
typedef struct ngp_job ngp_job;

/* ---- context ------------------------------------------------------------- */
ngp_status  ngp_ctx_create(int32_t device, ngp_ctx **out);
void        ngp_ctx_destroy(ngp_ctx *ctx);
ngp_status  ngp_set_spec(ngp_ctx *ctx, const ngp_spec *spec);
ngp_status  ngp_get_spec(const ngp_ctx *ctx, ngp_spec *spec);
void        ngp_default_spec(ngp_spec *spec);
const char *ngp_strerror(ngp_status st);
const char *ngp_version(void);
/* validates a kernel program (arity, stack depth, parameter count) */
ngp_status  ngp_kernel_check(const ngp_kernel *k);

/* ---- concurrent callers ----------------------------------------------------
 * The reference enters this boundary from one task per nowcast scenario
 * (Threads.@spawn, src/forecasting.jl:131-159): D tasks with the P particles of their own clone
 * each, all on the same dates.  One-shot calls that arrive while the device is busy —
 * ngp_logml_batch, ngp_predict_batch, ngp_logml_grad_batch, ngp_mixture_sample (S = 1) — are
 * therefore COMBINED: the calling thread that finds nobody serving takes every pending request,
 * and requests of the same entry point on bytewise identical dates (same n, same forecast dates
 * and flags) run as ONE launch sequence of sum(B) items with per-item observation rows; every
 * caller gets its own results and status.  A call that arrives alone runs exactly as before;
 * only a caller whose predecessors came in company gives that company at most 200 microseconds to
 * arrive before it serves (tasks woken together by one sequence come back within microseconds of
 * each other), and a wait that was in vain is not repeated.  src/forecasting.jl needs no edit to
 * reach batches of P x (concurrent tasks) items.  A result is the one a single call over the combined
 * items would give: equal to the caller's own call up to the last bits (batch size decides
 * launch shapes and therefore summation order, as it always did — see DESIGN.md section 4.14).
 * ngp_set_combining(ctx, 0) switches it off (every call then waits for ctx's lock and runs alone);
 * 1 (default) is as described; 2 combines what is pending but never waits for company (without the
 * wait a convoy of T tasks alternates between groups of 1 and T - 1: measured 1.45 x slower at
 * 8 x 24 items, n = 208 (380 launch sequences instead of 200), 1.08 x at 16 x 64 items, n = 2048 —
 * profiles/r04/combine_linger_ab.txt).
 * A group of logml or predictive requests that carry THE SAME kernels (byte for byte) and whose
 * observations differ only in their last few points — the scenario tasks of the reference's default
 * mode, n_mcmc = n_hmc = 0 (src/forecasting.jl:120, 133-155): clones of one model, each with its own
 * nowcast values on shared dates (src/create_nowcast_data.jl:36-37) — is served by ONE factorisation
 * per particle with the tasks' last points as scenarios (the arithmetic of ngp_nowcast_batch), not by
 * one factorisation per (particle, task): K does not depend on y.  Results equal the caller's own
 * call to rounding (1e-8 on predictive moments in the tests, condition-aware).
 * ngp_combine_stats: out6 = { requests seen, launch sequences run for them, largest group,
 * requests that shared a sequence with at least one other, requests served from one shared
 * factorisation per particle, reserved }; reset != 0 clears the counters.                      */
ngp_status ngp_set_combining(ngp_ctx *ctx, int32_t on);
/* Batch-invariant arithmetic (off by default; applies to jobs staged after the call).  By default
 * a few decisions follow the size of the batch an item travels in, for speed: late block columns of
 * a small chunk are split along k, a small mixed gradient batch runs all its items on the general
 * leaf while a large one gives its stationary trees to the Toeplitz leaf, the contraction of a small
 * batch cuts its tiles finer and picks one kernel shape for all its trees, a single-chunk job's
 * epilogue reads resident tables.  Each of these changes a summation order or an arithmetic path:
 * results agree to rounding (1e-13 on logml, 1e-11 on gradients in the tests) but not bit for bit, so
 * what a caller gets depends on who shared its launch sequence — with combining, on thread timing.
 * With this option on, every such decision is taken from the item and the series alone: an item's
 * outputs are the same bits whether it is evaluated alone, in a lockstep P x D call or in a
 * combined group (tests/test_combine_gpu.py, tests/test_lockstep_gpu.py), at the price of the
 * small-batch shortcuts (64 items at n = 2048: see DESIGN.md section 4.14).                    */
ngp_status ngp_set_batch_invariant(ngp_ctx *ctx, int32_t on);
ngp_status ngp_combine_stats(ngp_ctx *ctx, int64_t *out6, int32_t reset);

/* ---- covariance assembly (diagnostic / small blocks) ---------------------
 * out[b] (n1 x n2, row-major) = k_b(t1_i, t2_j) (+ (noise_b + jitter) on the
 * diagonal i==j when add_diag != 0).  Replaces AutoGP's covariance-matrix
 * builder used under fit_smc!/predict_mvn (src/make_and_fit_model.jl:91,
 * src/forecasting.jl:46).                                                   */
ngp_status ngp_cov_batch(ngp_ctx *ctx, int32_t B, const ngp_kernel *kernels,
                         int32_t n1, const double *t1, int32_t n2, const double *t2,
                         int32_t add_diag, double *out);

/* ---- log marginal likelihood --------------------------------------------
 * logml[b] = log N(y_b | 0, K_b(t,t) + (noise_b + jitter) I), b = 0..B-1.
 * y is [B x n] with row stride ldy (ldy == 0: one y shared by all items).
 * This is the per-particle evaluation inside fit_smc! / add_data! /
 * mcmc_structure! (src/make_and_fit_model.jl:91, src/forecasting.jl:135,146). */
ngp_status ngp_logml_batch(ngp_ctx *ctx, int32_t B, const ngp_kernel *kernels,
                           int32_t n, const double *t, const double *y, int64_t ldy,
                           double *logml, int32_t *info);

/* ---- posterior predictive ------------------------------------------------
 * Per item: mu[b] (m), sigma[b] (m x m) of  f(t_new) | y_b  (+ observation
 * noise on the new points when noise_on_new != 0), and logml[b] (may be NULL).
 * Replaces the per-particle conditional MVN inside AutoGP.predict_mvn
 * (src/forecasting.jl:46,66).                                              */
ngp_status ngp_predict_batch(ngp_ctx *ctx, int32_t B, const ngp_kernel *kernels,
                             int32_t n, const double *t, const double *y, int64_t ldy,
                             int32_t m, const double *t_new, int32_t noise_on_new,
                             double *mu, double *sigma, double *logml, int32_t *info);

/* ---- nowcast fan-out ------------------------------------------------------
 * The whole body of forecast_with_nowcasts' per-scenario task for the default
 * n_mcmc = n_hmc = 0 path (src/forecasting.jl:133-155), for P particles and D
 * scenarios at once.  All scenarios share the appended dates
 * (src/create_nowcast_data.jl:36-37), and K does not depend on y, so each
 * particle is factorised ONCE; per scenario only the d appended observations
 * differ.
 *   in : P kernels; base data (t[n], y[n]); appended times t_add[d];
 *        y_add [D x d] row-major; forecast times t_new[m].
 *   out: logml_base[P]      log p(y | particle)                 (n points)
 *        logml_full[P x D]  log p(y, y_add_s | particle)        (n+d points)
 *                           -> add_data! incremental weight = full - base
 *        mu   [P x D x m]   predictive mean given (y, y_add_s)
 *        sigma[P x m x m]   predictive covariance (scenario-independent)
 *        info [P]
 * Any output pointer may be NULL.                                            */
ngp_status ngp_nowcast_batch(ngp_ctx *ctx, int32_t P, const ngp_kernel *kernels,
                             int32_t n, const double *t, const double *y,
                             int32_t d, const double *t_add,
                             int32_t D, const double *y_add,
                             int32_t m, const double *t_new, int32_t noise_on_new,
                             double *logml_base, double *logml_full,
                             double *mu, double *sigma, int32_t *info);

/* ---- gradient of the log marginal likelihood ------------------------------
 * grad[b] has n_params_b + 1 entries: d logml / d params (RPN order) followed
 * by d logml / d noise; items are packed back to back (offsets = running sum of
 * n_params_b + 1).  Needed by the HMC moves of mcmc_parameters! / fit_smc!
 * (src/forecasting.jl:65,148; src/make_and_fit_model.jl:91).               */
ngp_status ngp_logml_grad_batch(ngp_ctx *ctx, int32_t B, const ngp_kernel *kernels,
                                int32_t n, const double *t, const double *y, int64_t ldy,
                                double *logml, double *grad, int32_t *info);

/* The same evaluation with its inputs resident on the device: trees, dates and observations are
 * staged once (ngp_grad_stage: exactly ngp_logml_grad_batch's arguments), every
 * ngp_grad_job_run evaluates logml and gradient for the parameters the job currently holds, and
 * ngp_grad_job_set_params replaces them between runs — `params`: the items' parameter vectors
 * back to back in the caller's (RPN) order, n_params_b each; noise[B].  The leapfrog steps of
 * one HMC move (src/forecasting.jl:65,148; src/make_and_fit_model.jl:91) change nothing but the
 * parameters: per step only they cross the bus (1 KiB per item instead of the observations and
 * the compiled trees).  ngp_logml_grad_batch = stage + run + destroy; a run's outputs are those
 * of the one-shot call, bit for bit.  The job keeps the spec it was staged under; it holds only
 * its inputs and results between runs (the factor storage is taken per run).                  */
typedef struct ngp_grad_job ngp_grad_job;
ngp_status ngp_grad_stage(ngp_ctx *ctx, int32_t B, const ngp_kernel *kernels, int32_t n,
                          const double *t, const double *y, int64_t ldy, ngp_grad_job **out);
ngp_status ngp_grad_job_set_params(ngp_grad_job *job, const double *params, const double *noise);
ngp_status ngp_grad_job_run(ngp_grad_job *job, double *logml, double *grad, int32_t *info);
/* How the job is carried (diagnostic; tests use it to pick the first and last item of every chunk):
 * out5 = { items of the general leaf, items per memory-driven chunk of its last run (0 before the
 * first), items of the Toeplitz leaf (stationary trees on a regular series, ngp_set_structured_storage),
 * items per chunk of its last run, 1 if the two leaves run side by side }.  Leaf items keep the
 * caller's order.                                                                                */
ngp_status ngp_grad_job_info(const ngp_grad_job *job, int32_t *out5);
void       ngp_grad_job_destroy(ngp_grad_job *job);

/* ---- particle weights -----------------------------------------------------
 * maybe_resample! arithmetic (src/forecasting.jl:138-141): normalise P
 * log-weights (logsumexp), effective sample size 1 / sum w^2.  In a multi-GPU
 * run the caller all-gathers the per-rank log-weights first (the only
 * collective on the path) and passes the gathered vector.
 * w_norm (P, may be NULL), ess, log_norm (log sum exp logw) out.              */
ngp_status ngp_weights_normalize(int32_t P, const double *logw,
                                 double *w_norm, double *ess, double *log_norm);
/* The same for D weight vectors at once — the D scenario clones forecast_with_nowcasts
 * advances together (src/forecasting.jl:131-141): logw and w_norm are [P x D] row-major (one
 * COLUMN per scenario, the layout of the all-gathered [P_local, D] shards), ess and log_norm
 * are [D].  Column s equals ngp_weights_normalize on that column.                          */
ngp_status ngp_weights_normalize_cols(int32_t P, int32_t D, const double *logw,
                                      double *w_norm, double *ess, double *log_norm);

/* ---- the collective of the path, for hosts without a collective library --------------
 * maybe_resample! is the only step of the hot path that needs every rank's particles
 * (src/forecasting.jl:138-141): the P_total log-weights of every scenario.  A Julia host that
 * runs one process per GPU has no torch.distributed; these entry points give it that one exchange
 * over RCCL (xGMI inside a node).  librccl.so is opened at run time (no link-time dependency);
 * without it every call returns NGP_ERR_UNAVAILABLE and the host must gather the weights itself
 * and call ngp_weights_normalize_cols.
 *   ngp_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other
 *                        ranks by the host's own means (a file, a socket, MPI, Julia Distributed)
 *   ngp_comm_create      collective over all ranks (ncclCommInitRank) on the context's device
 *   ngp_weights_allgather_normalize
 *        particles are block-partitioned over the ranks, remainder to the low ranks; this rank
 *        passes its rows logw_local [P_local x D] (D scenario columns, row-major); ONE all-gather
 *        of padded shards, then the normalisation of ngp_weights_normalize_cols on every rank:
 *        w_local [P_local x D] (may be NULL), w_all [P_total x D] (may be NULL: what resampling
 *        needs), ess [D], log_norm [D] (may be NULL).  Identical on every rank.
 *        Every rank must own at least one particle: P_total >= world (NGP_ERR_ARG otherwise).
 *        A non-zero return on ANY rank is fatal for the communicator — the others may be inside
 *        the all-gather: destroy it and make a new one.  ngp_comm_create gives this process's
 *        cached device blocks back before its ONE ncclCommInitRank (a collective: never retried
 *        by one rank alone).
 *   ngp_weights_unpad_normalize
 *        the host half of that exchange, for a host that brings its own all-gather: `padded` is
 *        what gathering equally sized zero-padded shards delivers — world blocks of pmax x D doubles,
 *        pmax = rows of rank 0 (ngp_shard), block r = rank r's rows, then padding — and comes out
 *        as w_all [P_total x D] (may be NULL), ess [D] (may be NULL), log_norm [D] (may be NULL).  */
typedef struct ngp_comm ngp_comm;
/* the block partition itself (host-only arithmetic): rank's first global particle and its count */
ngp_status ngp_shard(int32_t P_total, int32_t world, int32_t rank, int32_t *first, int32_t *rows);
ngp_status ngp_comm_unique_id(void *id128);
ngp_status ngp_comm_create(ngp_ctx *ctx, const void *id128, int32_t rank, int32_t world,
                           ngp_comm **out);
void       ngp_comm_destroy(ngp_comm *comm);
ngp_status ngp_weights_allgather_normalize(ngp_comm *comm, int32_t P_total, int32_t D,
                                           const double *logw_local, double *w_local,
                                           double *w_all, double *ess, double *log_norm);
ngp_status ngp_weights_unpad_normalize(int32_t P_total, int32_t world, int32_t D,
                                       const double *padded, double *w_all, double *ess,
                                       double *log_norm);

/* ---- staged execution (inputs resident in HBM before the timed region) ----
 * stage  : validate, allocate device buffers, copy inputs host -> device
 * run    : enqueue every kernel of the job and wait for completion
 * fetch  : copy outputs device -> host (same output arguments as the one-shot
 *          entry point that created the job; NULL pointers are skipped)
 * The one-shot entry points above are stage + run + fetch + destroy.          */
ngp_status ngp_logml_stage(ngp_ctx *ctx, int32_t B, const ngp_kernel *kernels,
                           int32_t n, const double *t, const double *y, int64_t ldy,
                           ngp_job **job);
ngp_status ngp_predict_stage(ngp_ctx *ctx, int32_t B, const ngp_kernel *kernels,
                             int32_t n, const double *t, const double *y, int64_t ldy,
                             int32_t m, const double *t_new, int32_t noise_on_new,
                             ngp_job **job);
ngp_status ngp_nowcast_stage(ngp_ctx *ctx, int32_t P, const ngp_kernel *kernels,
                             int32_t n, const double *t, const double *y,
                             int32_t d, const double *t_add,
                             int32_t D, const double *y_add,
                             int32_t m, const double *t_new, int32_t noise_on_new,
                             ngp_job **job);
ngp_status ngp_job_run(ngp_job *job);
ngp_status ngp_job_fetch(ngp_job *job, double *logml_base, double *logml_full,
                         double *mu, double *sigma, int32_t *info);
/* NGP_PREC_MIXED jobs, after ngp_job_run (any pointer may be NULL; all [B]):
 *   refine_steps  refinement steps taken for the item (0 for an fp64 job)
 *   refine_delta  relative size of the last correction applied to its Gram matrix
 *   frac_f32      share of its tile products that ran on the fp32 matrix cores         */
ngp_status ngp_job_mixed_stats(ngp_job *job, int32_t *refine_steps, double *refine_delta,
                               double *frac_f32);
void       ngp_job_destroy(ngp_job *job);

/* ---- mixture sampling on the device -------------
 * predict_mvn(...) |> rand of the reference (src/forecasting.jl:47, 67) for S
 * mixtures over the same P components at once (S = nowcast scenarios): component
 * k ~ Categorical(w[s][.]), then  mu[k][s][.] + chol(sigma[k]) z,  z ~ N(0, I).
 *   w     [S x P]      mixture weights, each row sums to 1 (ngp_weights_normalize)
 *   mu    [P x S x m]  component means   (layout of ngp_nowcast_batch's mu)
 *   sigma [P x m x m]  component covariances, shared by the S mixtures
 *   out   [S x draws x m]; comp [S x draws] (may be NULL) the component drawn;
 *   info  [P] (may be NULL): 0, or the first non-positive pivot of chol(sigma[k])
 * Randomness: Philox4x32-10 (counter = (draw, scenario, block, 0), key = seed):
 * reproducible for a given seed on any device and by the numpy restatement in
 * oracle/, NOT against Julia's Xoshiro stream — the reference's own tests only
 * check shapes and ranges of draws.  m <= NGP_MAX_AUX.     */
ngp_status ngp_mixture_sample(ngp_ctx *ctx, int32_t P, int32_t S, int32_t m,
                              const double *w, const double *mu, const double *sigma,
                              int32_t draws, uint64_t seed,
                              double *out, int32_t *comp, int32_t *info);
/* S INDEPENDENT mixtures of P components each — the scenario clones after a refinement
 * (mcmc_structure! / mcmc_parameters!, src/forecasting.jl:145-148) no longer share their
 * particles:
 *   w [S x P], mu [S x P x m], sigma [S x P x m x m], seeds [S],
 *   out [S x draws x m], comp [S x draws] (may be NULL), info [S x P] (may be NULL).
 * Mixture s is keyed by seeds[s] with scenario counter 0: it draws exactly what
 * ngp_mixture_sample(P, S = 1, ..., seed = seeds[s]) draws, so one call replaces S calls.   */
ngp_status ngp_mixture_sample_indep(ngp_ctx *ctx, int32_t P, int32_t S, int32_t m,
                                    const double *w, const double *mu, const double *sigma,
                                    int32_t draws, const uint64_t *seeds,
                                    double *out, int32_t *comp, int32_t *info);

/* ---- cached factor ------------------------------
 * A fitted model is queried many times with the same particles and the same
 * training data: forecast() on several date grids, forecast_with_nowcasts()
 * after it (src/forecasting.jl:46, 135).  ngp_factor_create factorises the P
 * training covariances ONCE and keeps L (and the block inverses the solves
 * use) resident on the device; a query then only fills its aux rows (appended
 * points, forecast points, y) and sweeps them through L: O(n^2 (d + m)) per
 * particle instead of n^3 / 3.  Queries return exactly what
 * ngp_nowcast_batch returns for the same arguments (d = 0: plain predict).
 * The handle owns device memory of about 8 n (n + 192) bytes per particle;
 * it copies the kernels and the data it was created with.
 *   create : y is [P][n] with row stride ldy (0: one y shared by all)
 *   logml  : log p(y | particle) and the factorisation status of create
 *   nowcast: d + m <= NGP_MAX_AUX - (n mod 64) - 1                            */
typedef struct ngp_factor ngp_factor;
ngp_status ngp_factor_create(ngp_ctx *ctx, int32_t P, const ngp_kernel *kernels,
                             int32_t n, const double *t, const double *y, int64_t ldy,
                             ngp_factor **out);
ngp_status ngp_factor_logml(const ngp_factor *f, double *logml, int32_t *info);
ngp_status ngp_factor_nowcast(ngp_factor *f, int32_t d, const double *t_add,
                              int32_t D, const double *y_add,
                              int32_t m, const double *t_new, int32_t noise_on_new,
                              double *logml_base, double *logml_full,
                              double *mu, double *sigma, int32_t *info);
void       ngp_factor_destroy(ngp_factor *f);

/* ---- measurement hooks -----------------------------------------------------
 * HIP-event timing of the kernels a job launches, on the stream they are
 * launched on.  Classes: 0 = chol_col_glds_kernel (fat steps: trailing-update
 * GEMM + fused solve, the dominant kernel), 1 = chol_diag, 2 = gram,
 * 3 = epilogue, 4 = cov fill, 5 = K^-1 = W W' of a gradient job (grad_kinv*), 6 = chol_col_kernel (thin /
 * full steps, aux solves of a resident factor), 7 = aux_update_kernel,
 * 8 = diag_ahead_kernel (side stream, overlaps classes 1 and 6), 9 = the
 * mixed-precision fat steps (chol_col_glds_kernel<MIXED>), 10 = Gram refinement
 * of NGP_PREC_MIXED (backward sweep, covariance apply, small products), 11 = the
 * reverse-mode contraction of a gradient job (grad_alpha / grad_contract* / grad_reduce),
 * 12 = the fat steps of a gradient job's general leaf (chol_col_glds_kernel<.., IDENT>: aux
 * rows [I ; y'], the sweep that also produces W = L^-T) — a different instantiation from class 0,
 * with its own flops, bytes and rate (the Toeplitz leaf of a gradient job runs class 0),
 * 13 = chol_small_kernel: the whole factorisation of a short series (n0 <= 256) in one launch,
 * in place of classes 0, 1, 6, 8 and 12 (ngp_set_short_series_path). */
#define NGP_NUM_KERNEL_CLASSES 14
typedef struct ngp_profile {
    double   ms[NGP_NUM_KERNEL_CLASSES];       /* summed device time per class */
    int64_t  launches[NGP_NUM_KERNEL_CLASSES]; /* kernel launches per class    */
    double   flops[NGP_NUM_KERNEL_CLASSES];    /* algorithmic flops executed   */
    double   bytes[NGP_NUM_KERNEL_CLASSES];    /* algorithmic HBM bytes        */
} ngp_profile;
/* Storage option of staged fp64 value jobs (on by default).  On a regular series (the main
 * block's dates at a constant lattice stride) the covariance matrix of a stationary kernel tree is
 * Toeplitz, K_ik = f(|i - k|): 127 numbers describe a 64 x 64 tile exactly.  Such an item's tiles
 * below the block diagonal are then never written to HBM — the column sweep regenerates a tile
 * from those numbers in LDS where it would have read the stored tile.  The values are the table
 * entries the fill would have stored, so every result is bit-identical with the option off
 * (tests/test_structured_storage_gpu.py); off exists for that comparison and for A/B timing.
 * The same switch governs the Toeplitz gradient path of gradient jobs (ngp_logml_grad_batch,
 * ngp_grad_stage): on a regular series the items whose trees are stationary are evaluated from the
 * Cholesky factor alone — the diagonal sums of K^-1 that the gradient needs follow from its first
 * column (Gohberg-Semencul), n^3/3 flops instead of n^3.  That is other arithmetic, not other
 * storage: logml and gradients agree with the general path to rounding (1e-11 on gradients in the
 * tests), not bit for bit.  An item qualifies if its tree has no Linear / ChangePoint node and at
 * most 16 leaves, the series is regular (128 <= n <= 8192) and noise + jitter >= 1e-9 k(0) (the
 * formula loses about eps cond(K) digits; with the default jitter the guard never binds); which
 * qualifying items of a MIXED batch actually take it depends on the batch size unless
 * ngp_set_batch_invariant is on.  Applies to jobs staged after the call. */
ngp_status ngp_set_structured_storage(ngp_ctx *ctx, int32_t on);
/* Short series in one launch (on by default).  The column sweep is a chain of dependent launches
 * per 64-wide block column; at the reference's everyday size (a few hundred points, 24-64
 * particles: docs/vignettes/getting-started.jl:266-268) that chain is the whole call.  With this
 * option a job whose main block is at most 256 points (value jobs: n < 320; gradient jobs:
 * n <= 256) is factorised by ONE kernel, one workgroup per item, the matrix in registers as
 * 16 x 16 blocks; such jobs store every tile (no structured storage) and
 * their gradient items all take the general leaf.  The rule is the geometry's alone, so it does
 * not make an item's bits depend on its batch.  Results agree with the column sweep to rounding
 * (another summation order); off exists for that comparison and for A/B timing.  Resident factors
 * (ngp_factor_*) and mixed-precision jobs stay on the column sweep.  Applies to jobs staged after
 * the call. */
ngp_status ngp_set_short_series_path(ngp_ctx *ctx, int32_t on);

ngp_status ngp_profile_enable(ngp_ctx *ctx, int32_t on);
ngp_status ngp_profile_reset(ngp_ctx *ctx);
ngp_status ngp_profile_get(ngp_ctx *ctx, ngp_profile *out);

/* fp64 MFMA issue-rate microbenchmark (v_mfma_f64_16x16x4_f64); returns the
 * measured dense TFLOP/s over `iters` back-to-back MFMAs per wave.            */
ngp_status ngp_microbench_mfma_f64(ngp_ctx *ctx, int32_t iters, double *tflops);
/* Same loop with in-kernel stamps: out[0] TFLOP/s (wall), out[1] median shader cycles per MFMA
 * per wave, out[2] median shader clock held under load in GHz, out[3] waves per SIMD.      */
ngp_status ngp_microbench_mfma_f64_detail(ngp_ctx *ctx, int32_t iters, int32_t blocks_per_cu,
                                          double *out);
/* MFMA operand-map self test, all row-major: D[0:256] = A[16x4] B[4x16] through one
 * v_mfma_f64_16x16x4_f64; D[256:512] = the same product through four DPP-rotated
 * v_mfma_f64_4x4x4_4b_f64 gathered back to the 16x16x4 C/D layout (the k-loop fast path).
 * The caller compares both halves with A @ B.  D must hold 512 doubles.                 */
ngp_status ngp_selftest_mfma_layout(ngp_ctx *ctx, const double *A, const double *B, double *D);
/* The same for the fp32 form of the mixed-precision path: D[32x32] = A[32x2] B[2x32] through one
 * v_mfma_f32_32x32x2_f32 (all row-major floats).                                          */
ngp_status ngp_selftest_mfma_f32_layout(ngp_ctx *ctx, const float *A, const float *B, float *D);
/* HBM streaming-write microbenchmark (GB/s) used to anchor the fill roofline. */
ngp_status ngp_microbench_hbm(ngp_ctx *ctx, int64_t bytes, double *write_gbs, double *copy_gbs);

#ifdef __cplusplus
}
#endif
#endif /* NGP_H */
