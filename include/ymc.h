/*
 * ymc.h -- C ABI of the MI355X chain-ensemble library (libymc.so).
 *
 * The reference (rkutri/yagre-mcmc) has no FFI: its hot path is the Python
 * loop MetropolisHastings.run (yagremcmc/chain/metropolisHastings.py:103-120)
 * over plug-in objects.  This header is the boundary a binding for that path
 * targets: plain pointers and sizes, opaque handles, int status codes, no
 * exceptions, no callbacks into the caller.  Each entry point names the
 * reference interface it stands in for.  The shipped binding is ctypes
 * (yagre-mcmc_amd/_ffi.py); INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Ownership: the caller owns every host buffer; the library owns all device
 * memory.  Handles are not thread-safe; one host thread drives one context.
 * All functions return YMC_OK (0) or a negative ymc_status;
 * ymc_last_error() gives the message of the calling thread's last failure.
 */
#ifndef YMC_H
#define YMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YMC_ABI_VERSION 4

typedef enum ymc_status {
    YMC_OK = 0,
    YMC_ERR_INVALID = -1,      /* bad argument / inconsistent description (ValueError in the reference's builders, builder.py:44-50) */
    YMC_ERR_DEVICE = -2,       /* HIP runtime error */
    YMC_ERR_UNSUPPORTED = -3,  /* a plug-in combination that has no device lowering */
    YMC_ERR_CAPACITY = -4,     /* record buffers full */
    YMC_ERR_COMM = -5          /* RCCL error */
} ymc_status;

/* ---- densities: lowered forms of DensityInterface.evaluate_log ---------
 * (yagremcmc/statistics/interface.py:6-10) */
typedef enum ymc_density_kind {
    /* GaussianTargetDensity2d (test/testSetup.py:32-40), any dim <= 8:
     * -0.5 * (log_norm + |(x - mean) whiten|^2) */
    YMC_DENSITY_GAUSS_WHITENED = 1,
    /* GaussianTargetDensity1d (test/testSetup.py:15-25): -0.5 (mean-x)^2 / variance */
    YMC_DENSITY_GAUSS_ISO1D = 2,
    /* UnnormalisedPosterior (chain/target.py:19-22) of an
     * AdditiveGaussianNoiseLikelihood (statistics/likelihood.py:49-87) over
     * ExampleLinearModelSolver (exampleSetup.py:42-47) and a Gaussian prior
     * (statistics/gaussian.py:19-24); diagonal covariances */
    YMC_DENSITY_POSTERIOR_LINEAR = 3,
    /* the same over LotkaVolterraSolver (test/testSetup.py:61-141) with the
     * ODE integrated by fixed-step RK4 or, with lv_rtol > 0, adaptively */
    YMC_DENSITY_POSTERIOR_LV = 4,
    /* EXTENSION kinds (csrc/ymc_kinds.def; INTEGRATION.md "Adding a forward model"): the same posterior over another
     * SolverInterface (model/interface.py:7-67) solved once per design row.  The worked example: logistic growth
     * dx/dt = r x (1 - x / K), theta = (log r, log K), design = [n_data] initial populations, t_final, rk4_steps;
     * data_dim = 1 (the population at t_final) */
    YMC_DENSITY_POSTERIOR_LOGISTIC = 5
} ymc_density_kind;

typedef struct ymc_density_desc {
    int32_t kind;                  /* ymc_density_kind */
    int32_t dim;                   /* parameter dimension d */
    const double *mean;            /* GAUSS_*: [d] */
    const double *whiten;          /* GAUSS_WHITENED: [d][d] row-major */
    double log_norm;               /* GAUSS_WHITENED: rank*log(2 pi) + log pdet */
    double variance;               /* GAUSS_ISO1D */
    int32_t n_data;                /* POSTERIOR_*: rows of Data.array (statistics/data.py:4-23) */
    int32_t data_dim;              /* POSTERIOR_*: columns */
    const double *data;            /* [n_data][data_dim] */
    const double *noise_precision; /* [data_dim]   (covariance.py:36-37) */
    const double *prior_mean;      /* [d] */
    const double *prior_precision; /* [d] */
    const double *fwd_matrix;      /* POSTERIOR_LINEAR: A [data_dim][d] */
    const double *fwd_shift;       /* POSTERIOR_LINEAR: b [data_dim] */
    const double *design;          /* POSTERIOR_LV: [n_data][2] initial conditions; extension kinds: [n_data][their design columns] */
    double t_final;                /* POSTERIOR_LV, POSTERIOR_LOGISTIC */
    double alpha;
    double gamma;
    int32_t rk4_steps;
    int32_t likelihood_only;       /* POSTERIOR_*: 1 = the likelihood alone (the target of pCN, pcn.py:77-78) */
    /* DenseCovarianceMatrix (covariance.py:69-94) as the NOISE covariance (noise.py:21-22) or the PRIOR covariance
     * (gaussian.py:19-24): the INVERSE of its lower Cholesky factor, [data_dim][data_dim] / [d][d] row-major
     * (entries above the diagonal are ignored), so that |r|^2 in the covariance's norm is |L^-1 r|^2.  NULL = the
     * diagonal covariance of noise_precision / prior_precision.  At lane-kernel sizes (d, data_dim <= 8) the
     * kernels form the norm per evaluation; beyond (POSTERIOR_LINEAR on the matrix cores: any d for YMC_METHOD_MRW,
     * YMC_METHOD_PCN, YMC_METHOD_MLDA2 and YMC_METHOD_MLDA_NESTED -- a sequence of plain kernels per transition
     * beyond 256 --; d <= 256 and any number of
     * outputs for YMC_METHOD_MRW, YMC_METHOD_PCN, YMC_METHOD_MLDA2 and YMC_METHOD_MLDA_NESTED, whose
     * forward matrices stream from L2 when they do not fit the LDS; d <= 128 and an LDS-resident coarse matrix for AEM2) a dense NOISE
     * covariance is whitened away when the problem is created -- A, b and the data rows are multiplied by L^-1
     * once, |L^-1 (A v + b - D_i)|^2 = |(L^-1 A) v + L^-1 b - L^-1 D_i|^2 -- and a dense PRIOR covariance becomes
     * d more outputs of the forward map (rows Lp^-1, shift 0, "data" Lp^-1 mu, weight 1: -0.5 |Lp^-1 (v - mu)|^2
     * is their log-likelihood);
     * ymc_forward_eval still returns
     * A v + b.  Not with YMC_METHOD_PCN (its proposal draws from a diagonal prior); YMC_METHOD_AEM2 takes
     * diagonal noise (the reference's AEMNoise refuses any other, noise.py:30-33) and, at that size, a diagonal
     * prior. */
    const double *noise_chol_inv;
    const double *prior_chol_inv;
    /* POSTERIOR_LV: lv_rtol > 0 integrates with the adaptive Dormand-Prince 5(4) pair (ymc_lv_dp45_solve,
     * csrc/ymc_numerics.h; the reference hands its `rtol` to scipy's adaptive solve_ivp, testSetup.py:124-126) and
     * rk4_steps is ignored: each component's local error within lv_atol + lv_rtol |y|.  0 = rk4_steps steps of
     * classical RK4 (which overflows where the dynamics outrun its step, e.g. at the starting point of
     * example_inference_lotkaVolterra_singleLevel.py:106).  A solve that fails counts as a non-finite forward
     * value: the proposal is rejected and the chain flagged (YMC_CHAIN_NONFINITE). */
    double lv_rtol;
    double lv_atol;
} ymc_density_desc;

typedef enum ymc_method {
    YMC_METHOD_MRW = 1,   /* MetropolisedRandomWalk, chain/method/mrw.py:41-57 */
    YMC_METHOD_MLDA2 = 2, /* MLDA with one surrogate, chain/method/mlda.py:100-110,146-154 */
    YMC_METHOD_PCN = 3,   /* PreconditionedCrankNicolson, chain/method/pcn.py:9-57 */
    /* MLDA with an EVEN number n of surrogates exactly as the reference executes it
     * (mlda.py:12-73,100-116; n = 2: example_mcmc_2d_hierarchical.py, test_mlda.py:47-60;
     * n = 4: test_mlda.py:189-256).  MLDAProposal.generate_proposal calls generate_proposal of the
     * top SurrogateTransition, which calls run() of the level below, which calls generate_proposal
     * of the one below that, ...: only every second level's accept step is ever executed.  What
     * runs is a nest of E = n/2 + 1 effective levels:
     *   level 0      a random walk on surrogateTargets[0], subChainLengths[1] steps
     *   level e      1 <= e <= E-2: subChainLengths[2e+1] steps, each = one level-(e-1) run accepted
     *                with exp(P(y) + Q(x) - Q(y) - P(x)), P = surrogateTargets[2e], Q = [2e-1]
     *   level E-1    the recorded chain: P = the target, Q = surrogateTargets[n-1]
     * (An odd n >= 3 raises AttributeError in the reference; the Python layer raises the same.)
     * Lane kernels up to d = 8; hierarchies of POSTERIOR_LINEAR models of any d on the matrix cores. */
    YMC_METHOD_MLDA_NESTED = 4,
    /* AdaptiveErrorModel with one surrogate (chain/method/aem.py:6-58): MLDA2 whose coarse
     * AEMLikelihood (statistics/likelihood.py:90-155, noise.py:25-61) learns the model error
     * F_fine - F_coarse from the accepted fine proposals, per chain, including the reference's
     * three-entry AEMCache of log-likelihood values (utility/memoisation.py:76-149).
     * POSTERIOR_LINEAR only (the reference's error model needs a vector-valued forward map). */
    YMC_METHOD_AEM2 = 5
} ymc_method;

#define YMC_MAX_LEVELS 8 /* effective levels E of YMC_METHOD_MLDA_NESTED (n <= 14 surrogates) */

typedef enum ymc_state_eq {
    YMC_EQ_EXACT = 0,   /* ParameterVector.__eq__, parameter/vector.py:37-45 */
    YMC_EQ_ISCLOSE = 1  /* ScalarParameter.__eq__,  parameter/scalar.py:39-44 */
} ymc_state_eq;

/* What MRWBuilder / MLDABuilder.build_method() assemble (mrw.py:60-91,
 * mlda.py:157-344), lowered to numbers. */
typedef struct ymc_problem_desc {
    int32_t method;                /* ymc_method */
    int32_t state_eq;              /* ymc_state_eq */
    ymc_density_desc target;       /* the (fine) target density */
    ymc_density_desc surrogate;    /* MLDA2: the coarse density (same kind and dim) */
    int32_t sub_chain_length;      /* MLDA2, AEM2: subChainLengths[0] */
    int32_t aem_min_data_size;     /* AEM2: AEMLikelihood minDataSize (>= 2, likelihood.py:99-100) */
    /* [d]: sqrt(reciprocal(precision)) of the proposal covariance (covariance.py:51-52);
     * PCN: of the PRIOR covariance (the proposal draws from the prior, pcn.py:29) */
    const double *proposal_sigma;
    double pcn_step_size;          /* PCN: stepSize h, 0 < h <= 0.5 (pcn.py:42) */
    /* MLDA_NESTED (all densities of the same kind and dim as `target`, which is level E-1's P): */
    int32_t n_levels;                         /* E, 2 .. YMC_MAX_LEVELS */
    int32_t aem_use_noise_heuristic;          /* AEM2: AEMLikelihood useNoiseHeuristic (noise.py:41-46) */
    const ymc_density_desc *level_target;     /* [E-1]: P of levels 0 .. E-2 */
    const ymc_density_desc *level_correction; /* [E-1]: Q of levels 1 .. E-1 */
    const int32_t *level_steps;               /* [E-1]: steps of one run of levels 0 .. E-2 */
    /* MRW with a DenseCovarianceMatrix proposal (covariance.py:69-94): its lower Cholesky factor
     * [d][d], row-major; NULL = the diagonal covariance of proposal_sigma.  Row i of L xi is summed left to
     * right with separate multiplies and adds by the lane kernels (d <= 8 and at most 8 forward outputs) and as
     * one k-ordered fma chain on the matrix cores beyond (linear posteriors of any d). */
    const double *proposal_chol;
} ymc_problem_desc;

/* record_flags */
#define YMC_RECORD_TRAJECTORY 1u  /* Chain.append, chain/chain.py:17-18 */
#define YMC_RECORD_ACCEPT 2u      /* AcceptanceRateDiagnostics.process, chain/diagnostics.py:48-54 */
#define YMC_RECORD_LOGPOST 4u     /* log-density of every recorded state (parity checks) */
#define YMC_RECORD_SUB_ACCEPT 8u  /* MLDA2: outcomes of the coarse sub-steps */
#define YMC_RECORD_MOMENTS 16u    /* FullDiagnostics' Welford accumulator, chain/diagnostics.py:91-94 */

/* per-chain status bits */
#define YMC_CHAIN_NONFINITE 1u    /* a proposal's log-density was not finite; it was rejected */
#define YMC_CHAIN_HANDOFF 2u      /* internal error: a launch cut into hand-over pieces (Lotka-Volterra, small ensembles)
                                     never received this chain's state from the preceding piece and the chain stopped there,
                                     or a group of workgroups sharing this chain's tile (linear models beyond 256
                                     parameters, small ensembles) gave up waiting for one of its members */

/* MetropolisHastings.run(chainLength, initialState) for n_chains chains
 * (metropolisHastings.py:103-110): chain k of this run is GLOBAL chain
 * chain_offset + k of the ensemble and draws from that id's stream. */
typedef struct ymc_run_desc {
    uint64_t seed;
    uint64_t chain_offset;
    uint64_t n_chains;
    const double *initial_state;   /* [x0_rows][d] */
    uint64_t x0_rows;              /* 1 (broadcast) or n_chains */
    uint64_t record_capacity;      /* transitions the record buffers hold */
    uint32_t record_flags;
    /* thinned record (SURVEY 8b record_mode; the examples thin by the IAT right after run,
     * example_mcmc_2d_singleLevel.py:71-79): 0 or 1 = every transition is recorded; s > 1 = the
     * trajectory / log-density rows hold the states after transitions s, 2s, 3s, ... of the run
     * (rows s-1::s of the full record), so record_capacity counts THINNED rows.  Accept counts and
     * Welford moments still see every transition; the per-transition decision records
     * (YMC_RECORD_ACCEPT, YMC_RECORD_SUB_ACCEPT) are full-rate data and cannot be combined with s > 1. */
    uint32_t record_stride;
} ymc_run_desc;

typedef struct ymc_device_info {
    char name[128];
    char arch[64];
    int32_t compute_units;
    int32_t clock_mhz;
    uint64_t hbm_bytes;
} ymc_device_info;

typedef struct ymc_ctx ymc_ctx;
typedef struct ymc_problem ymc_problem;
typedef struct ymc_run ymc_run;
typedef struct ymc_comm ymc_comm;

int ymc_abi_version(void);
const char *ymc_last_error(void);
int ymc_device_count(int *count);

/* Lifetime: handles may be destroyed in ANY order (a garbage-collecting host language gives none).  A
 * context that still has problems, runs or communicators, or a problem that still has runs, is only
 * marked released by its destroy call and is freed by its last dependent. */
int ymc_ctx_create(int device, ymc_ctx **out);
int ymc_ctx_destroy(ymc_ctx *ctx);
int ymc_ctx_device_info(const ymc_ctx *ctx, ymc_device_info *out);
/* The shader clock the device HOLDS while it works (the data-sheet clock_mhz is a ceiling: under a
 * sustained fp64 load the MI355X is power-limited and runs below it).  _start launches a one-wave
 * probe on a side stream that sleeps for duration_ms (<= 20 000) next to whatever the context's
 * stream is running and counts shader cycles (s_memtime) against the 100 MHz reference counter
 * (s_memrealtime); _read waits for it and returns cycles / time. */
int ymc_ctx_clock_probe_start(ymc_ctx *ctx, double duration_ms);
int ymc_ctx_clock_probe_read(ymc_ctx *ctx, double *mhz, double *measured_ms);

/* ChainBuilder.build_method(), chain/builder.py:72-83 */
int ymc_problem_create(ymc_ctx *ctx, const ymc_problem_desc *desc, ymc_problem **out);
int ymc_problem_destroy(ymc_problem *problem);

/* MetropolisHastings.run, entry part (:103-110).  The record holds the state
 * AFTER each transition; the initial states (row 0 of the reference's
 * trajectory) stay with the caller. */
int ymc_run_create(ymc_ctx *ctx, const ymc_problem *problem, const ymc_run_desc *desc, ymc_run **out);
int ymc_run_destroy(ymc_run *run);

/* The hot loop, metropolisHastings.py:112-120: `nsteps` more transitions of
 * every chain (MLDA2: fine steps).  Asynchronous on the run's stream. */
int ymc_run_advance(ymc_run *run, uint64_t nsteps);
int ymc_run_sync(ymc_run *run);
/* average device time of the advance kernel launches since the last call, by
 * HIP events on the run's stream; resets the accumulator */
int ymc_run_kernel_time_ms(ymc_run *run, double *avg_ms, uint64_t *launches);
/* forget recorded rows (chain state is kept), so the buffers can be refilled */
int ymc_run_rewind_record(ymc_run *run);
int ymc_run_steps_done(const ymc_run *run, uint64_t *total, uint64_t *recorded);

/* Chain.trajectory (chain/chain.py:9-11): rows [first, first+n) of the record,
 * dst[n][n_chains][d] */
int ymc_run_fetch_trajectory(ymc_run *run, uint64_t first, uint64_t n, double *dst, size_t dst_bytes);
/* AcceptanceRateDiagnostics._decisions (chain/diagnostics.py:22): dst[n][n_chains] */
int ymc_run_fetch_accept(ymc_run *run, uint64_t first, uint64_t n, uint8_t *dst, size_t dst_bytes);
int ymc_run_fetch_logpost(ymc_run *run, uint64_t first, uint64_t n, double *dst, size_t dst_bytes);
/* dst[n][sub_chain_length][n_chains] */
int ymc_run_fetch_sub_accept(ymc_run *run, uint64_t first, uint64_t n, uint8_t *dst, size_t dst_bytes);
/* MLDA_NESTED: accepted transitions of the lower levels so far, dst[E-1][n_chains], and the
 * transitions each of them has made, made[E-1] (either may be NULL) */
int ymc_run_fetch_level_accept(ymc_run *run, uint64_t *dst, size_t dst_bytes, uint64_t *made);
/* AEM2: the coarse likelihood's learned error model per chain (likelihood.py:107-109,142-150):
 * n[n_chains] accepted fine proposals, mean / m2 [n_chains][data_dim] Welford state (marginal
 * variance = m2 / (n - 1)), precision[n_chains][data_dim] of the inflated noise (valid where
 * has_precision[n_chains] != 0); any pointer may be NULL */
int ymc_run_fetch_error_model(ymc_run *run, uint64_t *n, double *mean, double *m2, double *precision,
                              int32_t *has_precision);
/* current state [n_chains][d] and its log-density [n_chains] (either may be NULL) */
int ymc_run_fetch_state(ymc_run *run, double *state, double *logpost);
/* per-chain summary: accepted transitions, Welford n / mean / M2
 * (statistics/estimation.py:4-58), status bits; any pointer may be NULL */
int ymc_run_fetch_summary(ymc_run *run, uint64_t *accepted, uint64_t *wf_n, double *wf_mean,
                          double *wf_m2, uint32_t *status);

/* ---- egress: the record on its way to the host -----------------------------------------------------
 * ymc_run_fetch_* synchronise, then copy into pageable caller memory (~10 GB/s) while the GPU idles.  The
 * egress calls stage rows through two page-locked buffers owned by the run, on a copy stream of their own:
 * _begin returns at once; the copy starts when the launches issued so far have finished and runs BESIDE
 * the launches issued afterwards; _wait blocks until the slot's copy is done and hands out pointers into
 * the page-locked buffer (valid until the next _begin on that slot or ymc_run_destroy).  A later
 * ymc_run_advance that would overwrite record rows still being copied waits for that copy ON THE DEVICE,
 * so the natural loop is: record capacity 2 n; advance n (rows [0, n)), begin(0, 0, n); advance n (rows
 * [n, 2n)), begin(1, n, n); wait(0); rewind; advance n; begin(0, 0, n); wait(1); ...
 * `what`: any of YMC_RECORD_TRAJECTORY | _ACCEPT | _LOGPOST the run records.  Rows are record rows
 * (thinned rows under record_stride).  The trajectory arrives as [n][n_chains][d], like ymc_run_fetch_trajectory. */
typedef struct ymc_egress_view {
    const double *trajectory;  /* [rows][n_chains][d] or NULL */
    const uint8_t *accept;     /* [rows][n_chains] or NULL */
    const double *logpost;     /* [rows][n_chains] or NULL */
    uint64_t first, rows;
} ymc_egress_view;
#define YMC_EGRESS_SLOTS 2
int ymc_run_egress_begin(ymc_run *run, int slot, uint64_t first, uint64_t n, uint32_t what);
int ymc_run_egress_wait(ymc_run *run, int slot, ymc_egress_view *out);

/* the packed per-chain summary records of this run (the payload of the multi-GPU gather below,
 * written by the same kernel): dst[n_chains][ymc_summary_words] */
int ymc_run_fetch_packed_summary(ymc_run *run, uint64_t *dst, size_t dst_bytes);

/* DensityInterface.evaluate_log for a batch: which = 0 target, 1 surrogate.
 * x[n][d] -> out[n], status[n] (may be NULL) */
int ymc_logdensity_eval(ymc_ctx *ctx, const ymc_problem *problem, int which, uint64_t n,
                        const double *x, double *out, uint32_t *status);
/* ForwardModel.evaluate (model/forwardModel.py:10-19) for a batch:
 * x[n][d] -> out[n][m], m = data_dim (linear) or 2*n_data (Lotka-Volterra) */
int ymc_forward_eval(ymc_ctx *ctx, const ymc_problem *problem, int which, uint64_t n,
                     const double *x, double *out);

/* ---- post-processing (yagremcmc/postprocessing/autocorrelation.py) --------
 * integrated_autocorrelation (:92-140) for every chain of a run, from record
 * rows [first, first + n) (the caller's burn-in is `first`): method 0 = 'mean'
 * (IAT of the per-state mean over the components), 1 = 'max' (largest IAT over
 * the components); sokal_const as :92.  iat[n_chains]; YMC_IAT_UNDEFINED for a chain whose series is
 * constant (zero variance: the autocorrelation is 0 / 0, the reference fails on it). */
#define YMC_IAT_UNDEFINED (-2147483647 - 1) /* INT32_MIN: a short anti-correlated series can have a genuine IAT of -1 */
int ymc_run_iat(ymc_run *run, uint64_t first, uint64_t n, int method, double sokal_const, int32_t *iat);
/* the same for states held by the caller: states[n][n_chains][d] (the shape of
 * ymc_run_fetch_trajectory's output; n_chains = 1 is the reference's (n, d) array) */
int ymc_iat_states(ymc_ctx *ctx, const double *states, uint64_t n, uint64_t n_chains, int32_t d, int method,
                   double sokal_const, int32_t *iat);
/* and for plain sequences: series[n_series][length] (row-major), one IAT per series;
 * acf (optional) receives the first acf_lags values of
 * estimate_autocorrelation_function_1d (:5-29) per series, acf[n_series][acf_lags] */
int ymc_iat_series(ymc_ctx *ctx, const double *series, uint64_t n_series, uint64_t length, double sokal_const,
                   int32_t *iat, double *acf, uint64_t acf_lags);

/* parity hooks: the device's draws and elementary functions, bit-for-bit
 * comparable with the CPU build of csrc/ymc_numerics.h.
 * xi[nt][K][d], u[nt][K]; fn: 0 log, 1 exp, 2 sincospi (y0 = sin, y1 = cos),
 * 3 sqrt, 4 x / y1_in (y1 holds the divisor on entry), 6 the Box-Muller radius sqrt(max(x, 2^-100)) */
int ymc_rng_fill(ymc_ctx *ctx, uint64_t seed, uint64_t chain0, uint64_t K, uint64_t t0, uint64_t nt,
                 int32_t d, double *xi, double *u);
int ymc_numerics_eval(ymc_ctx *ctx, int fn, uint64_t n, const double *x, double *y0, double *y1);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI --------------------
 * The ensemble shards by contiguous blocks of global chain ids; nothing is
 * exchanged while sampling.  The single collective is the gather of the
 * per-chain summaries at the end. */
#define YMC_UNIQUE_ID_BYTES 128
/* loads librccl (every rank can find out that it is missing BEFORE any rank blocks in the bring-up) */
int ymc_comm_preload(void);
int ymc_comm_unique_id(void *id, size_t id_bytes);
int ymc_comm_create(ymc_ctx *ctx, const void *id, size_t id_bytes, int nranks, int rank, ymc_comm **out);
int ymc_comm_destroy(ymc_comm *comm);
/* ncclCommCount / ncclCommUserRank: the size and rank RCCL itself reports (either may be NULL) */
int ymc_comm_count(const ymc_comm *comm, int *nranks, int *rank);
int ymc_comm_barrier(ymc_comm *comm);
int ymc_comm_allreduce_max(ymc_comm *comm, double *value);
/* words per chain of a packed summary record: 3*d + 5 (state, logpost,
 * accepted, wf_n, wf_mean, wf_m2, status), all 8-byte words */
int ymc_summary_words(const ymc_run *run, uint64_t *words);
/* ncclAllGather of the packed summaries of `run` (every rank passes the same
 * n_chains); dst[nranks][n_chains][words] on every rank */
int ymc_comm_allgather_summary(ymc_comm *comm, ymc_run *run, uint64_t *dst, size_t dst_bytes);

#ifdef __cplusplus
}
#endif
#endif /* YMC_H */
