/*
 * ymc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, one-chain-at-a-time restatement of the reference's
 * Metropolis-Hastings hot path (rkutri/yagre-mcmc @ 2025-01-03), used only as
 * the checker for the HIP kernels: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; nothing under yagre-mcmc_amd/
 * does.  Every function cites the reference file:line it follows.
 *
 * Parity status: PINNED.  tests/golden/make_golden.py imports the reference
 * in the build container, replays this file's random draws into the
 * unmodified reference classes and stores the reference's trajectories,
 * accept decisions and log-densities as .npz files under tests/golden/;
 * tests/test_oracle_golden.py checks this restatement against them (masks and
 * trajectories bit-exact, log-densities <= 1e-12 relative).
 *
 * Arithmetic: IEEE binary64, compiled with -ffp-contract=off; random draws
 * and elementary functions come from yagre-mcmc_amd/csrc/ymc_numerics.h (the
 * numeric specification shared with the device code, itself pinned against
 * libm/mpmath and the Random123 known-answer vectors in
 * tests/test_numerics.py), so the GPU path can be compared bit-for-bit.
 *
 * The one deliberate deviation from the reference is documented at
 * ymo_accept(): a non-finite proposal log-density is rejected and flagged
 * instead of being accepted through the reference's NaN hole (mrw.py:57).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../yagre-mcmc_amd/csrc/ymc_numerics.h"
#include "../yagre-mcmc_amd/csrc/kinds/logistic_numerics.h" /* the worked extension kind's arithmetic */

#ifdef _OPENMP
#include <omp.h>
#endif

#define YMO_DMAX 2048  /* largest parameter dimension the oracle handles */
#define YMO_MMAX 8192 /* largest forward-map output dimension (stack arrays: 64 KB each) */

enum {
    YMO_GAUSS_WHITENED = 1, /* testSetup.py:32-40  GaussianTargetDensity2d (any d) */
    YMO_GAUSS_ISO1D = 2,    /* testSetup.py:15-25  GaussianTargetDensity1d */
    YMO_POST_LINEAR = 3,    /* target.py:19-22 over exampleSetup.py:42-47 */
    YMO_POST_LV = 4,        /* target.py:19-22 over testSetup.py:96-141 (RK4 in place of solve_ivp) */
    YMO_POST_LOGISTIC = 5   /* the worked EXTENSION kind (csrc/kinds/logistic.h): the same posterior over a logistic-growth
                             * ODE solver, a second SolverInterface (model/interface.py:7-67) */
};

enum { YMO_EQ_EXACT = 0, YMO_EQ_ISCLOSE = 1 };

enum { YMO_STATUS_NONFINITE = 1 };

typedef struct ymo_density {
    int32_t kind;
    int32_t dim;
    /* GAUSS_WHITENED: logpi = -0.5 * (log_norm + |(x-mean) U|^2), U = whiten (dim x dim, row-major) */
    const double *mean;
    const double *whiten;
    double log_norm;
    /* GAUSS_ISO1D */
    double variance;
    /* POST_*: Gaussian-noise likelihood + Gaussian prior, diagonal covariances */
    int32_t n_data;
    int32_t data_dim;
    const double *data;            /* n_data x data_dim, row-major (data.py:4-23) */
    const double *noise_precision; /* data_dim (covariance.py:36-37) */
    const double *prior_mean;      /* dim */
    const double *prior_precision; /* dim */
    /* POST_LINEAR: G(theta) = A theta + b */
    const double *fwd_matrix;      /* data_dim x dim, row-major */
    const double *fwd_shift;       /* data_dim */
    /* POST_LV */
    const double *design;          /* n_data x 2 initial conditions */
    double t_final, alpha, gamma;
    int32_t rk4_steps;
    int32_t likelihood_only;       /* 1: evaluate the likelihood alone (the target of pCN, pcn.py:77-78) */
    /* POST_LINEAR: ymc_linear_data_summary of (data, noise_precision), filled by ymo_linear_summary */
    const double *lin_dbar, *lin_nprec;
    double lin_const;
    /* dense noise / prior covariance (DenseCovarianceMatrix, covariance.py:69-94): the inverse of its lower
     * Cholesky factor, or NULL for the diagonal covariance above (ymc_whitened_norm_sq, ymc_numerics.h) */
    const double *noise_linv, *prior_linv;
    /* POST_LV: lv_rtol > 0 = the adaptive integrator (ymc_lv_dp45_solve, ymc_numerics.h) instead of RK4 */
    double lv_rtol, lv_atol;
} ymo_density;

/* ---------------------------------------------------------------------- */
/* numeric-spec hooks (so tests can pin ymc_numerics.h against libm etc.)   */
/* ---------------------------------------------------------------------- */

void ymo_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out)
{
    ymc_u32x4 b = ymc_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    memcpy(out, b.v, sizeof b.v);
}

void ymo_log_v(int64_t n, const double *x, double *y)
{
    for (int64_t i = 0; i < n; ++i) y[i] = ymc_log(x[i]);
}

void ymo_accept_prob_v(int64_t n, const double *x, double *y)
{
    for (int64_t i = 0; i < n; ++i) y[i] = ymc_accept_prob(x[i]);
}

void ymo_exp_v(int64_t n, const double *x, double *y)
{
    for (int64_t i = 0; i < n; ++i) y[i] = ymc_exp(x[i]);
}

/* a[i] / b[i] through ymc_div_by_recip with y = 1/b[i] by the hardware division */
void ymo_div_by_recip_v(int64_t n, const double *a, const double *b, double *q)
{
    for (int64_t i = 0; i < n; ++i) q[i] = ymc_div_by_recip(a[i], b[i], 1.0 / b[i]);
}

void ymo_sincospi_v(int64_t n, const double *x, double *s, double *c)
{
    for (int64_t i = 0; i < n; ++i) ymc_sincospi(x[i], &s[i], &c[i]);
}

/*
 * The draws chain `chain0 + k` consumes in transitions t0 .. t0+nt-1:
 *   xi[t][k][0..d-1]  replaces standard_normal(d)      (gaussian.py:63)
 *   u[t][k]           replaces uniform(0,1,size=1)[0]  (metropolisHastings.py:68)
 */
void ymo_rng_fill(uint64_t seed, uint64_t chain0, int64_t K, uint64_t t0, int64_t nt, int32_t d,
                  double *xi, double *u)
{
    for (int64_t t = 0; t < nt; ++t) {
        for (int64_t k = 0; k < K; ++k) {
            uint32_t chain = (uint32_t)(chain0 + (uint64_t)k);
            if (u) u[t * K + k] = ymc_draw_uniform(seed, chain, t0 + (uint64_t)t);
            if (xi) {
                double *row = xi + (t * K + k) * (int64_t)d;
                for (int32_t j = 0; 2 * j < d; ++j) {
                    double z0, z1;
                    ymc_draw_normal_pair(seed, chain, t0 + (uint64_t)t, (uint32_t)j, &z0, &z1);
                    row[2 * j] = z0;
                    if (2 * j + 1 < d) row[2 * j + 1] = z1;
                }
            }
        }
    }
}

/* ---------------------------------------------------------------------- */
/* densities                                                              */
/* ---------------------------------------------------------------------- */

/* CovarianceMatrix.induced_norm_squared with a diagonal precision:
 * covariance.py:19-22 (np.dot(x, Px)) and :54-55 (Px = precision * x).
 * np.dot's summation order is BLAS' business; the ensemble fixes it as FOUR
 * INTERLEAVED partial sums (index j feeds partial j mod 4, each in increasing
 * j) combined as (S0 + S1) + (S2 + S3) -- the order in which the d = 128 tile
 * kernel finds the terms in the fp64 MFMA accumulator layout. */
static double ymo_diag_norm_sq(const double *x, const double *prec, int32_t n)
{
    double S[4] = {0.0, 0.0, 0.0, 0.0};
    for (int32_t j = 0; j < n; ++j) {
        double px = prec[j] * x[j];
        S[j & 3] = S[j & 3] + x[j] * px;
    }
    return (S[0] + S[1]) + (S[2] + S[3]);
}

/* GaussianDensity.evaluate_log, gaussian.py:19-24: -0.5 |x - mean|^2_{Sigma^-1},
 * no normalising constant. */
static double ymo_gaussian_prior_log(const ymo_density *p, const double *theta)
{
    double x[YMO_DMAX];
    for (int32_t j = 0; j < p->dim; ++j) x[j] = theta[j] - p->prior_mean[j];
    if (p->prior_linv) return -0.5 * ymc_whitened_norm_sq(p->prior_linv, x, p->dim);
    return -0.5 * ymo_diag_norm_sq(x, p->prior_precision, p->dim);
}

/* ExampleLinearModelSolver.invoke, exampleSetup.py:42-47: A @ theta + b.
 * The dot product is a k-ordered fma chain (the canonical order of the
 * ensemble; BLAS' own order is unspecified). */
static void ymo_forward_linear(const ymo_density *p, const double *theta, double *F)
{
    for (int32_t m = 0; m < p->data_dim; ++m) {
        const double *row = p->fwd_matrix + (int64_t)m * p->dim;
        double acc = 0.0;
        for (int32_t j = 0; j < p->dim; ++j) acc = ymc_fma(row[j], theta[j], acc);
        F[m] = acc + p->fwd_shift[m];
    }
}

/* LotkaVolterraParameter.evaluate (testSetup.py:56-58: exp of the coefficient)
 * + LotkaVolterraSolver.interpolate/invoke (testSetup.py:101-141): one ODE
 * solve per design row, final state only (:139).  The reference integrates
 * with scipy's adaptive solve_ivp; the ensemble fixes classical RK4 with
 * rk4_steps equal steps (BASELINE.json north_star), so this is the build's
 * integrator, checked against the reference solver to ~1e-3 in
 * tests/test_oracle_golden.py.  Returns 0 on success, 1 if any output is
 * non-finite (EvaluationStatus.FAILURE, evaluation.py:4-9). */
static int ymo_forward_lv(const ymo_density *p, const double *theta, double *F)
{
    double beta = ymc_exp(theta[0]);
    double delta = ymc_exp(theta[1]);
    double alpha = p->alpha, gamma = p->gamma;
    double h = p->t_final / (double)p->rk4_steps;
    double h2 = 0.5 * h;
    double h6 = h / 6.0;
    int bad = 0;
    for (int32_t n = 0; n < p->n_data; ++n) {
        double x = p->design[2 * n], y = p->design[2 * n + 1];
        if (p->lv_rtol > 0.0) {
            if (!ymc_lv_dp45_solve(&x, &y, alpha, beta, gamma, delta, p->t_final, p->lv_rtol, p->lv_atol,
                                   YMC_LV_MAX_ATTEMPTS))
                x = y = NAN;
        } else {
            for (int32_t s = 0; s < p->rk4_steps; ++s) {
                ymc_lv_rk4_step(&x, &y, alpha, beta, gamma, delta, h, h2, h6); /* ymc_numerics.h */
            }
        }
        F[2 * n] = x;
        F[2 * n + 1] = y;
        if (!(fabs(x) <= 1.79769313486231570815e308) || !(fabs(y) <= 1.79769313486231570815e308))
            bad = 1;
    }
    return bad;
}

/* the extension kind's forward map (csrc/kinds/logistic_numerics.h, shared with the device): theta = (log r, log K),
 * one solve per design row x0_n, F_n = x(t_final).  Returns 1 if any output is non-finite. */
static int ymo_forward_logistic(const ymo_density *p, const double *theta, double *F)
{
    const double r = ymc_exp(theta[0]), k = ymc_exp(theta[1]);
    int bad = 0;
    for (int32_t n = 0; n < p->n_data; ++n) {
        F[n] = ymc_logistic_solve(p->design[n], r, k, p->t_final, p->rk4_steps);
        if (!(fabs(F[n]) <= 1.79769313486231570815e308)) bad = 1;
    }
    return bad;
}

/* AdditiveGaussianNoiseLikelihood.compute_log_likelihood, likelihood.py:77-84:
 *   residual = fmEval - data.array          (:74-75, broadcast over rows)
 *   per row   noise.induced_norm_squared    (:33-39, noise.py:21-22)
 *   logL = -0.5 * sum(rows)                 (:80)
 * F is (data_dim,) for the linear map (one evaluation, broadcast against every
 * data row) and (n_data, data_dim) for Lotka-Volterra (row i pairs with D_i). */
static double ymo_log_likelihood(const ymo_density *p, const double *F, int per_row_eval)
{
    if (!per_row_eval && !p->noise_linv) {
        /* one forward value for all rows: |F - Dbar|^2_{nP} + sum_i |D_i - Dbar|^2_P (ymc_numerics.h) */
        double r0[YMO_MMAX];
        for (int32_t j = 0; j < p->data_dim; ++j) r0[j] = F[j] - p->lin_dbar[j];
        return -0.5 * (ymo_diag_norm_sq(r0, p->lin_nprec, p->data_dim) + p->lin_const);
    }
    double total = 0.0;
    double r[YMO_MMAX];
    for (int32_t i = 0; i < p->n_data; ++i) {
        const double *Fi = per_row_eval ? F + (int64_t)i * p->data_dim : F;
        const double *Di = p->data + (int64_t)i * p->data_dim;
        for (int32_t j = 0; j < p->data_dim; ++j) r[j] = Fi[j] - Di[j];
        total = total + (p->noise_linv ? ymc_whitened_norm_sq(p->noise_linv, r, p->data_dim)
                                       : ymo_diag_norm_sq(r, p->noise_precision, p->data_dim));
    }
    return -0.5 * total;
}

/* DensityInterface.evaluate_log (statistics/interface.py:6-10) for the lowered
 * target kinds.  *status gets YMO_STATUS_NONFINITE or'ed in when the forward
 * map fails; the value is then -inf. */
static double ymo_logdensity_one(const ymo_density *p, const double *x, uint32_t *status)
{
    switch (p->kind) {
    case YMO_GAUSS_WHITENED: {
        /* GaussianTargetDensity2d.evaluate_log, testSetup.py:38-40 ->
         * scipy.stats.multivariate_normal.logpdf:
         *   dev = x - mean; maha = sum(square(dev @ U)); -0.5*(rank*log(2pi) + log_pdet + maha)
         * with U the whitening matrix of scipy's CovViaPSD and
         * log_norm = rank*log(2pi) + log_pdet precomputed on the host. */
        double dev[YMO_DMAX];
        int32_t d = p->dim;
        for (int32_t i = 0; i < d; ++i) dev[i] = x[i] - p->mean[i];
        double maha = 0.0;
        for (int32_t j = 0; j < d; ++j) {
            double w = dev[0] * p->whiten[j];
            for (int32_t i = 1; i < d; ++i) w = ymc_fma(dev[i], p->whiten[(int64_t)i * d + j], w);
            maha = (j == 0) ? w * w : ymc_fma(w, w, maha);
        }
        return -0.5 * (p->log_norm + maha);
    }
    case YMO_GAUSS_ISO1D: {
        /* GaussianTargetDensity1d.evaluate_log, testSetup.py:22-25:
         *   -0.5 * square(mean - x) / var */
        double t = p->mean[0] - x[0];
        return (-0.5 * (t * t)) / p->variance;
    }
    case YMO_POST_LINEAR: {
        /* UnnormalisedPosterior.evaluate_log, target.py:19-22: logL + log prior */
        double F[YMO_MMAX];
        ymo_forward_linear(p, x, F);
        if (p->likelihood_only) return ymo_log_likelihood(p, F, 0);
        return ymo_log_likelihood(p, F, 0) + ymo_gaussian_prior_log(p, x);
    }
    case YMO_POST_LV: {
        double F[YMO_MMAX];
        if (ymo_forward_lv(p, x, F)) {
            /* ForwardModel.evaluate raises on failure (forwardModel.py:15-19);
             * per-chain equivalent: flag the chain, reject the proposal. */
            if (status) *status |= YMO_STATUS_NONFINITE;
            return -INFINITY;
        }
        if (p->likelihood_only) return ymo_log_likelihood(p, F, 1);
        return ymo_log_likelihood(p, F, 1) + ymo_gaussian_prior_log(p, x);
    }
    case YMO_POST_LOGISTIC: {
        double F[YMO_MMAX];
        if (ymo_forward_logistic(p, x, F)) {
            if (status) *status |= YMO_STATUS_NONFINITE;
            return -INFINITY;
        }
        if (p->likelihood_only) return ymo_log_likelihood(p, F, 1);
        return ymo_log_likelihood(p, F, 1) + ymo_gaussian_prior_log(p, x);
    }
    default:
        if (status) *status |= YMO_STATUS_NONFINITE;
        return NAN;
    }
}

void ymo_logdensity(const ymo_density *p, int64_t n, const double *x, double *out, uint32_t *status)
{
    for (int64_t i = 0; i < n; ++i) {
        uint32_t st = 0;
        out[i] = ymo_logdensity_one(p, x + i * (int64_t)p->dim, &st);
        if (status) status[i] = st;
    }
}

/* the forward map alone (ForwardModel.evaluate, forwardModel.py:10-19) */
int ymo_forward(const ymo_density *p, int64_t n, const double *x, double *out)
{
    int bad = 0;
    int64_t m = (p->kind == YMO_POST_LV) ? (int64_t)p->n_data * 2 : (p->kind == YMO_POST_LOGISTIC) ? (int64_t)p->n_data : p->data_dim;
    for (int64_t i = 0; i < n; ++i) {
        if (p->kind == YMO_POST_LINEAR) ymo_forward_linear(p, x + i * p->dim, out + i * m);
        else if (p->kind == YMO_POST_LV) bad |= ymo_forward_lv(p, x + i * p->dim, out + i * m);
        else if (p->kind == YMO_POST_LOGISTIC) bad |= ymo_forward_logistic(p, x + i * p->dim, out + i * m);
        else return -1;
    }
    return bad;
}

/* ---------------------------------------------------------------------- */
/* one chain: proposal, accept/reject, diagnostics                        */
/* ---------------------------------------------------------------------- */

/* ParameterVector.__eq__ (vector.py:37-45: numpy.array_equal) or
 * ScalarParameter.__eq__ (scalar.py:39-44: math.isclose, rel_tol 1e-9). */
static int ymo_states_equal(const double *a, const double *b, int32_t d, int eq_mode)
{
    if (eq_mode == YMO_EQ_ISCLOSE) {
        double x = a[0], y = b[0];
        if (x == y) return 1;
        if (isinf(x) || isinf(y)) return 0;
        double diff = fabs(y - x);
        return (diff <= fabs(1e-9 * y)) || (diff <= fabs(1e-9 * x));
    }
    for (int32_t j = 0; j < d; ++j)
        if (!(a[j] == b[j])) return 0;
    return 1;
}

/* MRWProposal.generate_proposal (mrw.py:32-38) -> Gaussian.generate_realisation
 * (gaussian.py:61-66) with DiagonalCovarianceMatrix.apply_chol_factor
 * (covariance.py:51-52): y = x + sigma * xi, separate multiply and add.
 * sigma = sqrt(reciprocal(precision)) is formed on the host exactly as NumPy
 * does. */
static void ymo_propose(const double *x, const double *sigma, uint64_t seed, uint32_t chain,
                        uint64_t t, int32_t d, double *y)
{
    for (int32_t j = 0; 2 * j < d; ++j) {
        double z0, z1;
        ymc_draw_normal_pair(seed, chain, t, (uint32_t)j, &z0, &z1);
        y[2 * j] = x[2 * j] + sigma[2 * j] * z0;
        if (2 * j + 1 < d) y[2 * j + 1] = x[2 * j + 1] + sigma[2 * j + 1] * z1;
    }
}

/* MetropolisHastings._accept_reject, metropolisHastings.py:55-73, given the
 * log of the acceptance ratio:
 *   a = exp(log_ratio); a = a if a < 1 else 1      (mrw.py:54-57, mlda.py:148-154)
 *   u = uniform();  ACCEPTED iff u <= a            (:68-70)
 * Deviation: the reference's `a if a < 1. else 1.` turns a NaN ratio into
 * a = 1 (always accepted).  Here a proposal whose log-density is not finite
 * (NaN, or -inf from a failed forward solve) is rejected and the chain is
 * flagged YMO_STATUS_NONFINITE.  `margin` receives |u - a| (SURVEY.md
 * section 7, "Accept test form"). */
static int ymo_accept(double log_ratio, int proposal_finite, uint64_t seed, uint32_t chain,
                      uint64_t t, uint32_t *status, double *margin)
{
    if (!proposal_finite) {
        *status |= YMO_STATUS_NONFINITE;
        if (margin) *margin = INFINITY;
        return 0;
    }
    double a = ymc_accept_prob(log_ratio); /* = exp(log_ratio) clipped at 1, see ymc_numerics.h */
    double u = ymc_draw_uniform(seed, chain, t);
    if (margin) *margin = fabs(u - a);
    return u <= a;
}

/* WelfordAccumulator.update, estimation.py:36-53 */
static void ymo_welford(const double *x, int32_t d, uint64_t *n, double *mean, double *m2)
{
    *n += 1;
    double dn = (double)*n;
    for (int32_t j = 0; j < d; ++j) {
        double delta = x[j] - mean[j];
        mean[j] = mean[j] + delta / dn;
        double delta2 = x[j] - mean[j];
        m2[j] = m2[j] + delta * delta2;
    }
}

static int ymo_is_finite(double v) { return fabs(v) <= 1.79769313486231570815e308; }

/* chains per cache block of the ensemble loops */
static int64_t ymo_block_size(int32_t d)
{
    int64_t b = 512 / (d > 0 ? d : 1);
    return b < 1 ? 1 : (b > 64 ? 64 : b);
}

/*
 * MetropolisHastings.run, metropolisHastings.py:103-120, for K independent
 * chains of the Metropolised random walk (mrw.py:41-57), advanced by `nsteps`
 * transitions t0 .. t0+nsteps-1 from the carried state.
 *
 *   x[K][d], lp[K]             state and logpi(state), updated in place
 *   acc_count[K]               accepted transitions (diagnostics.py:48-54)
 *   wf_n[K], wf_mean, wf_m2    FullDiagnostics' Welford over the
 *                              PRE-transition state (diagnostics.py:91-94)
 *   status[K]                  or'ed YMO_STATUS_*
 *   traj[nsteps][K][d]         state after each transition (:75-76), or NULL
 *   acc[nsteps][K]             0/1 outcome (transition.py:3-4), or NULL
 *   lp_out[nsteps][K]          logpi(state after the transition), or NULL
 *   margin[nsteps][K]          |u - a| (inf where no uniform was drawn), or NULL
 */
/* y = x + L xi for a DenseCovarianceMatrix proposal (covariance.py:69-94; mrw.py:32-38): row i of
 * L xi summed left to right with separate multiplies and adds (the reference's order is BLAS') */
static void ymo_propose_dense(const double *x, const double *chol, uint64_t seed, uint32_t chain, uint64_t t,
                              int32_t d, int fma_chain, double *y)
{
    double z[YMO_DMAX + 1];
    for (int32_t j = 0; 2 * j < d; ++j) ymc_draw_normal_pair(seed, chain, t, (uint32_t)j, &z[2 * j], &z[2 * j + 1]);
    for (int32_t i = 0; i < d; ++i) {
        double acc = chol[(int64_t)i * d] * z[0];
        if (fma_chain) /* the matrix-core kernel's order: row i of L xi as one k-ordered fma chain (v_mfma_f64) */
            for (int32_t j = 1; j <= i; ++j) acc = ymc_fma(chol[(int64_t)i * d + j], z[j], acc);
        else       /* the lane kernels' order: separate multiplies and adds, left to right */
            for (int32_t j = 1; j <= i; ++j) acc = acc + chol[(int64_t)i * d + j] * z[j];
        y[i] = x[i] + acc;
    }
}

/* chol: NULL for the diagonal proposal covariance of sigma, else its dense lower Cholesky factor */
void ymo_mrw_advance(const ymo_density *tgt, const double *sigma, int32_t eq_mode, uint64_t seed,
                     uint64_t chain0, int64_t K, uint64_t t0, int64_t nsteps, double *x, double *lp,
                     uint64_t *acc_count, uint64_t *wf_n, double *wf_mean, double *wf_m2,
                     uint32_t *status, double *traj, uint8_t *acc, double *lp_out, double *margin,
                     int32_t nthreads, const double *chol)
{
    int32_t d = tgt->dim;
    /* the path the product takes (needs_tile, ymc_api.hip): the matrix cores beyond the lane kernels' sizes */
    const int on_matrix_cores = d > 8 || (tgt->kind == YMO_POST_LINEAR && tgt->data_dim > 8);
    /* chains are independent, so they are walked in blocks (a block's record
     * rows are contiguous: cache-friendly on the CPU); per chain the order of
     * operations is exactly the reference's loop */
    const int64_t block = ymo_block_size(d);
    const int64_t nblocks = (K + block - 1) / block;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(dynamic, 1)
#endif
    for (int64_t kb = 0; kb < nblocks; ++kb) {
        const int64_t k_lo = kb * block, k_hi = (k_lo + block < K) ? k_lo + block : K;
        double y[YMO_DMAX];
        for (int64_t n = 0; n < nsteps; ++n) {
            uint64_t t = t0 + (uint64_t)n;
            for (int64_t k = k_lo; k < k_hi; ++k) {
                uint32_t chain = (uint32_t)(chain0 + (uint64_t)k);
                double *xs = x + k * d;
                uint32_t st = status ? status[k] : 0;
                if (chol)
                    ymo_propose_dense(xs, chol, seed, chain, t, d, on_matrix_cores, y);
                else
                    ymo_propose(xs, sigma, seed, chain, t, d, y);   /* :116-117 */
                int accepted = 0;
                double mg = INFINITY;
                double lpy = lp[k];
                if (!ymo_states_equal(y, xs, d, eq_mode)) {          /* :60-61 */
                    lpy = ymo_logdensity_one(tgt, y, &st);
                    accepted = ymo_accept(lpy - lp[k], ymo_is_finite(lpy), seed, chain, t, &st, &mg);
                }
                if (wf_n) ymo_welford(xs, d, &wf_n[k], wf_mean + k * d, wf_m2 + k * d);
                if (accepted) {                                      /* :79-92 */
                    memcpy(xs, y, sizeof(double) * (size_t)d);
                    lp[k] = lpy;
                    if (acc_count) acc_count[k] += 1;
                }
                if (status) status[k] = st;
                if (traj) memcpy(traj + (n * K + k) * d, xs, sizeof(double) * (size_t)d);
                if (acc) acc[n * K + k] = (uint8_t)accepted;
                if (lp_out) lp_out[n * K + k] = lp[k];
                if (margin) margin[n * K + k] = mg;
            }
        }
    }
}

/* PCNProposal.generate_proposal, pcn.py:23-36.  The prior's realisation is
 * mean + sigma_prior * z (Gaussian.generate_realisation, gaussian.py:61-66) and
 *   y = sqrt(1 - 2h) * x + sqrt(2h) * xi        (pcn.py:32-36)
 * c1 = sqrt(1 - 2h), c2 = sqrt(2h) are formed by the caller as NumPy does. */
static void ymo_propose_pcn(const double *x, const double *prior_sigma, const double *prior_mean, double c1,
                            double c2, uint64_t seed, uint32_t chain, uint64_t t, int32_t d, double *y)
{
    for (int32_t j = 0; 2 * j < d; ++j) {
        double z0, z1;
        ymc_draw_normal_pair(seed, chain, t, (uint32_t)j, &z0, &z1);
        double xi0 = prior_mean[2 * j] + prior_sigma[2 * j] * z0;
        y[2 * j] = c1 * x[2 * j] + c2 * xi0;
        if (2 * j + 1 < d) {
            double xi1 = prior_mean[2 * j + 1] + prior_sigma[2 * j + 1] * z1;
            y[2 * j + 1] = c1 * x[2 * j + 1] + c2 * xi1;
        }
    }
}

/*
 * PreconditionedCrankNicolson (pcn.py:38-57) for K chains: the loop of
 * metropolisHastings.py:103-120 with the pCN proposal and the LIKELIHOOD as
 * target (PCNBuilder.build_from_model, pcn.py:75-78: `tgt` has likelihood_only
 * set; its prior fields describe the Gaussian prior the proposal draws from).
 * Arguments as ymo_mrw_advance; prior_sigma = sqrt(reciprocal(prior precision)).
 */
void ymo_pcn_advance(const ymo_density *tgt, const double *prior_sigma, double step_size, int32_t eq_mode,
                     uint64_t seed, uint64_t chain0, int64_t K, uint64_t t0, int64_t nsteps, double *x, double *lp,
                     uint64_t *acc_count, uint64_t *wf_n, double *wf_mean, double *wf_m2, uint32_t *status,
                     double *traj, uint8_t *acc, double *lp_out, int32_t nthreads)
{
    int32_t d = tgt->dim;
    const double t2 = 2.0 * step_size;
    const double c1 = sqrt(1.0 - t2), c2 = sqrt(t2);
    const int64_t block = ymo_block_size(d);
    const int64_t nblocks = (K + block - 1) / block;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(dynamic, 1)
#endif
    for (int64_t kb = 0; kb < nblocks; ++kb) {
        const int64_t k_lo = kb * block, k_hi = (k_lo + block < K) ? k_lo + block : K;
        double y[YMO_DMAX];
        for (int64_t n = 0; n < nsteps; ++n) {
            uint64_t t = t0 + (uint64_t)n;
            for (int64_t k = k_lo; k < k_hi; ++k) {
                uint32_t chain = (uint32_t)(chain0 + (uint64_t)k);
                double *xs = x + k * d;
                uint32_t st = status ? status[k] : 0;
                ymo_propose_pcn(xs, prior_sigma, tgt->prior_mean, c1, c2, seed, chain, t, d, y);
                int accepted = 0;
                double lpy = lp[k];
                if (!ymo_states_equal(y, xs, d, eq_mode)) {          /* metropolisHastings.py:60-61 */
                    lpy = ymo_logdensity_one(tgt, y, &st);
                    accepted = ymo_accept(lpy - lp[k], ymo_is_finite(lpy), seed, chain, t, &st, NULL); /* pcn.py:51-57 */
                }
                if (wf_n) ymo_welford(xs, d, &wf_n[k], wf_mean + k * d, wf_m2 + k * d);
                if (accepted) {
                    memcpy(xs, y, sizeof(double) * (size_t)d);
                    lp[k] = lpy;
                    if (acc_count) acc_count[k] += 1;
                }
                if (status) status[k] = st;
                if (traj) memcpy(traj + (n * K + k) * d, xs, sizeof(double) * (size_t)d);
                if (acc) acc[n * K + k] = (uint8_t)accepted;
                if (lp_out) lp_out[n * K + k] = lp[k];
            }
        }
    }
}

/*
 * Two-level delayed acceptance: MLDA (mlda.py:120-154) with one surrogate,
 * i.e. MLDAProposal.generate_proposal's nSurrogates == 1 branch (:102-110):
 * the proposal of fine step n is the end point of an S-step random walk on
 * the surrogate density started at the fine state, and it is accepted with
 *   min(1, exp(pi_f(y) + pi_c(x) - pi_c(y) - pi_f(x)))        (:146-154)
 * Transition numbering: t = n*(S+1) + s, s < S coarse, s == S fine.
 * The surrogate's diagnostics are DummyDiagnostics by default (:271-280), so
 * only fine-level outcomes are accumulated; sub_acc[nsteps][S][K] (optional)
 * exposes the coarse outcomes for parity checks.
 *   lpf[K], lpc[K]   fine / coarse log-density of the carried fine state
 */
void ymo_mlda2_advance(const ymo_density *fine, const ymo_density *coarse, const double *sigma,
                       int32_t S, int32_t eq_mode, uint64_t seed, uint64_t chain0, int64_t K,
                       uint64_t n0, int64_t nsteps, double *x, double *lpf, double *lpc,
                       uint64_t *acc_count, uint64_t *wf_n, double *wf_mean, double *wf_m2,
                       uint32_t *status, double *traj, uint8_t *acc, uint8_t *sub_acc,
                       double *lp_out, double *margin, int32_t nthreads)
{
    int32_t d = fine->dim;
    const int64_t block = ymo_block_size(d);
    const int64_t nblocks = (K + block - 1) / block;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(dynamic, 1)
#endif
    for (int64_t kb = 0; kb < nblocks; ++kb) {
        const int64_t k_lo = kb * block, k_hi = (k_lo + block < K) ? k_lo + block : K;
        double z[YMO_DMAX], y[YMO_DMAX];
        for (int64_t n = 0; n < nsteps; ++n) {
            uint64_t tbase = (n0 + (uint64_t)n) * (uint64_t)(S + 1);
            for (int64_t k = k_lo; k < k_hi; ++k) {
                uint32_t chain = (uint32_t)(chain0 + (uint64_t)k);
                double *xs = x + k * d;
                uint32_t st = status ? status[k] : 0;
                /* surrogate.run(S+1, state): metropolisHastings.py:103-120 on pi_c */
                memcpy(z, xs, sizeof(double) * (size_t)d);
                double lpcz = lpc[k];
                for (int32_t s = 0; s < S; ++s) {
                    uint64_t t = tbase + (uint64_t)s;
                    ymo_propose(z, sigma, seed, chain, t, d, y);
                    int a = 0;
                    if (!ymo_states_equal(y, z, d, eq_mode)) {
                        double lpcy = ymo_logdensity_one(coarse, y, &st);
                        a = ymo_accept(lpcy - lpcz, ymo_is_finite(lpcy), seed, chain, t, &st, NULL);
                        if (a) {
                            memcpy(z, y, sizeof(double) * (size_t)d);
                            lpcz = lpcy;
                        }
                    }
                    if (sub_acc) sub_acc[(n * S + s) * K + k] = (uint8_t)a;
                }
                /* fine accept/reject of y = z, metropolisHastings.py:55-73 + mlda.py:146-154 */
                int accepted = 0;
                double mg = INFINITY;
                double lpfy = lpf[k];
                if (!ymo_states_equal(z, xs, d, eq_mode)) {
                    lpfy = ymo_logdensity_one(fine, z, &st);
                    double log_ratio = ((lpfy + lpc[k]) - lpcz) - lpf[k];
                    accepted = ymo_accept(log_ratio, ymo_is_finite(lpfy) && ymo_is_finite(lpcz), seed,
                                          chain, tbase + (uint64_t)S, &st, &mg);
                }
                if (wf_n) ymo_welford(xs, d, &wf_n[k], wf_mean + k * d, wf_m2 + k * d);
                if (accepted) {
                    memcpy(xs, z, sizeof(double) * (size_t)d);
                    lpf[k] = lpfy;
                    lpc[k] = lpcz;
                    if (acc_count) acc_count[k] += 1;
                }
                if (status) status[k] = st;
                if (traj) memcpy(traj + (n * K + k) * d, xs, sizeof(double) * (size_t)d);
                if (acc) acc[n * K + k] = (uint8_t)accepted;
                if (lp_out) lp_out[n * K + k] = lpf[k];
                if (margin) margin[n * K + k] = mg;
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Nested delayed acceptance: MLDA with an even number of surrogates as the   */
/* reference executes it (mlda.py:12-73,100-116; YMC_METHOD_MLDA_NESTED in    */
/* include/ymc.h).  P[e], Q[e]: densities of effective level e (Q[0] unused), */
/* steps[e]: steps of one run of level e < E-1.  Decisions take consecutive   */
/* transition indices *t in execution order.                                  */
/* ------------------------------------------------------------------------ */
typedef struct ymo_nest {
    const ymo_density *const *P, *const *Q;
    const int32_t *steps;
    const double *sigma;
    int32_t d, eq_mode;
    uint64_t seed;
    uint32_t chain;
    uint64_t t;
    uint32_t st;
    uint64_t *lacc; /* [E-1] accepted transitions of the lower levels of this chain */
} ymo_nest;

/* MetropolisHastings.run of effective level e from x (metropolisHastings.py:103-120) */
static void ymo_nest_run(ymo_nest *w, int32_t e, double *x)
{
    const int32_t d = w->d, S = w->steps[e];
    double y[YMO_DMAX];
    if (e == 0) {
        double lp = ymo_logdensity_one(w->P[0], x, &w->st);
        for (int32_t s = 0; s < S; ++s) {
            ymo_propose(x, w->sigma, w->seed, w->chain, w->t, d, y);
            if (!ymo_states_equal(y, x, d, w->eq_mode)) {
                double lpy = ymo_logdensity_one(w->P[0], y, &w->st);
                if (ymo_accept(lpy - lp, ymo_is_finite(lpy), w->seed, w->chain, w->t, &w->st, NULL)) {
                    memcpy(x, y, sizeof(double) * (size_t)d);
                    lp = lpy;
                    w->lacc[0] += 1;
                }
            }
            w->t += 1;
        }
        return;
    }
    double lpP = ymo_logdensity_one(w->P[e], x, &w->st);
    double lpQ = ymo_logdensity_one(w->Q[e], x, &w->st);
    for (int32_t s = 0; s < S; ++s) {
        memcpy(y, x, sizeof(double) * (size_t)d);
        ymo_nest_run(w, e - 1, y);                       /* SurrogateTransition.generate_proposal, mlda.py:23-33 */
        if (!ymo_states_equal(y, x, d, w->eq_mode)) {
            double lpPy = ymo_logdensity_one(w->P[e], y, &w->st);
            double lpQy = ymo_logdensity_one(w->Q[e], y, &w->st);
            double log_ratio = ((lpPy + lpQ) - lpQy) - lpP; /* mlda.py:35-43 */
            if (ymo_accept(log_ratio, ymo_is_finite(lpPy) && ymo_is_finite(lpQy), w->seed, w->chain, w->t,
                           &w->st, NULL)) {
                memcpy(x, y, sizeof(double) * (size_t)d);
                lpP = lpPy;
                lpQ = lpQy;
                w->lacc[e] += 1;
            }
        }
        w->t += 1;
    }
}

/* E effective levels; lp[K] = P[E-1](x), lpq[K] = Q[E-1](x) of the carried states;
 * level_acc[E-1][K]; cost_top = transitions one top-level step spans */
void ymo_mlda_nested_advance(int32_t E, const ymo_density *const *P, const ymo_density *const *Q,
                             const int32_t *steps, const double *sigma, int32_t eq_mode, uint64_t seed,
                             uint64_t chain0, int64_t K, uint64_t n0, uint64_t cost_top, int64_t nsteps,
                             double *x, double *lp, double *lpq, uint64_t *acc_count, uint64_t *level_acc,
                             uint64_t *wf_n, double *wf_mean, double *wf_m2, uint32_t *status, double *traj,
                             uint8_t *acc, double *lp_out, int32_t nthreads)
{
    const int32_t d = P[E - 1]->dim;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(dynamic, 16)
#endif
    for (int64_t k = 0; k < K; ++k) {
        uint64_t lacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        ymo_nest w;
        w.P = P;
        w.Q = Q;
        w.steps = steps;
        w.sigma = sigma;
        w.d = d;
        w.eq_mode = eq_mode;
        w.seed = seed;
        w.chain = (uint32_t)(chain0 + (uint64_t)k);
        w.st = status ? status[k] : 0;
        w.lacc = lacc;
        double *xs = x + k * d;
        double y[YMO_DMAX];
        for (int64_t n = 0; n < nsteps; ++n) {
            w.t = (n0 + (uint64_t)n) * cost_top;
            memcpy(y, xs, sizeof(double) * (size_t)d);
            ymo_nest_run(&w, E - 2, y);                  /* MLDAProposal.generate_proposal, mlda.py:100-116 */
            int accepted = 0;
            if (!ymo_states_equal(y, xs, d, eq_mode)) {
                double lpPy = ymo_logdensity_one(P[E - 1], y, &w.st);
                double lpQy = ymo_logdensity_one(Q[E - 1], y, &w.st);
                double log_ratio = ((lpPy + lpq[k]) - lpQy) - lp[k]; /* mlda.py:146-154 */
                accepted = ymo_accept(log_ratio, ymo_is_finite(lpPy) && ymo_is_finite(lpQy), seed, w.chain, w.t,
                                      &w.st, NULL);
                if (wf_n) ymo_welford(xs, d, &wf_n[k], wf_mean + k * d, wf_m2 + k * d);
                if (accepted) {
                    memcpy(xs, y, sizeof(double) * (size_t)d);
                    lp[k] = lpPy;
                    lpq[k] = lpQy;
                }
            } else if (wf_n) {
                ymo_welford(xs, d, &wf_n[k], wf_mean + k * d, wf_m2 + k * d);
            }
            if (accepted && acc_count) acc_count[k] += 1;
            if (traj) memcpy(traj + (n * K + k) * d, xs, sizeof(double) * (size_t)d);
            if (acc) acc[n * K + k] = (uint8_t)accepted;
            if (lp_out) lp_out[n * K + k] = lp[k];
        }
        if (status) status[k] = w.st;
        if (level_acc)
            for (int32_t e = 0; e < E - 1; ++e) level_acc[(int64_t)e * K + k] += lacc[e];
    }
}

/* ------------------------------------------------------------------------ */
/* Adaptive error model (AEM): two-level delayed acceptance whose coarse      */
/* likelihood learns the model error (chain/method/aem.py:6-58,               */
/* statistics/likelihood.py:90-155, noise.py:25-61), ONE surrogate, linear    */
/* forward maps.  Per chain the coarse AEMLikelihood carries                  */
/*   - a Welford accumulator of the error F_fine(y) - F_coarse(y) over the    */
/*     ACCEPTED fine proposals (aem.py:44-56, estimation.py:36-53);           */
/*   - once nData >= minDataSize the residual is shifted by its mean          */
/*     (likelihood.py:128-133); once nData > minDataSize the noise variance   */
/*     becomes scaling * mVar + dataVar (likelihood.py:142-150, noise.py:41-  */
/*     54), scaling = min(2 max(mVar) / max(min(mVar), 1e-6), 100) if the     */
/*     heuristic is on, else 1;                                               */
/*   - the 3-entry AEMCache of log-LIKELIHOOD values keyed by parameter       */
/*     equality, most recently used last (memoisation.py:76-149).  A hit      */
/*     returns the value computed when the entry was made, i.e. possibly      */
/*     under an older error model: that staleness is part of the reference's  */
/*     behaviour and is reproduced entry by entry.                            */
/* The fine AEMLikelihood's accumulator is never updated, so it stays the     */
/* plain likelihood and its cache never changes a value.                      */
/* ------------------------------------------------------------------------ */
typedef struct ymo_aem {
    const ymo_density *coarse;
    int32_t min_data, use_heuristic, d, m, eq_mode;
    uint64_t *n;      /* [1] */
    double *mean, *m2, *prec; /* [m] each */
    int32_t *has_prec, *nkeys;
    double *key;      /* [3][d] */
    double *val;      /* [3] */
} ymo_aem;

/* AEMLikelihood.compute_log_likelihood under the CURRENT error model */
static double ymo_aem_loglik_now(const ymo_aem *a, const double *theta)
{
    const ymo_density *p = a->coarse;
    double F[YMO_MMAX], r[YMO_MMAX];
    ymo_forward_linear(p, theta, F);
    const int shift = *a->n >= (uint64_t)a->min_data;
    const double *prec = *a->has_prec ? a->prec : p->noise_precision;
    double total = 0.0;
    for (int32_t i = 0; i < p->n_data; ++i) {
        const double *Di = p->data + (int64_t)i * a->m;
        for (int32_t j = 0; j < a->m; ++j) {
            r[j] = F[j] - Di[j];
            if (shift) r[j] = r[j] + a->mean[j];
        }
        total = total + ymo_diag_norm_sq(r, prec, a->m);
    }
    return -0.5 * total;
}

/* AEMLikelihood.query_log_likelihood (likelihood.py:121-126) + update_cache (:135-141) */
static double ymo_aem_query(ymo_aem *a, const double *theta)
{
    const int32_t d = a->d;
    int32_t nk = *a->nkeys;
    for (int32_t i = 0; i < nk; ++i) {
        if (ymo_states_equal(a->key + (int64_t)i * d, theta, d, a->eq_mode)) {
            /* retrieve: move entry i to the back (memoisation.py:98-103,139-147) */
            double v = a->val[i];
            for (int32_t q = i; q + 1 < nk; ++q) {
                memcpy(a->key + (int64_t)q * d, a->key + (int64_t)(q + 1) * d, sizeof(double) * (size_t)d);
                a->val[q] = a->val[q + 1];
            }
            memcpy(a->key + (int64_t)(nk - 1) * d, theta, sizeof(double) * (size_t)d);
            a->val[nk - 1] = v;
            return v;
        }
    }
    double v = ymo_aem_loglik_now(a, theta);
    if (nk >= 3) { /* evict the oldest (memoisation.py:92-96,113-114) */
        for (int32_t q = 0; q + 1 < nk; ++q) {
            memcpy(a->key + (int64_t)q * d, a->key + (int64_t)(q + 1) * d, sizeof(double) * (size_t)d);
            a->val[q] = a->val[q + 1];
        }
        nk = 2;
    }
    memcpy(a->key + (int64_t)nk * d, theta, sizeof(double) * (size_t)d);
    a->val[nk] = v;
    *a->nkeys = nk + 1;
    return v;
}

/* AEMLikelihood.update_error_estimate (likelihood.py:142-150) with error = F_fine(y) - F_coarse(y) */
static void ymo_aem_update(ymo_aem *a, const ymo_density *fine, const double *y)
{
    double Ff[YMO_MMAX], Fc[YMO_MMAX], err[YMO_MMAX];
    if (a->m < 1) return; /* (a model without outputs has no error to learn; also tells the compiler err is written) */
    ymo_forward_linear(fine, y, Ff);
    ymo_forward_linear(a->coarse, y, Fc);
    for (int32_t j = 0; j < a->m; ++j) err[j] = Ff[j] - Fc[j];
    ymo_welford(err, a->m, a->n, a->mean, a->m2);
    if (*a->n > (uint64_t)a->min_data) {
        double mvar[YMO_MMAX];
        const double nm1 = (double)(*a->n - 1);
        double lo = 0.0, hi = 0.0;
        for (int32_t j = 0; j < a->m; ++j) {
            mvar[j] = a->m2[j] / nm1;
            lo = (j == 0 || mvar[j] < lo) ? mvar[j] : lo;
            hi = (j == 0 || mvar[j] > hi) ? mvar[j] : hi;
        }
        double scaling = 1.0;
        if (a->use_heuristic) { /* noise.py:41-46 */
            double minval = lo > 1e-6 ? lo : 1e-6;
            scaling = 2.0 * hi / minval;
            scaling = scaling < 100.0 ? scaling : 100.0;
        }
        for (int32_t j = 0; j < a->m; ++j) {
            double data_var = 1.0 / a->coarse->noise_precision[j]; /* covariance.py:40-41 */
            a->prec[j] = 1.0 / (scaling * mvar[j] + data_var);   /* noise.py:51-53, covariance.py:36-37 */
        }
        *a->has_prec = 1;
    }
}

/* state arrays per chain k: aem_n[k], aem_mean/m2/prec[k][m], aem_flags[k][2] = {has_prec, nkeys},
 * cache_key[k][3][d], cache_val[k][3]; lpf[k] = fine log-posterior of the carried state */
void ymo_aem_advance(const ymo_density *fine, const ymo_density *coarse, const double *sigma, int32_t S,
                     int32_t min_data, int32_t use_heuristic, int32_t eq_mode, uint64_t seed, uint64_t chain0,
                     int64_t K, uint64_t n0, int64_t nsteps, double *x, double *lpf, uint64_t *acc_count,
                     uint64_t *wf_n, double *wf_mean, double *wf_m2, uint32_t *status, uint64_t *aem_n,
                     double *aem_mean, double *aem_m2, double *aem_prec, int32_t *aem_flags, double *cache_key,
                     double *cache_val, double *traj, uint8_t *acc, uint8_t *sub_acc, double *lp_out,
                     int32_t nthreads)
{
    const int32_t d = fine->dim, m = coarse->data_dim;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(dynamic, 16)
#endif
    for (int64_t k = 0; k < K; ++k) {
        ymo_aem a;
        a.coarse = coarse;
        a.min_data = min_data;
        a.use_heuristic = use_heuristic;
        a.d = d;
        a.m = m;
        a.eq_mode = eq_mode;
        a.n = aem_n + k;
        a.mean = aem_mean + k * m;
        a.m2 = aem_m2 + k * m;
        a.prec = aem_prec + k * m;
        a.has_prec = aem_flags + 2 * k;
        a.nkeys = aem_flags + 2 * k + 1;
        a.key = cache_key + k * 3 * d;
        a.val = cache_val + k * 3;
        const uint32_t chain = (uint32_t)(chain0 + (uint64_t)k);
        double *xs = x + k * d;
        uint32_t st = status ? status[k] : 0;
        double z[YMO_DMAX], y[YMO_DMAX];
        for (int64_t n = 0; n < nsteps; ++n) {
            const uint64_t tbase = (n0 + (uint64_t)n) * (uint64_t)(S + 1);
            memcpy(z, xs, sizeof(double) * (size_t)d);
            for (int32_t s = 0; s < S; ++s) {
                const uint64_t t = tbase + (uint64_t)s;
                ymo_propose(z, sigma, seed, chain, t, d, y);
                int acc_s = 0;
                if (!ymo_states_equal(y, z, d, eq_mode)) {
                    /* mrw.py:51-57: evaluate_log(proposal) first, then evaluate_log(state) */
                    double ly = ymo_aem_query(&a, y) + ymo_gaussian_prior_log(coarse, y);
                    double lz = ymo_aem_query(&a, z) + ymo_gaussian_prior_log(coarse, z);
                    acc_s = ymo_accept(ly - lz, ymo_is_finite(ly), seed, chain, t, &st, NULL);
                    if (acc_s) memcpy(z, y, sizeof(double) * (size_t)d);
                }
                if (sub_acc) sub_acc[(n * S + s) * K + k] = (uint8_t)acc_s;
            }
            int accepted = 0;
            if (!ymo_states_equal(z, xs, d, eq_mode)) {
                /* mlda.py:146-154: tgt(y), finest(x), finest(y), tgt(x) in this order */
                double lpfy = ymo_logdensity_one(fine, z, &st);
                double cx = ymo_aem_query(&a, xs) + ymo_gaussian_prior_log(coarse, xs);
                double cy = ymo_aem_query(&a, z) + ymo_gaussian_prior_log(coarse, z);
                double log_ratio = ((lpfy + cx) - cy) - lpf[k];
                accepted = ymo_accept(log_ratio, ymo_is_finite(lpfy) && ymo_is_finite(cy), seed, chain,
                                      tbase + (uint64_t)S, &st, NULL);
                if (wf_n) ymo_welford(xs, d, &wf_n[k], wf_mean + k * d, wf_m2 + k * d);
                if (accepted) {
                    ymo_aem_update(&a, fine, z); /* aem.py:26-56, before the chain moves */
                    memcpy(xs, z, sizeof(double) * (size_t)d);
                    lpf[k] = lpfy;
                    if (acc_count) acc_count[k] += 1;
                }
            } else if (wf_n) {
                ymo_welford(xs, d, &wf_n[k], wf_mean + k * d, wf_m2 + k * d);
            }
            if (traj) memcpy(traj + (n * K + k) * d, xs, sizeof(double) * (size_t)d);
            if (acc) acc[n * K + k] = (uint8_t)accepted;
            if (lp_out) lp_out[n * K + k] = lpf[k];
        }
        if (status) status[k] = st;
    }
}

/* ymc_lower_times (ymc_numerics.h): the whitening of a linear problem with a dense noise covariance beyond the
 * lane kernels' sizes, shared with the product's lowering so that both hold the same matrices */
void ymo_lower_times(const double *linv, int32_t m, const double *src, int32_t cols, double *out)
{
    ymc_lower_times(linv, m, src, cols, out);
}

void ymo_linear_summary(const double *data, const double *prec, int32_t n, int32_t m, double *dbar,
                        double *nprec, double *cst)
{
    ymc_linear_data_summary(data, prec, n, m, dbar, nprec, cst);
}

int32_t ymo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
