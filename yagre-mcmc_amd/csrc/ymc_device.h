/*
 * ymc_device.h -- device-side building blocks of the chain-ensemble kernels:
 * the lowered densities, the proposal, the accept/reject rule and the
 * per-chain diagnostics.  One lane owns one chain; its state, log-density and
 * counters live in registers for the whole launch.
 *
 * The arithmetic (operation order, where an fma is used) is the normative
 * order of the ensemble and is mirrored operation by operation by the CPU
 * checker, so results are compared bit-for-bit.  Built with
 * -ffp-contract=off: a*b+c below is a separate multiply and add, as in the
 * reference's NumPy expressions.
 */
#ifndef YMC_DEVICE_H
#define YMC_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ymc.h"
#include "ymc_numerics.h"

#define YMC_LANE_DMAX 8 /* largest parameter dimension of the lane-per-chain kernels */
#define YMC_LANE_MMAX 8 /* largest forward-map output dimension they hold in registers */
#define YMC_DBL_MAX 1.79769313486231570815e308

/* A lowered density, passed by value as a kernel argument: the small blocks
 * (mean, whitening matrix, constants) arrive in SGPRs; the posterior's
 * arrays are read through uniform (scalar) loads. */
template <int D>
struct DensityDev {
    double mean[D];
    double whiten[D * D];
    double log_norm;
    double variance;
    int32_t n_data, data_dim, rk4_steps, likelihood_only;
    const double *data, *noise_prec, *prior_mean, *prior_prec, *A, *b, *design;
    double t_final, alpha, gamma;
    /* POSTERIOR_LINEAR: ymc_linear_data_summary of (data, noise_prec) */
    const double *dbar, *nprec;
    double lik_const;
    /* dense noise / prior covariance: inverse Cholesky factors (null: diagonal), ymc_whitened_norm_sq */
    const double *noise_linv, *prior_linv;
    /* POSTERIOR_LV: lv_rtol > 0 = the adaptive integrator (ymc_lv_dp45_solve) instead of rk4_steps steps of RK4 */
    double lv_rtol, lv_atol;
};

/* ymc_whitened_norm_sq for a register-resident residual (same operations in the same order) */
template <int N>
__device__ __forceinline__ double ymc_whitened_norm_sq_reg(const double *__restrict__ linv, const double (&r)[N], int n)
{
    double S[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i < n) {
            double w = linv[i * n] * r[0];
#pragma unroll
            for (int j = 1; j <= i; ++j) w = ymc_fma(linv[i * n + j], r[j], w);
            S[i & 3] = S[i & 3] + w * w;
        }
    }
    return (S[0] + S[1]) + (S[2] + S[3]);
}

__device__ __forceinline__ bool ymc_finite(double v) { return __builtin_fabs(v) <= YMC_DBL_MAX; }

/* covariance.py:19-22,54-55: dot(x, precision * x) as four interleaved partial
 * sums (j mod 4), combined (S0 + S1) + (S2 + S3): the ensemble's canonical order */
template <int N>
__device__ __forceinline__ double ymc_diag_norm_sq(const double (&x)[N], const double *__restrict__ prec, int n)
{
    double S[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (j < n) {
            double px = prec[j] * x[j];
            S[j & 3] = S[j & 3] + x[j] * px;
        }
    }
    return (S[0] + S[1]) + (S[2] + S[3]);
}

/* gaussian.py:19-24 */
template <int D>
__device__ __forceinline__ double ymc_prior_log(const DensityDev<D> &p, const double (&theta)[D])
{
    double x[D];
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = theta[j] - p.prior_mean[j];
    if (p.prior_linv) return -0.5 * ymc_whitened_norm_sq_reg<D>(p.prior_linv, x, D); /* wave-uniform */
    return -0.5 * ymc_diag_norm_sq<D>(x, p.prior_prec, D);
}

template <int KIND, int D>
struct Density;

/* GaussianTargetDensity2d.evaluate_log (testSetup.py:38-40 -> scipy logpdf) */
template <int D>
struct Density<YMC_DENSITY_GAUSS_WHITENED, D> {
    static __device__ __forceinline__ double eval(const DensityDev<D> &p, const double (&x)[D], uint32_t &st)
    {
        (void)st;
        double dev[D];
#pragma unroll
        for (int i = 0; i < D; ++i) dev[i] = x[i] - p.mean[i];
        /* dev @ U and the sum of squares as fma chains (scipy leaves the order to BLAS / pairwise
         * summation; values agree with it to 1e-15) */
        double maha = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double w = dev[0] * p.whiten[j];
#pragma unroll
            for (int i = 1; i < D; ++i) w = ymc_fma(dev[i], p.whiten[i * D + j], w);
            maha = (j == 0) ? w * w : ymc_fma(w, w, maha);
        }
        return -0.5 * (p.log_norm + maha);
    }
};

/* GaussianTargetDensity1d.evaluate_log (testSetup.py:22-25) */
template <int D>
struct Density<YMC_DENSITY_GAUSS_ISO1D, D> {
    static __device__ __forceinline__ double eval(const DensityDev<D> &p, const double (&x)[D], uint32_t &st)
    {
        (void)st;
        double t = p.mean[0] - x[0];
        return (-0.5 * (t * t)) / p.variance;
    }
};

/* UnnormalisedPosterior.evaluate_log (target.py:19-22) with the linear forward
 * map (exampleSetup.py:42-47) held per lane: data_dim <= YMC_LANE_MMAX */
template <int D>
struct Density<YMC_DENSITY_POSTERIOR_LINEAR, D> {
    static __device__ __forceinline__ void forward(const DensityDev<D> &p, const double (&x)[D],
                                                   double (&F)[YMC_LANE_MMAX])
    {
#pragma unroll
        for (int m = 0; m < YMC_LANE_MMAX; ++m) {
            F[m] = 0.0;
            if (m < p.data_dim) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) acc = ymc_fma(p.A[m * D + j], x[j], acc);
                F[m] = acc + p.b[m];
            }
        }
    }
    static __device__ __forceinline__ double eval(const DensityDev<D> &p, const double (&x)[D], uint32_t &st)
    {
        (void)st;
        double F[YMC_LANE_MMAX];
        forward(p, x, F);
        if (p.noise_linv) { /* dense noise covariance: row by row (likelihood.py:33-39,80) */
            double total = 0.0;
            for (int i = 0; i < p.n_data; ++i) {
                double ri[YMC_LANE_MMAX];
#pragma unroll
                for (int j = 0; j < YMC_LANE_MMAX; ++j) ri[j] = (j < p.data_dim) ? F[j] - p.data[i * p.data_dim + j] : 0.0;
                total = total + ymc_whitened_norm_sq_reg<YMC_LANE_MMAX>(p.noise_linv, ri, p.data_dim);
            }
            const double logl_dense = -0.5 * total;
            return p.likelihood_only ? logl_dense : logl_dense + ymc_prior_log<D>(p, x);
        }
        /* likelihood.py:74-84, :33-39 through the row-mean identity (ymc_linear_data_summary) */
        double r[YMC_LANE_MMAX];
#pragma unroll
        for (int j = 0; j < YMC_LANE_MMAX; ++j) r[j] = (j < p.data_dim) ? F[j] - p.dbar[j] : 0.0;
        const double total = ymc_diag_norm_sq<YMC_LANE_MMAX>(r, p.nprec, p.data_dim) + p.lik_const;
        double logl = -0.5 * total;
        return p.likelihood_only ? logl : logl + ymc_prior_log<D>(p, x);
    }
};

/* one design row of the Lotka-Volterra forward map by the adaptive integrator; a failed solve reads as NaN */
__device__ __forceinline__ void ymc_lv_solve_adaptive(double &x, double &y, double alpha, double beta, double gamma,
                                                      double delta, double t_final, double rtol, double atol)
{
    if (!ymc_lv_dp45_solve(&x, &y, alpha, beta, gamma, delta, t_final, rtol, atol, YMC_LV_MAX_ATTEMPTS))
        x = y = __builtin_nan("");
}

/* the same with the Lotka-Volterra forward map (testSetup.py:101-141),
 * integrated by rk4_steps steps of classical RK4 per design row, or adaptively (lv_rtol > 0) */
template <int D>
struct Density<YMC_DENSITY_POSTERIOR_LV, D> {
    static __device__ __forceinline__ double eval(const DensityDev<D> &p, const double (&th)[D], uint32_t &st)
    {
        static_assert(D == 2, "Lotka-Volterra has two parameters");
        const double beta = ymc_exp(th[0]);
        const double delta = ymc_exp(th[1]);
        const double alpha = p.alpha, gamma = p.gamma;
        const double h = p.t_final / (double)p.rk4_steps;
        const double h2 = 0.5 * h;
        const double h6 = h / 6.0;
        double total = 0.0;
        bool bad = false;
        for (int n = 0; n < p.n_data; ++n) {
            double x = p.design[2 * n], y = p.design[2 * n + 1];
            if (p.lv_rtol > 0.0) {
                ymc_lv_solve_adaptive(x, y, alpha, beta, gamma, delta, p.t_final, p.lv_rtol, p.lv_atol);
            } else {
                int s = 0;
#pragma unroll 1
                for (; s + 5 <= p.rk4_steps; s += 5) { /* five steps per iteration: the loop branch is a visible bubble */
#pragma unroll
                    for (int r = 0; r < 5; ++r) ymc_lv_rk4_step(&x, &y, alpha, beta, gamma, delta, h, h2, h6);
                }
#pragma unroll 1
                for (; s < p.rk4_steps; ++s) ymc_lv_rk4_step(&x, &y, alpha, beta, gamma, delta, h, h2, h6);
            }
            bad = bad || !ymc_finite(x) || !ymc_finite(y);
            double r[2] = {x - p.data[2 * n], y - p.data[2 * n + 1]};
            total = total + (p.noise_linv ? ymc_whitened_norm_sq_reg<2>(p.noise_linv, r, 2)
                                          : ymc_diag_norm_sq<2>(r, p.noise_prec, 2));
        }
        double lp = p.likelihood_only ? -0.5 * total : -0.5 * total + ymc_prior_log<D>(p, th);
        if (bad) {
            /* forwardModel.py:15-19 raises; per chain: flag it, reject the proposal */
            st |= YMC_CHAIN_NONFINITE;
            lp = -__builtin_inf();
        }
        return lp;
    }
};

/* vector.py:37-45 (array_equal) / scalar.py:39-44 (math.isclose, rel_tol 1e-9) */
template <int D>
__device__ __forceinline__ bool ymc_states_equal(const double (&a)[D], const double (&b)[D], int eq_mode)
{
    if (eq_mode == YMC_EQ_ISCLOSE) {
        double x = a[0], y = b[0];
        double diff = __builtin_fabs(y - x);
        bool close = (diff <= __builtin_fabs(1e-9 * y)) || (diff <= __builtin_fabs(1e-9 * x));
        bool anyinf = __builtin_isinf(x) || __builtin_isinf(y);
        return (x == y) || (!anyinf && close);
    }
    bool eq = true;
#pragma unroll
    for (int j = 0; j < D; ++j) eq = eq && (a[j] == b[j]);
    return eq;
}

/* mrw.py:32-38 -> gaussian.py:61-66 -> covariance.py:51-52: y = x + sigma * xi */
template <int D>
__device__ __forceinline__ void ymc_propose(const double (&x)[D], const double (&sigma)[D], uint64_t seed,
                                            uint32_t chain, uint64_t t, double (&y)[D])
{
#pragma unroll
    for (int j = 0; 2 * j < D; ++j) {
        double z0, z1;
        ymc_draw_normal_pair(seed, chain, t, (uint32_t)j, &z0, &z1);
        y[2 * j] = x[2 * j] + sigma[2 * j] * z0;
        if (2 * j + 1 < D) y[2 * j + 1] = x[2 * j + 1] + sigma[2 * j + 1] * z1;
    }
}

/* the same with a DENSE proposal covariance (covariance.py:69-94): y = x + L xi, L the lower Cholesky
 * factor (row-major, read through uniform loads).  Row i of L xi is summed left to right with
 * separate multiplies and adds -- the ensemble's order; the reference leaves it to BLAS' dgemv, so
 * states agree with it to rounding (1e-15), not bit for bit. */
template <int D>
__device__ __forceinline__ void ymc_propose_dense(const double (&x)[D], const double *__restrict__ chol,
                                                  uint64_t seed, uint32_t chain, uint64_t t, double (&y)[D])
{
    double z[D + 1];
#pragma unroll
    for (int j = 0; 2 * j < D; ++j) ymc_draw_normal_pair(seed, chain, t, (uint32_t)j, &z[2 * j], &z[2 * j + 1]);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double acc = chol[i * D] * z[0];
#pragma unroll
        for (int j = 1; j <= i; ++j) acc = acc + chol[i * D + j] * z[j];
        y[i] = x[i] + acc;
    }
}

/* PCNProposal.generate_proposal, pcn.py:23-36: xi = prior mean + sigma_prior * z (the prior's
 * realisation, gaussian.py:61-66), y = sqrt(1 - 2h) * x + sqrt(2h) * xi */
template <int D>
__device__ __forceinline__ void ymc_propose_pcn(const double (&x)[D], const double (&sigma)[D],
                                                const double *__restrict__ prior_mean, double c1, double c2,
                                                uint64_t seed, uint32_t chain, uint64_t t, double (&y)[D])
{
#pragma unroll
    for (int j = 0; 2 * j < D; ++j) {
        double z0, z1;
        ymc_draw_normal_pair(seed, chain, t, (uint32_t)j, &z0, &z1);
        const double xi0 = prior_mean[2 * j] + sigma[2 * j] * z0;
        y[2 * j] = c1 * x[2 * j] + c2 * xi0;
        if (2 * j + 1 < D) {
            const double xi1 = prior_mean[2 * j + 1] + sigma[2 * j + 1] * z1;
            y[2 * j + 1] = c1 * x[2 * j + 1] + c2 * xi1;
        }
    }
}

/* the uniforms of transitions 2p and 2p+1 share one Philox block */
struct UniformCache {
    ymc_u32x4 block;
    uint64_t pair;
    bool valid;
};

__device__ __forceinline__ double ymc_uniform_cached(UniformCache &c, uint64_t seed, uint32_t chain, uint64_t t)
{
#if defined(YMC_DRAW_SPEC_V2X)
    (void)c;
    return ymc_draw_uniform(seed, chain, t); /* experiment: the transition's own block (shared with its first normal pair) */
#endif
    uint64_t p = t >> 1;
    if (!c.valid || c.pair != p) { /* wave-uniform: t is the same in every lane */
        c.block = ymc_uniform_block(seed, chain, t);
        c.pair = p;
        c.valid = true;
    }
    return ymc_uniform_from_block(c.block, t);
}

/* metropolisHastings.py:55-73 given the log acceptance ratio; a non-finite
 * proposal density is rejected and flagged (the reference's NaN -> accept
 * hole, mrw.py:57, is not reproduced) */
__device__ __forceinline__ bool ymc_accept(double log_ratio, bool moved, bool finite, double u, uint32_t &st)
{
    const double a = ymc_accept_prob(log_ratio);
    if (moved && !finite) st |= YMC_CHAIN_NONFINITE;
    return moved && finite && (u <= a);
}

/* recip[n] of the launch's reciprocal table, read through the SCALAR data cache (constant address
 * space): a vector load here would share the vmcnt counter with the record stores, and on gfx9-family
 * hardware waiting for it means waiting for every store issued before it -- at one wave per SIMD a
 * stall of the whole SIMD per transition.  The table is written by recip_kernel before the launch. */
__device__ __forceinline__ double ymc_recip_at(const double *table, uint32_t n)
{
    typedef const double __attribute__((address_space(4))) *const_ptr;
    return ((const_ptr)(uintptr_t)table)[n];
}

/* estimation.py:36-53 with n already incremented.  Every lane divides by the
 * same sample count dn, so delta / dn is formed from rdn = RN(1 / dn) (one IEEE
 * division per transition for the whole ensemble, recip_kernel) by Markstein's
 * correction -- the correctly rounded quotient, bit for bit the `/` of the
 * reference and of the CPU checker. */
template <int D>
__device__ __forceinline__ void ymc_welford(const double (&x)[D], double dn, double rdn, double (&mean)[D],
                                            double (&m2)[D])
{
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double delta = x[j] - mean[j];
        mean[j] = mean[j] + ymc_div_by_recip(delta, dn, rdn);
        double delta2 = x[j] - mean[j];
        m2[j] = m2[j] + delta * delta2;
    }
}

/* Thinned record (ymc_run_desc.record_stride): of the transitions of a run only every stride-th one is
 * written to the trajectory / log-density record.  `since` counts the transitions made after the last
 * recorded one (carried from launch to launch by the host), `row` the rows this launch has written.
 * Wave-uniform: scalar registers and scalar branches only. */
struct RecordClock {
    uint32_t stride, since, row;
    __device__ __forceinline__ RecordClock(uint32_t stride_, uint32_t since_)
        : stride(stride_ ? stride_ : 1u), since(since_), row(0u)
    {
    }
    /* called once per transition: is this one recorded?  (the caller writes row `row`, then calls wrote()) */
    __device__ __forceinline__ bool due()
    {
        if (++since >= stride) {
            since = 0u;
            return true;
        }
        return false;
    }
    __device__ __forceinline__ void wrote() { ++row; }
    /* start n transitions into the launch (a segment of it): the state after n calls of due() / wrote() */
    __device__ __forceinline__ void skip(uint32_t n)
    {
        const uint32_t total = since + n;
        row += total / stride;
        since = total % stride;
    }
};

template <int D>
__device__ __forceinline__ void ymc_load_row(const double *__restrict__ base, uint64_t k, double (&v)[D])
{
    if constexpr (D % 2 == 0) {
        const double2 *p = reinterpret_cast<const double2 *>(base + k * D);
#pragma unroll
        for (int j = 0; j < D / 2; ++j) {
            double2 q = p[j];
            v[2 * j] = q.x;
            v[2 * j + 1] = q.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) v[j] = base[k * D + j];
    }
}

template <int D>
__device__ __forceinline__ void ymc_store_row(double *__restrict__ base, uint64_t k, const double (&v)[D])
{
    if constexpr (D % 2 == 0) {
        double2 *p = reinterpret_cast<double2 *>(base + k * D);
#pragma unroll
        for (int j = 0; j < D / 2; ++j) p[j] = make_double2(v[2 * j], v[2 * j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) base[k * D + j] = v[j];
    }
}

#endif /* YMC_DEVICE_H */
