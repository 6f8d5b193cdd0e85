/* ymc_kinds.h -- the extension point for new density / forward-model kinds.
 *
 * The reference accepts any DensityInterface / SolverInterface (statistics/interface.py:6-10, model/interface.py:7-67);
 * the device runs what has a lowering.  A new kind = one header under kinds/ + one line in ymc_kinds.def; this header
 * provides what such a header builds on and includes the registered ones.
 */
#ifndef YMC_KINDS_H
#define YMC_KINDS_H

#include "ymc_device.h"

#if defined(__HIPCC__)
/* The posterior of an AdditiveGaussianNoiseLikelihood whose forward model is solved once per DESIGN ROW
 * (likelihood.py:74-84 with an evaluation of shape (n_data, outputs): row n of F pairs with row n of the data) and a
 * Gaussian prior (gaussian.py:19-24): rows in order, each row's residual norm in the noise precision (diagonal, or
 * dense through the inverse Cholesky factor), `total = total + norm`; a non-finite forward value flags the chain and
 * reads as -inf (forwardModel.py:15-19 raises; per chain: reject).  KIND supplies forward_row() and kOutputs.
 * This is the order the checker's ymo_log_likelihood(.., per_row_eval = 1) sums in. */
template <int KIND, int D>
__device__ __forceinline__ double ymc_rowwise_posterior(const DensityDev<D> &p, const double (&th)[D], uint32_t &st)
{
    constexpr int M = Density<KIND, D>::kOutputs;
    double total = 0.0;
    bool bad = false;
    for (int n = 0; n < p.n_data; ++n) {
        double F[M], r[M];
        Density<KIND, D>::forward_row(p, th, n, F);
#pragma unroll
        for (int j = 0; j < M; ++j) {
            bad = bad || !ymc_finite(F[j]);
            r[j] = F[j] - p.data[M * n + j];
        }
        total = total + (p.noise_linv ? ymc_whitened_norm_sq_reg<M>(p.noise_linv, r, M) : ymc_diag_norm_sq<M>(r, p.noise_prec, M));
    }
    double lp = p.likelihood_only ? -0.5 * total : -0.5 * total + ymc_prior_log<D>(p, th);
    if (bad) {
        st |= YMC_CHAIN_NONFINITE;
        lp = -__builtin_inf();
    }
    return lp;
}
#endif

/* the registered kinds: one #include per line of ymc_kinds.def */
#if defined(__HIPCC__)
#include "kinds/logistic.h"
#endif

/* host-side table, generated from ymc_kinds.def: dimension range, outputs per design row, design columns and the
 * description check of every extension kind */
struct ymc_ext_kind_info {
    int kind, dmin, dmax, outputs, design_cols;
    const char *name;
#if defined(__HIPCC__)
    const char *(*check)(const ymc_density_desc &);
#endif
};

#if defined(__HIPCC__)
inline const ymc_ext_kind_info *ymc_ext_kind(int kind)
{
    static const ymc_ext_kind_info table[] = {
#define YMC_EXT_KIND(ENUM, NAME, DMIN, DMAX, OUT, COLS) {ENUM, DMIN, DMAX, OUT, COLS, #NAME, ymc_kind_check_##NAME},
#include "ymc_kinds.def"
#undef YMC_EXT_KIND
    };
    for (const ymc_ext_kind_info &e : table)
        if (e.kind == kind) return &e;
    return nullptr;
}
#endif

#endif /* YMC_KINDS_H */
