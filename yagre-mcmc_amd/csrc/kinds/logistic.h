/* kinds/logistic.h -- the worked example of an EXTENSION kind ("Adding a forward model"): the posterior of
 * an AdditiveGaussianNoiseLikelihood (statistics/likelihood.py:49-87) over a logistic-growth ODE solver -- a second
 * SolverInterface (model/interface.py:7-67) next to the reference's LotkaVolterraSolver (test/testSetup.py:61-141) -- and
 * a Gaussian prior.  theta = (log r, log K) (the exp parametrisation of LotkaVolterraParameter, testSetup.py:47-58); one
 * solve per design row x0_n over [0, t_final] by rk4_steps steps of classical RK4; F_n = x(t_final), one output per row.
 *
 * A contributor writes: this header (the forward map of one design row + the description check), one line in
 * ymc_kinds.def, the enum value in include/ymc.h, and on the Python side a solver class + `lowering.register_solver`
 * (yagre-mcmc_amd/kinds/logistic.py).  The checker's counterpart lives in oracle/ymc_oracle.c (test infrastructure). */
#ifndef YMC_KIND_LOGISTIC_H
#define YMC_KIND_LOGISTIC_H

#include "logistic_numerics.h"

template <int D>
struct Density<YMC_DENSITY_POSTERIOR_LOGISTIC, D> {
    static constexpr int kOutputs = 1; /* forward outputs per design row */

    /* SolverInterface.invoke for design row n: F[0] = x(t_final) */
    static __device__ __forceinline__ void forward_row(const DensityDev<D> &p, const double (&th)[D], int n, double (&F)[kOutputs])
    {
        static_assert(D == 2, "logistic growth has two parameters");
        F[0] = ymc_logistic_solve(p.design[n], ymc_exp(th[0]), ymc_exp(th[1]), p.t_final, p.rk4_steps);
    }

    /* UnnormalisedPosterior.evaluate_log (chain/target.py:19-22) */
    static __device__ __forceinline__ double eval(const DensityDev<D> &p, const double (&th)[D], uint32_t &st)
    {
        return ymc_rowwise_posterior<YMC_DENSITY_POSTERIOR_LOGISTIC, D>(p, th, st);
    }
};

/* host: what the descriptor must satisfy beyond the common posterior fields (-> message, or nullptr) */
inline const char *ymc_kind_check_logistic(const ymc_density_desc &in)
{
    if (in.rk4_steps < 1) return "rk4_steps must be positive";
    if (!(in.t_final > 0.0) || !(in.t_final < 1e300)) return "t_final must be positive and finite";
    for (int32_t n = 0; n < in.n_data; ++n)
        if (!(in.design[n] > 0.0)) return "logistic growth: the initial populations (design) must be positive";
    return nullptr;
}

#endif /* YMC_KIND_LOGISTIC_H */
