/* kinds/logistic_numerics.h -- the arithmetic of the worked extension kind (logistic growth), shared by the device
 * (kinds/logistic.h) and the CPU checker (oracle/ymc_oracle.c): plain C, every fused multiply-add explicit, compiled
 * with -ffp-contract=off on both sides, so that the two agree bit for bit (the rule of ymc_numerics.h).
 *
 *     dx/dt = r x (1 - x / K),   x(0) = x0,   theta = (log r, log K)
 *
 * one step of classical RK4 with f(x) = (r x) (1 - x invK), invK = RN(1 / K) computed once per evaluation. */
#ifndef YMC_KIND_LOGISTIC_NUMERICS_H
#define YMC_KIND_LOGISTIC_NUMERICS_H

#include "../ymc_numerics.h"

YMC_FN double ymc_logistic_rate(double x, double r, double inv_k)
{
    return (r * x) * ymc_fma(-x, inv_k, 1.0);
}

YMC_FN double ymc_logistic_rk4_step(double x, double r, double inv_k, double h, double h2, double h6)
{
    const double k1 = ymc_logistic_rate(x, r, inv_k);
    const double k2 = ymc_logistic_rate(ymc_fma(h2, k1, x), r, inv_k);
    const double k3 = ymc_logistic_rate(ymc_fma(h2, k2, x), r, inv_k);
    const double k4 = ymc_logistic_rate(ymc_fma(h, k3, x), r, inv_k);
    return ymc_fma(h6, (k1 + k4) + 2.0 * (k2 + k3), x);
}

/* the population at t_final from x0: `steps` equal RK4 steps */
YMC_FN double ymc_logistic_solve(double x0, double r, double k, double t_final, int32_t steps)
{
    const double h = t_final / (double)steps, h2 = 0.5 * h, h6 = h / 6.0, inv_k = 1.0 / k;
    double x = x0;
    for (int32_t s = 0; s < steps; ++s) x = ymc_logistic_rk4_step(x, r, inv_k, h, h2, h6);
    return x;
}

#endif /* YMC_KIND_LOGISTIC_NUMERICS_H */
