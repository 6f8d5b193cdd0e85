/* ymc_nested.h -- launch interface of the nested delayed-acceptance kernel
 * (YMC_METHOD_MLDA_NESTED, include/ymc.h). */
#ifndef YMC_NESTED_H
#define YMC_NESTED_H

#include "ymc_device.h"

template <int D>
struct NestedArgs {
    /* device arrays of E lowered densities each: P[e], Q[e] of level e (Q[0] is not used);
     * read through uniform loads, they do not fit the kernel-argument segment for E = 4, d = 8 */
    const DensityDev<D> *P, *Q;
    int32_t steps[YMC_MAX_LEVELS]; /* steps of one run of level e, e < E-1 */
    int32_t E, eq_mode;
    double sigma[D];
    uint64_t seed, chain0, K;
    uint64_t t0;        /* first top-level step of this launch */
    uint64_t cost_top;  /* transitions one top-level step spans (c_{E-1}) */
    uint64_t wf_n0;
    uint32_t nsteps, flags;
    uint32_t rec_stride, rec_since; /* thinned record (RecordClock) */
    /* carried per-chain state: x, P_top(x), Q_top(x) */
    double *x, *lp, *lpq;
    uint64_t *acc_count;
    uint64_t *level_acc; /* [E-1][K] accepted transitions of the lower levels */
    double *wf_mean, *wf_m2;
    const double *recip;
    uint32_t *status;
    double *traj;
    uint8_t *acc;
    double *lp_out;
};

/* nests of up to four effective levels / of more (the same file built with -DYMC_NESTED_DEEP) */
template <int D>
hipError_t ymc_launch_nested(int kind, const NestedArgs<D> &a, hipStream_t s);
template <int D>
hipError_t ymc_launch_nested_deep(int kind, const NestedArgs<D> &a, hipStream_t s);

#endif /* YMC_NESTED_H */
