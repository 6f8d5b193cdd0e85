/* ymc_aem.h -- launch interface of the adaptive-error-model kernel (YMC_METHOD_AEM2). */
#ifndef YMC_AEM_H
#define YMC_AEM_H

#include "ymc_device.h"

#define YMC_AEM_CACHE 3 /* AEMCache.CACHESIZE, utility/memoisation.py:84 */

template <int D>
struct AemArgs {
    DensityDev<D> fine, coarse; /* POSTERIOR_LINEAR, data_dim <= YMC_LANE_MMAX */
    double sigma[D];
    uint64_t seed, chain0, K;
    uint64_t t0, wf_n0;
    uint32_t nsteps, flags;
    uint32_t rec_stride, rec_since; /* thinned record (RecordClock) */
    int32_t eq_mode, S, min_data, use_heuristic;
    /* carried per chain */
    double *x, *lp;
    uint64_t *acc_count;
    double *wf_mean, *wf_m2;
    const double *recip;
    uint32_t *status;
    /* the coarse AEMLikelihood's state per chain */
    uint64_t *err_n;                        /* [K]     accumulator nData */
    double *err_mean, *err_m2, *aem_prec;   /* [K][m]  Welford mean / M2, inflated noise precision */
    int32_t *aem_flags;                     /* [K][2]  {precision set, cache fill} */
    double *cache_key, *cache_val;          /* [K][3][D], [K][3]: oldest first */
    /* record rows of this launch or null */
    double *traj;
    uint8_t *acc;
    double *lp_out;
    uint8_t *sub_acc;
};

template <int D>
hipError_t ymc_launch_aem(const AemArgs<D> &a, hipStream_t s);

#endif /* YMC_AEM_H */
