/* ymc_kernels.h -- launch interface between the C ABI (ymc_api.hip) and the
 * gfx950 kernels (ymc_kernels.hip). */
#ifndef YMC_KERNELS_H
#define YMC_KERNELS_H

#include "ymc_device.h"

/* Everything one launch of a lane-per-chain kernel needs, by value. */
template <int D>
struct ChainArgs {
    DensityDev<D> tgt;   /* the (fine) target */
    DensityDev<D> sur;   /* MLDA2: the coarse density */
    double sigma[D];     /* proposal standard deviations (pCN: of the prior) */
    double pcn_c1, pcn_c2; /* pCN: sqrt(1 - 2h), sqrt(2h) */
    uint64_t seed, chain0, K;
    uint64_t t0;         /* first transition (MLDA2: first fine step) of this launch */
    uint64_t wf_n0;      /* Welford count before this launch */
    uint32_t nsteps, flags;
    int32_t eq_mode, S;
    int32_t lv_rows;     /* Lotka-Volterra: spread a chain over n_data lanes (a speed choice, same results) */
    /* ... and cut the launch into lv_pieces equal pieces of lv_piece_len transitions handed from wave to wave
     * (mrw_lv_rows_balanced_kernel; 0 = one wave per chain wave); handoff[chain wave] = transitions published */
    int32_t lv_pieces;
    uint32_t lv_piece_len;
    uint32_t *handoff;
    uint32_t lv_poll_limit; /* polls of a predecessor's publication before a wave gives up (YMC_CHAIN_HANDOFF) */
    uint32_t *fault;        /* the run's count of abandoned hand-overs: checked by ymc_run_sync and the fetch calls */
    int32_t split;       /* random walk, at most one wave of chains per SIMD: draws made by helper waves (same results) */
    uint32_t rec_stride, rec_since; /* thinned record (RecordClock): stride, transitions since the last recorded one */
    /* carried per-chain state */
    double *x, *lp, *lpc;
    uint64_t *acc_count;
    double *wf_mean, *wf_m2;
    const double *chol;  /* dense proposal covariance: lower Cholesky factor [D][D] on the device, else null */
    const double *recip; /* recip[n] = RN(1 / (wf_n0 + n + 1)) for the transitions of this launch (moments only) */
    uint32_t *status;
    /* record rows of this launch (already offset to the write position), or null */
    double *traj;
    uint8_t *acc;
    double *lp_out;
    uint8_t *sub_acc;
};

template <int D>
hipError_t ymc_launch_chain(int kind, int method, const ChainArgs<D> &a, hipStream_t s);
/* ymc_mlda2.hip: two-level kernels with a compile-time sub-chain length; false = not covered */
template <int D>
bool ymc_launch_mlda2_static_full(int kind, unsigned rec, const ChainArgs<D> &a, hipStream_t s); /* full-rate record masks */
template <int D>
bool ymc_launch_mlda2_static_thin(int kind, unsigned rec, const ChainArgs<D> &a, hipStream_t s); /* thinned record masks */
template <int D>
inline bool ymc_launch_mlda2_static(int kind, unsigned rec, const ChainArgs<D> &a, hipStream_t s)
{
    return a.rec_stride > 1 ? ymc_launch_mlda2_static_thin<D>(kind, rec, a, s) : ymc_launch_mlda2_static_full<D>(kind, rec, a, s);
}
template <int D>
hipError_t ymc_launch_logdensity(int kind, const DensityDev<D> &p, uint64_t n, const double *x, double *out,
                                 uint32_t *status, hipStream_t s);
template <int D>
hipError_t ymc_launch_forward(int kind, const DensityDev<D> &p, uint64_t n, const double *x, double *out,
                              hipStream_t s);
hipError_t ymc_launch_recip(uint64_t n0, uint64_t count, double *out, hipStream_t s);
hipError_t ymc_launch_rng_fill(uint64_t seed, uint64_t chain0, uint64_t K, uint64_t t0, uint64_t nt, int d,
                               double *xi, double *u, hipStream_t s);
hipError_t ymc_launch_numerics(int fn, uint64_t n, const double *x, double *y0, double *y1, hipStream_t s);
hipError_t ymc_launch_clock_probe(uint64_t ticks_100mhz, uint64_t *out, hipStream_t s);
hipError_t ymc_launch_pack_summary(uint64_t K, int d, uint64_t wf_n, const double *x, const double *lp,
                                   const uint64_t *accepted, const double *wmean, const double *wm2,
                                   const uint32_t *status, uint64_t *out, hipStream_t s);

#endif /* YMC_KERNELS_H */
