/* ymc_post.h -- launch interface of the post-processing kernels (ymc_post.hip). */
#ifndef YMC_POST_H
#define YMC_POST_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* tile_KS: 0 for the plain record layout [row][K][d], else the KS of the tile kernel's layout */
hipError_t ymc_launch_series_from_record(const double *traj, uint64_t row0, uint64_t N, uint64_t K, int d, int tile_KS,
                                         int mean_mode, double *out, hipStream_t s);
hipError_t ymc_launch_iat(const double *series, uint64_t S, uint64_t N, double sokal, int32_t *iat, int32_t *lag,
                          hipStream_t s);
hipError_t ymc_launch_acf(const double *series, uint64_t S, uint64_t N, uint64_t L, double *acf, hipStream_t s);

#endif /* YMC_POST_H */
