/* ymc_tile_dev.h -- device helpers shared by the matrix-core kernels (ymc_tile.hip, ymc_tile_aem.hip). */
#ifndef YMC_TILE_DEV_H
#define YMC_TILE_DEV_H

#include "ymc_kernels.h"
#include "ymc_tile.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double shfl_xor_f64(double v, int mask)
{
    uint64_t b = ymc_bits(v);
    uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
    lo = (uint32_t)__shfl_xor((int)lo, mask, 64);
    hi = (uint32_t)__shfl_xor((int)hi, mask, 64);
    return ymc_from_bits(((uint64_t)hi << 32) | lo);
}

/* The 2 x 2 transpose between the lane rows (16 lanes each) g = 0 | 1 and g = 2 | 3: afterwards `a` holds
 * (own a, partner's a) and `b` (partner's b, own b) for (even, odd) rows -- i.e. a's odd rows and b's even
 * rows are swapped.  gfx950's v_permlane16_swap_b32 does exactly this per dword, so a normal pair changes
 * hands in two instructions (it replaced four 64-bit selects and a 64-bit ds_bpermute per pair). */
__device__ __forceinline__ void swap_odd_even_rows(double &a, double &b)
{
    const uint64_t ba = ymc_bits(a), bb = ymc_bits(b);
    const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)ba, (uint32_t)bb, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(ba >> 32), (uint32_t)(bb >> 32), false, false);
    a = ymc_from_bits(((uint64_t)hi[0] << 32) | lo[0]);
    b = ymc_from_bits(((uint64_t)hi[1] << 32) | lo[1]);
}

/* (S0 + S1) + (S2 + S3) over the four lanes (g = 0..3) that share a chain */
__device__ __forceinline__ double combine4(double s)
{
    double v = s + shfl_xor_f64(s, 16);
    return v + shfl_xor_f64(v, 32);
}

}  // namespace

#endif /* YMC_TILE_DEV_H */
