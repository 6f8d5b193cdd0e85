/* ymc_tile_dev.h -- device helpers shared by the matrix-core kernels (ymc_tile.hip, ymc_tile_aem.hip,
 * ymc_tile_nested.hip). */
#ifndef YMC_TILE_DEV_H
#define YMC_TILE_DEV_H

#include "ymc_kernels.h"
#include "ymc_tile.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

#ifdef YMC_TILE_PROFILE
/* Development build only (make prof -> libymc_prof.so, tools/tile_sections.py): cycles per section of the
 * instrumented kernel (the two-level one in ymc_tile.hip, or with -DYMC_TILE_PROFILE_AEM the adaptive error model's) as seen
 * by wave 0 of workgroup 0 (s_memtime; every stamp drains the wave's LDS queue, so
 * the sections add up to slightly more than the uninstrumented loop). */
#ifndef YMC_TILE_PROF_EXTERN
__device__ unsigned long long ymc_tile_prof[16];
#endif
#define YMC_PROF_DECL unsigned long long prof_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_ = 0
#define YMC_PROF_START() stamp_ = __builtin_amdgcn_s_memtime()
#define YMC_PROF(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof_[i] += now_ - stamp_; stamp_ = now_; } while (0)
#define YMC_PROF_FLUSH() do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 10; ++i_) ymc_tile_prof[i_] = prof_[i_]; } while (0)
#else
#define YMC_PROF_DECL
#define YMC_PROF_START()
#define YMC_PROF(i)
#define YMC_PROF_FLUSH()
#endif


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

/* A workgroup copies `n2` 16-byte words of a problem block from global memory into LDS, every thread's loads eight at a
 * time: one load per loop iteration, each waited for before its LDS write, left the staging of a 128 KB matrix at
 * sixteen L2 round trips in a row per workgroup. */
template <int THREADS>
__device__ __forceinline__ void ymc_stage_block(double *lds_dst, const double *__restrict__ src, size_t n2, int tid)
{
    double2 *dst = reinterpret_cast<double2 *>(lds_dst);
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    size_t i = tid;
    for (; i + 7 * (size_t)THREADS < n2; i += 8 * (size_t)THREADS) {
        double2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = s2[i + (size_t)j * THREADS];
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[i + (size_t)j * THREADS] = v[j];
    }
    for (; i < n2; i += THREADS) dst[i] = s2[i];
}

/* The pieces of a linear-model posterior in the kernels' operand layout: lane (g, c) of a wave holds component
 * 4 s + g of chain c in register s (the MFMA B operand) and, after a pass over output tiles, rows 4 r + g of every
 * tile in accumulator entry r.  Sums over a chain's outputs / components are the canonical four interleaved
 * partial sums (one per g) joined by combine4 -- the order of ymc_linear_data_summary / ymo_diag_norm_sq. */

/* adds the terms of output tiles mt, mt + 1 (accumulators acc0, acc1 of the forward map without its shift) to this
 * lane's partial sum of |F - Dbar|^2 in the precision n P (likelihood.py:74-84 through the row-mean identity) */
__device__ __forceinline__ void ymc_tile_add_residuals(int mt, const d4 &acc0, const d4 &acc1, const double *bias,
                                                       const double *nprec, const double *dbar, int g, double &Ssum)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * (mt + h) + 4 * r + g;
            const double F = (h ? acc1[r] : acc0[r]) + bias[m];
            const double res = F - dbar[m];
            const double px = nprec[m] * res;
            Ssum = Ssum + res * px;
        }
    }
}

/* this lane's partial sum for the forward map A v + b with A's fragments [MT][KS][64] in LDS or in global memory
 * (every wave reading its own): two accumulator chains of KS MFMAs per pair of output tiles.
 * The compiler requests each fragment one or two MFMAs before its use; requesting them a block ahead (stages with
 * scheduling barriers) bought nothing at one wave per SIMD and cost 19 % at two (profiles/r03_tile_experiments.txt):
 * the matrix pipe takes 66-68 cycles per f64 MFMA in every order (tools/probe_mfma_rate.hip). */
template <int KS>
__device__ __forceinline__ double ymc_tile_residual_sum(const double *A, int MT, const double *bias, const double *nprec,
                                                        const double *dbar, int lane, int g, const double (&v)[KS])
{
    double Ssum = 0.0;
#pragma unroll 1
    for (int mt = 0; mt < MT; mt += 2) {
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const double *A0 = A + (size_t)mt * KS * 64 + lane;
        const double *A1 = A0 + (size_t)KS * 64;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A0[s * 64], v[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1[s * 64], v[s], acc1, 0, 0, 0);
        }
        ymc_tile_add_residuals(mt, acc0, acc1, bias, nprec, dbar, g, Ssum);
    }
    return Ssum;
}

/* Gaussian prior with a diagonal covariance (gaussian.py:19-24): -0.5 |v - mean|^2 in the precision */
template <int KS>
__device__ __forceinline__ double ymc_tile_logprior(const double *pmean, const double *pprec, int g, const double (&v)[KS])
{
    double P = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int j = 4 * s + g;
        const double dx = v[s] - pmean[j];
        const double px = pprec[j] * dx;
        P = P + dx * px;
    }
    return -0.5 * combine4(P);
}

/* `proposal == state` of metropolisHastings.py:60-61 for the 16 chains of a wave: true where any component differs */
template <int KS>
__device__ __forceinline__ bool ymc_tile_any_differs(const double (&p)[KS], const double (&q)[KS])
{
    bool dl = false;
#pragma unroll
    for (int s = 0; s < KS; ++s) dl = dl || !(p[s] == q[s]);
    int mv = dl ? 1 : 0;
    mv |= __shfl_xor(mv, 16, 64);
    mv |= __shfl_xor(mv, 32, 64);
    return mv != 0;
}

/* A matrix that does not live in LDS (a fine or upper-level forward map, a dense proposal's factor), streamed
 * from L2 ONCE PER WORKGROUP through a double buffer in LDS -- `stage`, 2 x ymc_stream_unit<KS, DIV>() doubles --
 * instead of once per wave: with one or two waves per SIMD a wave that reads its own MFMA fragments from L2 keeps
 * too few loads in flight (measured 5.7 x the cost of an LDS-fed GEMV).  A is in fragment order [MT][KS][64];
 * unit (pair, q) = k-steps [q KS/DIV, (q + 1) KS/DIV) of output tiles 2 pair and 2 pair + 1.  Every thread
 * fetches its share of the next unit into registers while the MFMAs of the current one run; one barrier per unit
 * (it also guarantees that everyone has left the buffer that is written next).  emit(pair, acc0, acc1) receives
 * the accumulators of the two tiles; per accumulator the MFMA order is k ascending, as in a loop over LDS- or
 * L2-resident fragments: the same bits.
 * EVERY wave of the workgroup must call this at the same point of its loop (the callers keep waves past the last
 * tile alive for it); `v` is the wave's B operand (component 4 s + g of chain c in register s of lane (g, c)).
 * Back-to-back calls: a call starts writing buffer 0 with no barrier of its own, which is safe only if the
 * previous call's LAST unit was read from buffer 1, i.e. if it staged an even number of units.  DIV is even
 * (asserted), so n_pairs * DIV always is.
 * Measured alternatives, none faster (profiles/r03_tile_experiments.txt): writing unit u + 1 before the MFMAs of unit
 * u with the barrier at the end of the unit (the LDS write latency off the critical path: +4...15 % time), reading
 * all fragments of a unit before its first MFMA. */
template <int KS, int DIV>
constexpr int ymc_stream_unit() { return 2 * (KS / DIV) * 64; }

template <int KS, int DIV, int THREADS, typename Emit>
__device__ __forceinline__ void ymc_stream_tile_pairs(const double *__restrict__ A, int n_pairs, double *stage, int tid,
                                                      int lane, const double (&v)[KS], Emit &&emit)
{
    static_assert(KS % DIV == 0, "staged units are whole k-steps");
    static_assert(DIV % 2 == 0, "an even number of units per call: the next call's first write goes to buffer 0 unguarded");
    constexpr int kSteps = KS / DIV, kUnit = ymc_stream_unit<KS, DIV>(), kPer = (kUnit + THREADS - 1) / THREADS;
    auto fetch = [&](int pair, int q, double (&r)[kPer]) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int idx = tid + j * THREADS;
            if (kUnit % THREADS == 0 || idx < kUnit) {
                const int blk = idx / (kSteps * 64), off = idx % (kSteps * 64);
                r[j] = A[((size_t)(2 * pair + blk) * KS + (size_t)q * kSteps) * 64 + off];
            }
        }
    };
    double r[kPer];
    fetch(0, 0, r);
    int buf = 0;
#pragma unroll 1
    for (int pair = 0; pair < n_pairs; ++pair) {
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < DIV; ++q) {
            double *stg = stage + buf * kUnit;
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const int idx = tid + j * THREADS;
                if (kUnit % THREADS == 0 || idx < kUnit) stg[idx] = r[j];
            }
            __syncthreads();
            if (q + 1 < DIV) fetch(pair, q + 1, r);
            else if (pair + 1 < n_pairs) fetch(pair + 1, 0, r);
#pragma unroll
            for (int sl = 0; sl < kSteps; ++sl) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(stg[sl * 64 + lane], v[q * kSteps + sl], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(stg[(kSteps + sl) * 64 + lane], v[q * kSteps + sl], acc1, 0, 0, 0);
            }
            buf ^= 1;
        }
        emit(pair, acc0, acc1);
    }
}

/* The same stream with every unit requested TWO units ahead (two register sets): with units of KS/8 k-steps the MFMAs of
 * one unit (8 of them, ~530 cycles) are shorter than a trip to L2, so a request made one unit ahead is still in flight
 * when its LDS write comes up and every unit stalls on it.  Same units, same buffers, same barrier, same MFMA order. */
template <int KS, int DIV, int THREADS, typename Emit>
__device__ __forceinline__ void ymc_stream_tile_pairs_ahead2(const double *__restrict__ A, int n_pairs, double *stage, int tid,
                                                             int lane, const double (&v)[KS], Emit &&emit)
{
    static_assert(KS % DIV == 0, "staged units are whole k-steps");
    static_assert(DIV % 2 == 0, "an even number of units per call (buffer and register-set parity)");
    constexpr int kSteps = KS / DIV, kUnit = ymc_stream_unit<KS, DIV>(), kPer = (kUnit + THREADS - 1) / THREADS;
    auto fetch = [&](int pair, int q, double (&r)[kPer]) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int idx = tid + j * THREADS;
            if (kUnit % THREADS == 0 || idx < kUnit) {
                const int blk = idx / (kSteps * 64), off = idx % (kSteps * 64);
                r[j] = A[((size_t)(2 * pair + blk) * KS + (size_t)q * kSteps) * 64 + off];
            }
        }
    };
    double r0[kPer], r1[kPer];
    fetch(0, 0, r0);
    if (DIV > 1) fetch(0, 1, r1);
#pragma unroll 1
    for (int pair = 0; pair < n_pairs; ++pair) {
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < DIV; ++q) {
            double *stg = stage + (q & 1) * kUnit;
            double (&r)[kPer] = (q & 1) ? r1 : r0;
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const int idx = tid + j * THREADS;
                if (kUnit % THREADS == 0 || idx < kUnit) stg[idx] = r[j];
            }
            __syncthreads();
            if (q + 2 < DIV) fetch(pair, q + 2, r);
            else if (pair + 1 < n_pairs) fetch(pair + 1, q + 2 - DIV, r);
#pragma unroll
            for (int sl = 0; sl < kSteps; ++sl) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(stg[sl * 64 + lane], v[q * kSteps + sl], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(stg[(kSteps + sl) * 64 + lane], v[q * kSteps + sl], acc1, 0, 0, 0);
            }
        }
        emit(pair, acc0, acc1);
    }
}

}  // namespace

#endif /* YMC_TILE_DEV_H */
