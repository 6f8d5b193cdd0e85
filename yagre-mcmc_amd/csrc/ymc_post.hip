/*
 * ymc_post.hip -- batched post-processing of the recorded trajectories on the
 * device: autocorrelation function and integrated autocorrelation time (IAT)
 * per chain, the step every example script of the reference performs right
 * after run() to pick a thinning stride
 * (yagremcmc/postprocessing/autocorrelation.py:5-140; e.g.
 * example_mcmc_2d_singleLevel.py:71-79).  With K chains the record never has
 * to leave HBM for it.
 *
 *   series_from_record_kernel  record traj[T][K][d] (or the tile kernel's layout) -> series-major s[S][N]
 *                              ('mean': one series per chain, the mean over the
 *                              d components of every state, :126-129;
 *                              'max': one series per chain and component, :131-138)
 *   iat_kernel                 one workgroup per series: centre (:22), lag sums
 *                              c_l = sum_t x_t x_{t+l} and acf_l = c_l / c_0
 *                              (:25-26) computed lag by lag, iatSeq_l =
 *                              2 cumsum(acf)_l - 1 (:85), stopping at Sokal's
 *                              window, the first l with l >= c * iatSeq_l
 *                              (:51-59); IAT = rint(iatSeq at that lag) (:89)
 *   acf_kernel                 all requested lags of the ACF of a series (plots)
 *
 * The reference forms the lag sums with scipy.signal.correlate (FFT for long
 * series); here they are direct sums in a fixed order, so values agree to
 * rounding (~1e-13) and the integer IAT agrees except at exact ties.
 */
#include <cstdlib>

#include "ymc_kernels.h"
#include "ymc_post.h"

namespace {

constexpr int kPostThreads = 256;
constexpr int kLdsDoubles = 19456; /* 152 KB of the 160 KB LDS for the centred series */

__device__ __forceinline__ double block_sum(double v, double *scratch)
{
    /* wave reduction by xor-shuffles, then the 4 wave sums in order */
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t b = ymc_bits(v);
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)b, m, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(b >> 32), m, 64);
        v = v + ymc_from_bits(((uint64_t)hi << 32) | lo);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

/* component j of chain k in row n of the record: plain layout [row][K][d], or the tile kernel's
 * [row][ceil(K/16)][KS][64] with (chain = 16 tile + lane % 16, component = 4 s + lane / 16) */
struct RecordView {
    const double *traj;
    uint64_t K;
    int d, KS; /* KS == 0: plain layout */
    __device__ __forceinline__ double at(uint64_t n, uint64_t k, int j) const
    {
        if (KS == 0) return traj[(n * K + k) * (uint64_t)d + j];
        const uint64_t nct = (K + 15) / 16;
        return traj[(((n * nct + k / 16) * (uint64_t)KS + (uint64_t)(j / 4)) * 64) + (uint64_t)(j % 4) * 16 + k % 16];
    }
};

/* np.mean(seq, axis=1) (autocorrelation.py:126-129) in NumPy's own summation order for a contiguous axis
 * (pairwise_sum): a block of fewer than 8 terms left to right; a block of up to 128 as eight interleaved partial
 * sums combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the remainder added one by one; a longer one split at
 * n/2 rounded down to a multiple of 8, each half summed the same way; then a division by d. */
__device__ __forceinline__ double numpy_block_sum(const RecordView &v, uint64_t n, uint64_t k, int j0, int len)
{
    double res;
    if (len < 8) {
        res = v.at(n, k, j0);
        for (int j = 1; j < len; ++j) res = res + v.at(n, k, j0 + j);
    } else {
        double r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = v.at(n, k, j0 + q);
        int i = 8;
        for (; i < len - (len % 8); i += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = r[q] + v.at(n, k, j0 + i + q);
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < len; ++i) res = res + v.at(n, k, j0 + i);
    }
    return res;
}

__device__ __forceinline__ double numpy_row_mean(const RecordView &v, uint64_t n, uint64_t k)
{
    const int d = v.d;
    double res;
    if (d <= 128) {
        res = numpy_block_sum(v, n, k, 0, d);
    } else if (d <= 256) {
        int n2 = d / 2;
        n2 -= n2 % 8;
        res = numpy_block_sum(v, n, k, 0, n2) + numpy_block_sum(v, n, k, n2, d - n2);
    } else {
        /* the general recursion (beyond 256 components: the wide path's records), unrolled into a loop over frames:
         * phase 0 = not started, 1 = left half running, 2 = right half running */
        struct Frame { int start, len, n2, phase; double left; } fr[24];
        int depth = 0;
        fr[0] = {0, d, 0, 0, 0.0};
        double ret = 0.0;
        while (depth >= 0) {
            Frame &f = fr[depth];
            if (f.phase == 0) {
                if (f.len <= 128) {
                    ret = numpy_block_sum(v, n, k, f.start, f.len);
                    --depth;
                } else {
                    f.n2 = f.len / 2;
                    f.n2 -= f.n2 % 8;
                    f.phase = 1;
                    fr[depth + 1] = {f.start, f.n2, 0, 0, 0.0};
                    ++depth;
                }
            } else if (f.phase == 1) {
                f.left = ret;
                f.phase = 2;
                fr[depth + 1] = {f.start + f.n2, f.len - f.n2, 0, 0, 0.0};
                ++depth;
            } else {
                ret = f.left + ret;
                --depth;
            }
        }
        res = ret;
    }
    return res / (double)d;
}

/* s[c][n], c = chain * per_chain + component, for rows [row0, row0 + N) of the record:
 * a 32 x 32 tile transpose through LDS (coalesced reads along c, coalesced writes along n) */
__global__ __launch_bounds__(256) void series_from_record_kernel(const RecordView v, uint64_t row0, uint64_t N,
                                                                 int mean_mode, double *__restrict__ out)
{
    __shared__ double tile[32][33];
    const int per_chain = mean_mode ? 1 : v.d;
    const uint64_t C = v.K * (uint64_t)per_chain;
    const uint64_t c0 = (uint64_t)blockIdx.x * 32, n0 = (uint64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const uint64_t n = n0 + r, c = c0 + tx;
        double val = 0.0;
        if (n < N && c < C)
            val = mean_mode ? numpy_row_mean(v, row0 + n, c) : v.at(row0 + n, c / (uint64_t)v.d, (int)(c % (uint64_t)v.d));
        tile[r][tx] = val;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const uint64_t c = c0 + r, n = n0 + tx;
        if (n < N && c < C) out[c * N + n] = tile[tx][r];
    }
}

/*
 * IAT of many series at once: ONE WAVE per series, lags in register blocks of 32.
 *
 * iat_kernel (below, kept for the ACF of single long series) gives a series a whole workgroup, keeps it in
 * LDS and forms every lag sum with two LDS reads per product: with 10^5 series of 10^4 states
 * (BASELINE configs[1] right after run()) that is LDS-bandwidth bound and leaves most of the chip idle
 * (measured 51 ms per 65 536 series).  Here the lags l = 32 b .. 32 b + 31 of block b are 32 accumulators
 * per lane; the centred series streams through a small wave-private LDS tile (coalesced global loads), lane
 * i takes 8 consecutive elements a_j of the tile and slides a register window over the elements they pair
 * with, so a product costs one fma and 1/32 LDS read.  After each block the 32 lag sums are reduced over
 * the wave and Sokal's window test (autocorrelation.py:51-59) runs on them; most series close their window
 * inside the third block.  Lag sums are formed in a different order than iat_kernel's: values agree to
 * rounding, the integer IAT except at exact ties (as against the reference's FFT-based sums).
 */
constexpr int kIatLags = 32;   /* lags per register block */
constexpr int kIatPer = 8;     /* consecutive elements per lane and tile */
constexpr int kIatTile = 64 * kIatPer;
constexpr int kIatNear = 96;   /* lag blocks starting at l0 <= kIatNear take both factors from one staged tile */

__host__ __device__ __forceinline__ int iat_pad(int t) { return t + (t >> 3); } /* one spare double per 8: lane bases 9 apart */

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t b = ymc_bits(v);
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)b, m, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(b >> 32), m, 64);
        v = v + ymc_from_bits(((uint64_t)hi << 32) | lo);
    }
    return v;
}

__global__ __launch_bounds__(64) void iat_wave_kernel(const double *__restrict__ series, uint64_t n_series, uint64_t N,
                                                      double sokal, int32_t *__restrict__ iat_out,
                                                      int32_t *__restrict__ lag_out)
{
    constexpr int L = kIatLags, P = kIatPer, TB = kIatTile;
    /* x[T0 .. T0 + TB + kIatNear + L) - mean, 0 past the end: for the first lag blocks (l0 <= kIatNear, where almost
     * every series closes its window) both factors of a product come from this one tile, so a block reads the
     * series once; later blocks stage the shifted factors separately */
    __shared__ double tileA[TB + kIatNear + L + (TB + kIatNear + L) / 8 + 8];
    __shared__ double tileB[TB + L + (TB + L) / 8 + 8];  /* x[T0 + l0 .. T0 + l0 + TB + L) - mean (far blocks only) */
    const uint64_t sidx = blockIdx.x;
    if (sidx >= n_series) return;
    const double *x = series + sidx * N;
    const int lane = threadIdx.x;

    /* the lane's elements in the order t = lane, lane + 64, ...: eight loads in flight at a time (one at a time, each
     * waited for its own trip to HBM: the kernel spent 78 % of its wave cycles parked), summed in the same order */
    double part = 0.0;
    {
        uint64_t t = lane;
        for (; t + 7 * 64 < N; t += 8 * 64) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = x[t + (uint64_t)j * 64];
#pragma unroll
            for (int j = 0; j < 8; ++j) part = part + v[j];
        }
        for (; t < N; t += 64) part = part + x[t];
    }
    const double mean = wave_sum(part) / (double)N;

    double c0 = 0.0, cum = 1.0, iat_at = 1.0;
    int64_t stop = 0;
    bool found = false;
    for (uint64_t l0 = 0; l0 < N && !found; l0 += L) {
        double acc[L];
#pragma unroll
        for (int l = 0; l < L; ++l) acc[l] = 0.0;
        const bool near = l0 <= (uint64_t)kIatNear;
        const double *Bt = near ? tileA : tileB;        /* the shifted factors: Bt[iat_pad(bo + q)] = x[T0 + l0 + q] */
        const int bo = near ? (int)l0 : 0;
        /* a tile's elements: requested (issue) and, centred, written to LDS (commit) separately, so that the NEXT tile's loads
         * are in flight while this one is multiplied (+3.5 %: the kernel re-reads its series once per block of lags and runs
         * at 3.7 TB/s of such reads -- it is the passes that cost, not their latency) */
        constexpr int kStage = (TB + kIatNear + L + 63) / 64;
        auto issue = [&](double (&v)[kStage], uint64_t base, int count) {
#pragma unroll
            for (int j = 0; j < kStage; ++j) {
                const int q = lane + 64 * j;
                const uint64_t t = base + (uint64_t)q;
                v[j] = (q < count && t < N) ? x[t] : mean;      /* past the end: x - mean = 0 */
            }
        };
        auto commit = [&](double *dst, const double (&v)[kStage], int count) {
#pragma unroll
            for (int j = 0; j < kStage; ++j) {
                const int q = lane + 64 * j;
                if (q < count) dst[iat_pad(q)] = v[j] - mean;
            }
        };
        const int na = near ? TB + (int)l0 + L : TB;
        double va[kStage];
        issue(va, 0, na);
        for (uint64_t T0 = 0; T0 + l0 < N; T0 += TB) { /* tiles whose elements still have a partner at lag l0 */
            __syncthreads();
            commit(tileA, va, na);
            if (!near) {
                double vb[kStage];
                issue(vb, T0 + l0, TB + L);
                commit(tileB, vb, TB + L);
            }
            __syncthreads();
            if (T0 + TB + l0 < N) issue(va, T0 + TB, na);
            double a[P], w[L];
#pragma unroll
            for (int j = 0; j < P; ++j) a[j] = tileA[iat_pad(P * lane + j)];
#pragma unroll
            for (int l = 0; l < L; ++l) w[l] = Bt[iat_pad(bo + P * lane + l)];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                /* w[(j + l) % L] holds x[T0 + l0 + P lane + j + l] for l = 0 .. L-1 */
#pragma unroll
                for (int l = 0; l < L; ++l) acc[l] = ymc_fma(a[j], w[(j + l) % L], acc[l]);
                if (j + 1 < P) w[j % L] = Bt[iat_pad(bo + P * lane + j + L)];
            }
        }
        /* reduce the block's lag sums over the wave and run the window test on them */
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const double c = wave_sum(acc[l]);
            const uint64_t lag = l0 + (uint64_t)l;
            if (lag == 0) {
                c0 = c;
            } else if (!found && lag < N) {
                cum = cum + c / c0;
                const double iat_l = 2.0 * cum - 1.0;
                if (!((double)lag < sokal * iat_l)) {
                    found = true;
                    stop = (int64_t)lag;
                    iat_at = iat_l;
                }
            }
        }
    }
    if (lane == 0) {
        const double value = found ? iat_at : 1.0;
        const bool defined = c0 > 0.0 && __builtin_fabs(value) < 2147483647.0;
        iat_out[sidx] = defined ? (int32_t)__builtin_rint(value) : YMC_IAT_UNDEFINED;
        if (lag_out) lag_out[sidx] = (int32_t)stop;
    }
}

/*
 * IAT of many series with every series read from HBM ONCE (round 4): a persistent workgroup of eight waves per CU keeps the
 * centred series of the moment in LDS (padded as above: 88 KB for 9 800 states) and walks the lag blocks over it, the eight
 * waves sharing a block's tiles; the NEXT series' elements are requested into registers before the lag sums start, so the
 * trip to HBM hides behind them.  iat_wave_kernel re-reads its series once per block of 32 lags (4.0 x the algorithmic
 * bytes at BASELINE's shape).  Per block: every wave reduces its 32 partial lag sums over its lanes (a butterfly that halves
 * the values held at every step: v_permlane32_swap / v_permlane16_swap for the upper two), the eight waves' partials meet in
 * LDS, and the window test runs lane-parallel (one division per lane, a prefix sum over the lags).  Sums are formed in
 * another order than iat_wave_kernel's: values agree to rounding, the integer IAT except at exact ties.
 */
constexpr int kResWaves = 8, kResThreads = 64 * kResWaves;
constexpr int kResTail = kIatTile + kIatLags + kIatPer + 8; /* zeros past the end: what a tile that still has partners reads */

__device__ __forceinline__ double join_words(uint32_t lo, uint32_t hi) { return ymc_from_bits(((uint64_t)hi << 32) | lo); }

/* 32 values per lane -> the sum over the wave's 64 lanes of value number lane / 2, in every lane */
__device__ __forceinline__ double iat_butterfly(const double (&acc)[kIatLags], int lane)
{
    double r16[16], r8[8], r4[4], r2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { /* lanes < 32 end up with value i of both halves, lanes >= 32 with value 16 + i */
        const uint64_t a = ymc_bits(acc[i]), b = ymc_bits(acc[16 + i]);
        const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)a, (uint32_t)b, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(a >> 32), (uint32_t)(b >> 32), false, false);
        r16[i] = join_words(lo[0], hi[0]) + join_words(lo[1], hi[1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { /* even rows of 16 lanes: value i, odd rows: value 8 + i */
        const uint64_t a = ymc_bits(r16[i]), b = ymc_bits(r16[8 + i]);
        const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)a, (uint32_t)b, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(a >> 32), (uint32_t)(b >> 32), false, false);
        r8[i] = join_words(lo[0], hi[0]) + join_words(lo[1], hi[1]);
    }
    auto xchg = [&](double keep, double send, int mask) {
        const uint64_t b = ymc_bits(send);
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)b, mask, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(b >> 32), mask, 64);
        return keep + join_words(lo, hi);
    };
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) r4[i] = xchg(b3 ? r8[4 + i] : r8[i], b3 ? r8[i] : r8[4 + i], 8);
#pragma unroll
    for (int i = 0; i < 2; ++i) r2[i] = xchg(b2 ? r4[2 + i] : r4[i], b2 ? r4[i] : r4[2 + i], 4);
    const double r1 = xchg(b1 ? r2[1] : r2[0], b1 ? r2[0] : r2[1], 2);
    return xchg(r1, r1, 1);
}

/* where the series live: element t of series s at base[s * series_step + t * stride]: the series-major scratch (N, 1).
 * Measured and not kept (profiles/r04_iat.txt): 'max' mode straight from the record [row][K][d], whose column k d + j is
 * series k d + j (1, K d), the workgroups of an XCD on neighbouring columns so that they share the 64-byte sectors in L2 --
 * every element then sits on a row of its own a megabyte from the next, and the kernel took 21-23 ms where the transposition
 * (4.9 ms) and the kernel on its output (10.3 ms) take 15 */
struct SeriesSource {
    const double *base;
    uint64_t series_step, stride;
};

template <int NPER>
__global__ __launch_bounds__(kResThreads) void iat_resident_kernel(const SeriesSource src, uint64_t n_series, uint64_t N64,
                                                                   double sokal, int32_t *__restrict__ iat_out,
                                                                   int32_t *__restrict__ lag_out)
{
    constexpr int L = kIatLags, P = kIatPer, TB = kIatTile, W = kResWaves;
    extern __shared__ double xs[];          /* xs[iat_pad(t)] = x[t] - mean, 0 for N <= t < N + kResTail */
    const int N = (int)N64;
    double *part = xs + iat_pad(N + kResTail) + 8; /* [2][W][L]: the waves' partial lag sums of a block, by block parity */
    double *wsum = part + 2 * W * L;               /* [W] */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = N + tid; t < N + kResTail; t += kResThreads) xs[iat_pad(t)] = 0.0;

    double nx[NPER];
#pragma unroll
    for (int j = 0; j < NPER; ++j) nx[j] = 0.0;
    auto request = [&](uint64_t sidx) {
        /* loads under the lanes' t < N and nothing else: a select on the loaded value would wait for it on the spot, and a
         * spilled register anywhere in this kernel would too (scratch reloads count in vmcnt with these) */
        const double *x = src.base + sidx * src.series_step + (uint64_t)tid * src.stride;
#pragma unroll
        for (int j = 0; j < NPER; ++j) {
            if (tid + kResThreads * j < N) nx[j] = *x;
            x += (uint64_t)kResThreads * src.stride;
        }
    };
    const uint64_t G = gridDim.x, first = blockIdx.x;
    uint64_t sidx = first;
    if (sidx < n_series) request(sidx);
    for (; sidx < n_series; sidx += G) {
        double psum = 0.0;
#pragma unroll
        for (int j = 0; j < NPER; ++j) psum = psum + (tid + kResThreads * j < N ? nx[j] : 0.0);
        const double ws = wave_sum(psum);
        if (lane == 0) wsum[wave] = ws;
        __syncthreads(); /* also: everybody has left the previous series */
        const double mean = (((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + ((wsum[4] + wsum[5]) + (wsum[6] + wsum[7]))) / (double)N;
#pragma unroll
        for (int j = 0; j < NPER; ++j) {
            const int t = tid + kResThreads * j;
            if (t < N) xs[iat_pad(t)] = nx[j] - mean;
        }
        __syncthreads();
        if (sidx + G < n_series) request(sidx + G); /* in flight while this series is worked on */

        double c0 = 0.0, cum = 1.0, iat_at = 1.0;
        int stop = 0, par = 0;
        bool found = false;
        for (int l0 = 0; l0 < N && !found; l0 += L, par ^= 1) {
            double acc[L];
#pragma unroll
            for (int l = 0; l < L; ++l) acc[l] = 0.0;
            for (int T0 = wave * TB; T0 + l0 < N; T0 += W * TB) { /* tiles whose elements still have a partner at lag l0 */
                const double *A = xs + iat_pad(T0 + P * lane), *B = xs + iat_pad(T0 + l0 + P * lane);
                double a[P], w[L];
#pragma unroll
                for (int j = 0; j < P; ++j) a[j] = A[j];
#pragma unroll
                for (int l = 0; l < L; ++l) w[l] = B[iat_pad(l)];
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    /* w[(j + l) % L] holds x[T0 + l0 + P lane + j + l] for l = 0 .. L-1 */
#pragma unroll
                    for (int l = 0; l < L; ++l) acc[l] = ymc_fma(a[j], w[(j + l) % L], acc[l]);
                    if (j + 1 < P) w[j % L] = B[iat_pad(j + L)];
                }
            }
            const double mine = iat_butterfly(acc, lane);
            if ((lane & 1) == 0) part[(par * W + wave) * L + (lane >> 1)] = mine;
            __syncthreads();
            /* the window test (autocorrelation.py:51-59, 85-89), lag l0 + (lane mod 32) in every lane */
            const int li = lane & 31, lag = l0 + li;
            const double *pp = part + par * W * L + li;
            const double c = ((pp[0] + pp[L]) + (pp[2 * L] + pp[3 * L])) + ((pp[4 * L] + pp[5 * L]) + (pp[6 * L] + pp[7 * L]));
            if (l0 == 0) c0 = join_words((uint32_t)__shfl((int)(uint32_t)ymc_bits(c), 0, 64),
                                         (uint32_t)__shfl((int)(uint32_t)(ymc_bits(c) >> 32), 0, 64));
            const bool counts = lag >= 1 && lag < N;
            double pre = counts ? c / c0 : 0.0;
#pragma unroll
            for (int sft = 1; sft < 32; sft <<= 1) { /* inclusive prefix sum over the 32 lags */
                const uint64_t b = ymc_bits(pre);
                const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)b, sft, 32);
                const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(b >> 32), sft, 32);
                if (li >= sft) pre = pre + join_words(lo, hi);
            }
            const double cum_l = cum + pre, iat_l = 2.0 * cum_l - 1.0;
            const bool closes = counts && !((double)lag < sokal * iat_l);
            const uint32_t mask = (uint32_t)__builtin_amdgcn_ballot_w64(closes);
            const int src = mask ? __builtin_ctz(mask) : 31;
            const uint64_t pick = ymc_bits(mask ? iat_l : cum_l);
            const double got = join_words((uint32_t)__shfl((int)(uint32_t)pick, src, 64),
                                          (uint32_t)__shfl((int)(uint32_t)(pick >> 32), src, 64));
            if (mask) {
                found = true;
                stop = l0 + src;
                iat_at = got;
            } else {
                cum = got;
            }
        }
        if (tid == 0) {
            const double value = found ? iat_at : 1.0;
            const bool defined = c0 > 0.0 && __builtin_fabs(value) < 2147483647.0;
            iat_out[sidx] = defined ? (int32_t)__builtin_rint(value) : YMC_IAT_UNDEFINED;
            if (lag_out) lag_out[sidx] = (int32_t)stop;
        }
    }
}

__global__ __launch_bounds__(kPostThreads) void iat_kernel(const double *__restrict__ series, uint64_t N, double sokal,
                                                           int32_t *__restrict__ iat_out, int32_t *__restrict__ lag_out)
{
    extern __shared__ double lds[];
    __shared__ double scratch[4];
    const double *x = series + (uint64_t)blockIdx.x * N;
    const bool in_lds = N <= (uint64_t)kLdsDoubles;
    const int tid = threadIdx.x;

    double part = 0.0;
    for (uint64_t t = tid; t < N; t += kPostThreads) part = part + x[t];
    const double mean = block_sum(part, scratch) / (double)N;
    if (in_lds)
        for (uint64_t t = tid; t < N; t += kPostThreads) lds[t] = x[t] - mean;
    __syncthreads();

    auto at = [&](uint64_t t) { return in_lds ? lds[t] : x[t] - mean; };
    part = 0.0;
    for (uint64_t t = tid; t < N; t += kPostThreads) {
        const double v = at(t);
        part = ymc_fma(v, v, part);
    }
    const double c0 = block_sum(part, scratch);

    /* iatSeq_0 = 2 * acf_0 - 1 = 1; scan the lags until the heuristic fails */
    double cum = 1.0;
    double iat_at = 1.0;
    int64_t stop = 0; /* np.argmin of an all-True mask */
    bool found = false; /* lag 0 always satisfies 0 < c * 1 (c > 0 is checked on the host) */
    for (uint64_t l = 1; l < N && !found; ++l) {
        part = 0.0;
        for (uint64_t t = tid; t + l < N; t += kPostThreads) part = ymc_fma(at(t), at(t + l), part);
        const double acf = block_sum(part, scratch) / c0;
        cum = cum + acf;
        const double iat_l = 2.0 * cum - 1.0;
        if (!((double)l < sokal * iat_l)) {
            found = true;
            stop = (int64_t)l;
            iat_at = iat_l;
        }
    }
    if (tid == 0) {
        /* a constant series (a chain that never moved, or one flagged NONFINITE) has c0 = 0: its
         * autocorrelation is 0 / 0 and the reference fails on it; report YMC_IAT_UNDEFINED instead of
         * converting a NaN to an integer */
        const double value = found ? iat_at : 1.0;
        const bool defined = c0 > 0.0 && __builtin_fabs(value) < 2147483647.0;
        iat_out[blockIdx.x] = defined ? (int32_t)__builtin_rint(value) : YMC_IAT_UNDEFINED;
        if (lag_out) lag_out[blockIdx.x] = (int32_t)stop;
    }
}

__global__ __launch_bounds__(kPostThreads) void acf_kernel(const double *__restrict__ series, uint64_t N, uint64_t L,
                                                           double *__restrict__ acf_out)
{
    extern __shared__ double lds[];
    __shared__ double scratch[4];
    const double *x = series + (uint64_t)blockIdx.x * N;
    const bool in_lds = N <= (uint64_t)kLdsDoubles;
    const int tid = threadIdx.x;
    double part = 0.0;
    for (uint64_t t = tid; t < N; t += kPostThreads) part = part + x[t];
    const double mean = block_sum(part, scratch) / (double)N;
    if (in_lds)
        for (uint64_t t = tid; t < N; t += kPostThreads) lds[t] = x[t] - mean;
    __syncthreads();
    auto at = [&](uint64_t t) { return in_lds ? lds[t] : x[t] - mean; };
    double c0 = 1.0;
    for (uint64_t l = 0; l < L && l < N; ++l) {
        part = 0.0;
        for (uint64_t t = tid; t + l < N; t += kPostThreads) part = ymc_fma(at(t), at(t + l), part);
        const double c = block_sum(part, scratch);
        if (l == 0) c0 = c;
        if (tid == 0) acf_out[(uint64_t)blockIdx.x * L + l] = c / c0;
    }
}

size_t lds_for(uint64_t N) { return N <= (uint64_t)kLdsDoubles ? (size_t)N * sizeof(double) : 0; }

}  // namespace

hipError_t ymc_launch_series_from_record(const double *traj, uint64_t row0, uint64_t N, uint64_t K, int d, int tile_KS,
                                         int mean_mode, double *out, hipStream_t s)
{
    const uint64_t C = K * (uint64_t)(mean_mode ? 1 : d);
    RecordView v{traj, K, d, tile_KS};
    hipLaunchKernelGGL(series_from_record_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((N + 31) / 32)), dim3(256), 0,
                       s, v, row0, N, mean_mode, out);
    return hipGetLastError();
}

/* LDS of iat_resident_kernel for series of N states, or 0 if they do not fit */
static size_t iat_resident_lds(uint64_t N)
{
    if (N > 20 * (uint64_t)kResThreads) return 0; /* the next series is held in 20 registers per thread while this one is worked on */
    const size_t doubles = (size_t)iat_pad((int)N + kResTail) + 8 + 2 * kResWaves * kIatLags + kResWaves;
    const size_t bytes = doubles * sizeof(double);
    return bytes <= 160 * 1024 ? bytes : 0;
}

static hipError_t launch_resident(const SeriesSource &src, uint64_t S, uint64_t N, size_t lds, double sokal, int32_t *iat,
                                  int32_t *lag, hipStream_t s)
{
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(S < (uint64_t)cus ? S : (uint64_t)cus);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&iat_resident_kernel<20>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(iat_resident_kernel<20>, dim3(grid), dim3(kResThreads), lds, s, src, S, N, sokal, iat, lag);
    return hipGetLastError();
}

hipError_t ymc_launch_iat(const double *series, uint64_t S, uint64_t N, double sokal, int32_t *iat, int32_t *lag,
                          hipStream_t s)
{
    /* many series: a persistent workgroup per CU with the series of the moment in LDS (read from HBM once), or -- series
     * that do not fit the LDS -- a wave each (register-blocked lags, the series re-read per block of lags); a few long
     * ones: a workgroup each with the series in LDS.  YMC_IAT_KERNEL=resident|wave|block overrides (tests run all) */
    const size_t res_lds = iat_resident_lds(N);
    char pick = S >= 64 ? (res_lds ? 'r' : 'w') : 'b';
    if (const char *env = std::getenv("YMC_IAT_KERNEL")) pick = env[0];
    if (pick == 'r' && !res_lds) pick = 'w';
    if (pick == 'r') return launch_resident(SeriesSource{series, N, 1}, S, N, res_lds, sokal, iat, lag, s);
    if (pick == 'w') {
        hipLaunchKernelGGL(iat_wave_kernel, dim3((unsigned)S), dim3(64), 0, s, series, S, N, sokal, iat, lag);
        return hipGetLastError();
    }
    const size_t lds = lds_for(N);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&iat_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsDoubles * sizeof(double)));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(iat_kernel, dim3((unsigned)S), dim3(kPostThreads), lds, s, series, N, sokal, iat, lag);
    return hipGetLastError();
}

hipError_t ymc_launch_acf(const double *series, uint64_t S, uint64_t N, uint64_t L, double *acf, hipStream_t s)
{
    const size_t lds = lds_for(N);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&acf_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsDoubles * sizeof(double)));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(acf_kernel, dim3((unsigned)S), dim3(kPostThreads), lds, s, series, N, L, acf);
    return hipGetLastError();
}
