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

/* np.mean(seq, axis=1) (autocorrelation.py:126-129) in NumPy's own summation order for a contiguous
 * axis of length d <= 128 (pairwise_sum's unrolled block): fewer than 8 terms left to right; else
 * eight interleaved partial sums combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the remainder added
 * one by one; then a division by d. */
__device__ __forceinline__ double numpy_row_mean(const RecordView &v, uint64_t n, uint64_t k)
{
    const int d = v.d;
    double res;
    if (d < 8) {
        res = v.at(n, k, 0);
        for (int j = 1; j < d; ++j) res = res + v.at(n, k, j);
    } else {
        double r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = v.at(n, k, q);
        int i = 8;
        for (; i < d - (d % 8); i += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = r[q] + v.at(n, k, i + q);
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < d; ++i) res = res + v.at(n, k, i);
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
        iat_out[blockIdx.x] = defined ? (int32_t)__builtin_rint(value) : -1;
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

hipError_t ymc_launch_iat(const double *series, uint64_t S, uint64_t N, double sokal, int32_t *iat, int32_t *lag,
                          hipStream_t s)
{
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
