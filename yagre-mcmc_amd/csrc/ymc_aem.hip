/*
 * ymc_aem.hip -- two-level delayed acceptance with the ADAPTIVE ERROR MODEL
 * (chain/method/aem.py:6-58; statistics/likelihood.py:90-155; noise.py:25-61;
 * utility/memoisation.py:76-149), one surrogate, linear forward maps.
 *
 * One lane owns one chain and, with it, the state the reference keeps inside
 * the coarse AEMLikelihood object of that chain:
 *   - the Welford accumulator of the model error F_fine(y) - F_coarse(y) over
 *     the ACCEPTED fine proposals (aem.py:44-56);
 *   - the residual shift (nData >= minDataSize, likelihood.py:128-133) and the
 *     inflated noise precision (nData > minDataSize, :142-150, noise.py:41-54);
 *   - the three-entry AEMCache of log-likelihood values, most recently used
 *     last.  A hit returns the value stored when the entry was made -- possibly
 *     under an older error model.  The reference's chains depend on that, so
 *     the cache is reproduced entry by entry (slots are named registers; all
 *     indexing below is static).
 * The fine likelihood never adapts (its accumulator is never fed), so the fine
 * log-posterior of the carried state is simply carried.
 */
#define YMC_COEFF_HOISTED 1
#include "ymc_aem.h"

namespace {

constexpr int kBlock = 256;
constexpr int M = YMC_LANE_MMAX;

template <int D>
struct CoarseLikelihood {
    const DensityDev<D> &p;
    int32_t min_data, use_heuristic, eq_mode;
    uint64_t n;
    double mean[M], m2[M], prec[M];
    bool has_prec;
    int nkeys;
    double key[YMC_AEM_CACHE][D];
    double val[YMC_AEM_CACHE];

    __device__ __forceinline__ explicit CoarseLikelihood(const DensityDev<D> &coarse) : p(coarse) {}

    /* AEMLikelihood.compute_log_likelihood under the CURRENT error model (likelihood.py:77-84,128-133) */
    __device__ __forceinline__ double now(const double (&theta)[D]) const
    {
        double F[M], pe[M];
        Density<YMC_DENSITY_POSTERIOR_LINEAR, D>::forward(p, theta, F);
        const bool shift = n >= (uint64_t)min_data;
#pragma unroll
        for (int j = 0; j < M; ++j) pe[j] = (j < p.data_dim) ? (has_prec ? prec[j] : p.noise_prec[j]) : 0.0;
        double total = 0.0;
        for (int i = 0; i < p.n_data; ++i) {
            double r[M];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                r[j] = 0.0;
                if (j < p.data_dim) {
                    r[j] = F[j] - p.data[i * p.data_dim + j];
                    if (shift) r[j] = r[j] + mean[j];
                }
            }
            total = total + ymc_diag_norm_sq<M>(r, pe, p.data_dim);
        }
        return -0.5 * total;
    }

    __device__ __forceinline__ void put(int slot, const double (&theta)[D], double v)
    {
#pragma unroll
        for (int j = 0; j < D; ++j) key[slot][j] = theta[j];
        val[slot] = v;
    }
    __device__ __forceinline__ void copy(int dst, int src)
    {
#pragma unroll
        for (int j = 0; j < D; ++j) key[dst][j] = key[src][j];
        val[dst] = val[src];
    }

    /* AEMLikelihood.query_log_likelihood + update_cache (likelihood.py:121-126,135-141) */
    __device__ __forceinline__ double query(const double (&theta)[D])
    {
        const bool h0 = nkeys > 0 && ymc_states_equal<D>(key[0], theta, eq_mode);
        const bool h1 = nkeys > 1 && ymc_states_equal<D>(key[1], theta, eq_mode);
        const bool h2 = nkeys > 2 && ymc_states_equal<D>(key[2], theta, eq_mode);
        if (h0) { /* retrieve: the entry moves to the back (memoisation.py:98-103) */
            const double v = val[0];
            if (nkeys == 2) {
                copy(0, 1);
                put(1, theta, v);
            } else if (nkeys == 3) {
                copy(0, 1);
                copy(1, 2);
                put(2, theta, v);
            }
            return v;
        }
        if (h1) {
            const double v = val[1];
            if (nkeys == 3) {
                copy(1, 2);
                put(2, theta, v);
            }
            return v;
        }
        if (h2) return val[2];
        const double v = now(theta);
        if (nkeys == 3) { /* evict the oldest (memoisation.py:92-96) */
            copy(0, 1);
            copy(1, 2);
            put(2, theta, v);
        } else if (nkeys == 2) {
            put(2, theta, v);
            nkeys = 3;
        } else if (nkeys == 1) {
            put(1, theta, v);
            nkeys = 2;
        } else {
            put(0, theta, v);
            nkeys = 1;
        }
        return v;
    }

    /* update_error_estimate(F_fine(y) - F_coarse(y)) (aem.py:44-56, likelihood.py:142-150) */
    __device__ __forceinline__ void update(const DensityDev<D> &fine, const double (&y)[D])
    {
        double Ff[M], Fc[M];
        Density<YMC_DENSITY_POSTERIOR_LINEAR, D>::forward(fine, y, Ff);
        Density<YMC_DENSITY_POSTERIOR_LINEAR, D>::forward(p, y, Fc);
        n += 1;
        const double dn = (double)n;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if (j < p.data_dim) { /* estimation.py:36-53 */
                const double e = Ff[j] - Fc[j];
                const double delta = e - mean[j];
                mean[j] = mean[j] + delta / dn;
                const double delta2 = e - mean[j];
                m2[j] = m2[j] + delta * delta2;
            }
        }
        if (n > (uint64_t)min_data) {
            const double nm1 = (double)(n - 1);
            double mvar[M], lo = 0.0, hi = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                mvar[j] = 0.0;
                if (j < p.data_dim) {
                    mvar[j] = m2[j] / nm1;
                    lo = (j == 0 || mvar[j] < lo) ? mvar[j] : lo;
                    hi = (j == 0 || mvar[j] > hi) ? mvar[j] : hi;
                }
            }
            double scaling = 1.0;
            if (use_heuristic) { /* noise.py:41-46 */
                const double minval = lo > 1e-6 ? lo : 1e-6;
                scaling = 2.0 * hi / minval;
                scaling = scaling < 100.0 ? scaling : 100.0;
            }
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (j < p.data_dim) prec[j] = 1.0 / (scaling * mvar[j] + 1.0 / p.noise_prec[j]);
            has_prec = true;
        }
    }
};

template <int D>
__global__ __launch_bounds__(kBlock) void aem2_kernel(const AemArgs<D> a)
{
    ymc_stage_tables();
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= a.K) return;
    const uint32_t chain = (uint32_t)(a.chain0 + k);
    const int S = a.S, m = a.coarse.data_dim;

    CoarseLikelihood<D> L(a.coarse);
    L.min_data = a.min_data;
    L.use_heuristic = a.use_heuristic;
    L.eq_mode = a.eq_mode;
    L.n = a.err_n[k];
    L.has_prec = a.aem_flags[2 * k] != 0;
    L.nkeys = a.aem_flags[2 * k + 1];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        L.mean[j] = j < m ? a.err_mean[k * m + j] : 0.0;
        L.m2[j] = j < m ? a.err_m2[k * m + j] : 0.0;
        L.prec[j] = j < m ? a.aem_prec[k * m + j] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < YMC_AEM_CACHE; ++c) {
#pragma unroll
        for (int j = 0; j < D; ++j) L.key[c][j] = a.cache_key[(k * YMC_AEM_CACHE + c) * D + j];
        L.val[c] = a.cache_val[k * YMC_AEM_CACHE + c];
    }

    double x[D], wmean[D], wm2[D], sigma[D];
    ymc_load_row<D>(a.x, k, x);
#pragma unroll
    for (int j = 0; j < D; ++j) sigma[j] = a.sigma[j];
    double lpf = a.lp[k];
    uint64_t accepted = a.acc_count[k];
    uint32_t st = a.status[k];
    const bool moments = (a.flags & YMC_RECORD_MOMENTS) != 0;
    if (moments) {
        ymc_load_row<D>(a.wf_mean, k, wmean);
        ymc_load_row<D>(a.wf_m2, k, wm2);
    }
    UniformCache uc;
    uc.valid = false;
    uc.pair = 0;

    RecordClock clk(a.rec_stride, a.rec_since);
#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t tbase = (a.t0 + n) * (uint64_t)(S + 1);
        double z[D];
#pragma unroll
        for (int j = 0; j < D; ++j) z[j] = x[j];
#pragma unroll 1
        for (int s = 0; s < S; ++s) { /* surrogate.run(S + 1, state), mlda.py:102-110 */
            const uint64_t t = tbase + (uint64_t)s;
            double y[D];
            ymc_propose<D>(z, sigma, a.seed, chain, t, y);
            const double u = ymc_uniform_cached(uc, a.seed, chain, t);
            bool acc = false;
            if (!ymc_states_equal<D>(y, z, a.eq_mode)) {
                /* mrw.py:51-57: evaluate_log(proposal), then evaluate_log(state) */
                const double ly = L.query(y) + ymc_prior_log<D>(a.coarse, y);
                const double lz = L.query(z) + ymc_prior_log<D>(a.coarse, z);
                acc = ymc_accept(ly - lz, true, ymc_finite(ly), u, st);
            }
#pragma unroll
            for (int j = 0; j < D; ++j) z[j] = acc ? y[j] : z[j];
            if (a.sub_acc) a.sub_acc[((uint64_t)n * S + s) * a.K + k] = acc ? 1 : 0;
        }
        /* fine accept/reject, mlda.py:146-154: tgt(y), finest(x), finest(y), tgt(x) */
        const double u = ymc_uniform_cached(uc, a.seed, chain, tbase + (uint64_t)S);
        bool acc = false;
        double lpfy = lpf;
        if (!ymc_states_equal<D>(z, x, a.eq_mode)) {
            lpfy = Density<YMC_DENSITY_POSTERIOR_LINEAR, D>::eval(a.fine, z, st);
            const double cx = L.query(x) + ymc_prior_log<D>(a.coarse, x);
            const double cy = L.query(z) + ymc_prior_log<D>(a.coarse, z);
            const double log_ratio = ((lpfy + cx) - cy) - lpf;
            acc = ymc_accept(log_ratio, true, ymc_finite(lpfy) && ymc_finite(cy), u, st);
        }
        if (moments) ymc_welford<D>(x, (double)(a.wf_n0 + n + 1), ymc_recip_at(a.recip, n), wmean, wm2);
        if (acc) {
            L.update(a.fine, z); /* aem.py:26-56 */
#pragma unroll
            for (int j = 0; j < D; ++j) x[j] = z[j];
            lpf = lpfy;
            accepted += 1u;
        }
        if (a.acc) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
        if (clk.due()) {
            if (a.traj) ymc_store_row<D>(a.traj + (uint64_t)clk.row * a.K * D, k, x);
            if (a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + k] = lpf;
            clk.wrote();
        }
    }

    ymc_store_row<D>(a.x, k, x);
    a.lp[k] = lpf;
    a.acc_count[k] = accepted;
    a.status[k] = st;
    if (moments) {
        ymc_store_row<D>(a.wf_mean, k, wmean);
        ymc_store_row<D>(a.wf_m2, k, wm2);
    }
    a.err_n[k] = L.n;
    a.aem_flags[2 * k] = L.has_prec ? 1 : 0;
    a.aem_flags[2 * k + 1] = L.nkeys;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        if (j < m) {
            a.err_mean[k * m + j] = L.mean[j];
            a.err_m2[k * m + j] = L.m2[j];
            a.aem_prec[k * m + j] = L.prec[j];
        }
    }
#pragma unroll
    for (int c = 0; c < YMC_AEM_CACHE; ++c) {
#pragma unroll
        for (int j = 0; j < D; ++j) a.cache_key[(k * YMC_AEM_CACHE + c) * D + j] = L.key[c][j];
        a.cache_val[k * YMC_AEM_CACHE + c] = L.val[c];
    }
}

}  // namespace

template <int D>
hipError_t ymc_launch_aem(const AemArgs<D> &a, hipStream_t s)
{
    hipLaunchKernelGGL((aem2_kernel<D>), dim3((unsigned)((a.K + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

#define YMC_INSTANTIATE_AEM(D) template hipError_t ymc_launch_aem<D>(const AemArgs<D> &, hipStream_t);
YMC_INSTANTIATE_AEM(1)
YMC_INSTANTIATE_AEM(2)
YMC_INSTANTIATE_AEM(3)
YMC_INSTANTIATE_AEM(4)
YMC_INSTANTIATE_AEM(5)
YMC_INSTANTIATE_AEM(6)
YMC_INSTANTIATE_AEM(7)
YMC_INSTANTIATE_AEM(8)
