/*
 * ymc_api.hip -- the C ABI of include/ymc.h: handles, device memory, launches.
 * No exceptions cross the boundary; every failure returns a negative
 * ymc_status and leaves a message for ymc_last_error().
 */
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <functional>
#include <vector>

#include "ymc_aem.h"
#include "ymc_kernels.h"
#include "ymc_kinds.h"
#include "ymc_nested.h"
#include "ymc_post.h"
#include "ymc_tile.h"

int ymc_internal_pack_summary(ymc_run *run, uint64_t **dev_out, uint64_t *words, hipStream_t *stream, int *device);

namespace {

thread_local std::string g_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

/* a failed runtime call also leaves its code in the runtime's "last error", where the next launch check
 * (hipGetLastError after a kernel launch) would find it and blame the launch: an allocation refused for lack of memory
 * made every later launch on the context fail with "out of memory".  Reading the last error resets it. */
#define YMC_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                           \
            return fail(YMC_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
        }                                                                                      \
    } while (0)

#define YMC_TRY(call)          \
    do {                       \
        int rc_ = (call);      \
        if (rc_ != YMC_OK) return rc_; \
    } while (0)

/* page-locked host memory, grown on demand */
struct PinnedBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~PinnedBuf() { release(); }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    int reserve(size_t n);
};

struct DeviceBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DeviceBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n)
    {
        release();
        if (n == 0) return YMC_OK;
        YMC_HIP(hipMalloc(&p, n));
        bytes = n;
        return YMC_OK;
    }
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

int PinnedBuf::reserve(size_t n)
{
    if (n <= bytes) return YMC_OK;
    release();
    YMC_HIP(hipHostMalloc(&p, n, hipHostMallocDefault));
    bytes = n;
    return YMC_OK;
}

/* host copy of one lowered density + its device-resident arrays */
struct DensityHost {
    ymc_density_desc d{};
    std::vector<double> mean, whiten, data, noise_prec, prior_mean, prior_prec, A, b, design;
    std::vector<double> dbar, nprec; /* POSTERIOR_LINEAR: ymc_linear_data_summary */
    std::vector<double> noise_linv, prior_linv; /* dense covariances: inverse Cholesky factors (empty: diagonal) */
    /* matrix-core path with a dense noise covariance: the problem is whitened when it is lowered (A, b, data
     * hold L^-1 A, L^-1 b, L^-1 D_i and the noise is the identity, ymc_lower_times); the forward map a caller
     * asks for is still A v + b */
    std::vector<double> A_raw, b_raw;
    size_t off_A_raw = 0, off_b_raw = 0;
    int m_raw = 0; /* a dense prior covariance on that path: the forward map's own output count (d.data_dim counts
                      the d pseudo-observations of the prior too) */
    size_t off_nlinv = 0, off_plinv = 0;
    double lik_const = 0.0;
    DeviceBuf dev;  /* one allocation: data | noise_prec | prior_mean | prior_prec | A | b | design | dbar | nprec */
    size_t off_data = 0, off_noise = 0, off_pmean = 0, off_pprec = 0, off_A = 0, off_b = 0, off_design = 0,
           off_dbar = 0, off_nprec = 0;
    /* large-dimension linear posterior (ymc_tile.hip): KS k-steps of 4, MT output tiles of 16 */
    bool tile = false;
    bool wide = false; /* beyond the matrix-core kernels' 256 parameters: the composed wide path (WideArgs, ymc_tile.h) */
    int KS = 0, MT = 0;
    DeviceBuf tile_blob;
    DeviceBuf aem_rows; /* tile adaptive error model: noise precision [m_pad] | data rows [n_data][m_pad] */
};

}  // namespace

/* Handles may be destroyed in any order (a garbage-collecting host language gives no guarantee): a
 * context outlives its problems, runs and communicators, a problem its runs.  `users` counts the
 * dependents; a destroy call on an object that is still in use only marks it released, and the
 * last dependent to go frees it. */
struct ymc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop{};
    int users = 0;
    bool released = false;
    /* clock probe (ymc_ctx_clock_probe_*): its own stream, so that it runs NEXT to the chain kernels */
    hipStream_t probe_stream = nullptr;
    uint64_t *probe_out = nullptr;
    bool probe_pending = false;
    /* post-processing scratch (series-major copy of the record, results), kept between calls: allocating
     * 10 GB per ymc_run_iat call cost more than the kernels */
    DeviceBuf post_series, post_out;
};

struct ymc_problem {
    ymc_ctx *ctx = nullptr;
    mutable int users = 0;
    mutable bool released = false;
    int method = 0, eq_mode = 0, S = 0, D = 0, kind = 0;
    int aem_min_data = 0, aem_heuristic = 0;
    double pcn_c1 = 0.0, pcn_c2 = 0.0;
    bool tile = false;
    DensityHost tgt, sur;
    std::vector<double> sigma;
    DeviceBuf chol; /* dense proposal covariance: lower Cholesky factor [D][D] */
    DeviceBuf chol_frag; /* the same as MFMA A-operand fragments [MT_L][KS][64] (matrix-core kernel) */
    bool wide = false;   /* d > 256: the composed wide path */
    DeviceBuf wide_consts; /* sigma[d] | prior mean[d] on the device */
    int MT_L = 0;
    /* MLDA_NESTED: P of levels 0..E-2 (level E-1's is tgt), Q of levels 1..E-1 (index 0 unused) */
    int E = 0;
    DensityHost lvlP[YMC_MAX_LEVELS], lvlQ[YMC_MAX_LEVELS];
    int lvl_steps[YMC_MAX_LEVELS] = {};
    uint64_t lvl_cost[YMC_MAX_LEVELS] = {}; /* transitions one step of level e spans */
    DeviceBuf lvlP_dev, lvlQ_dev;                     /* DensityDev<D>[E] each */
    DeviceBuf aem_pair_blob; /* matrix-core adaptive error model: both levels' A fragments interleaved pair by pair (TileArgs::blob_fc) */
    DeviceBuf tile_levels;                            /* matrix-core path: TileLevel[2 E], P / Q of level l at 2 l, 2 l + 1 */
};

struct ymc_run {
    ymc_ctx *ctx = nullptr;
    const ymc_problem *problem = nullptr;
    uint64_t seed = 0, chain0 = 0, K = 0;
    uint32_t flags = 0;
    uint64_t capacity = 0, recorded = 0, steps_done = 0;
    uint32_t stride = 1; /* record_stride: rows hold the states after transitions stride, 2 stride, ... of the run */
    /* beyond 256 parameters: which kernel advances this run -- fixed at creation, because the persistent kernels write the
     * trajectory in the matrix-core kernels' TILED layout and the launches per stage in plain rows */
    enum WideKind { WIDE_STAGED = 0, WIDE_FUSED, WIDE_FUSED2, WIDE_DENSE, WIDE_GROUP } wide_kind = WIDE_STAGED;
    bool traj_tiled() const { return problem->tile || wide_kind != WIDE_STAGED; }
    DeviceBuf x, lp, lpc, acc_count, wf_mean, wf_m2, status, traj, acc, lp_out, sub_acc, summary, x_before, recip;
    DeviceBuf level_acc; /* MLDA_NESTED: [E-1][K] */
    DeviceBuf handoff;   /* Lotka-Volterra rows kernel in hand-over pieces: transitions published per chain wave */
    PinnedBuf fault;     /* ... and the count of hand-overs a wave gave up on (YMC_CHAIN_HANDOFF): one word of page-locked
                          * host memory the kernel adds to with a system-scope atomic, read by the host without a copy */
    DeviceBuf x_tiled;   /* two-level tile kernel: the carried state in the record's tiled layout */
    DeviceBuf w_tiled;   /* dense proposal on the tile kernel: L xi in the record's tiled layout */
    DeviceBuf err_n, err_mean, err_m2, aem_prec, aem_flags, cache_key, cache_val; /* AEM2 */
    DeviceBuf err_mean_alt, err_m2_alt; /* AEM2 on the matrix cores: the kernel's second copy of mean / M2 (scratch) */
    DeviceBuf wide_xi, wide_u, wide_y, wide_yt, wide_lpy, wide_flag; /* the wide path's per-transition scratch */
    DeviceBuf wide_xch, wide_arrived; /* small ensembles on the wide path: a group's exchange slots and arrival counters */
    DeviceBuf wide_z, wide_zt, wide_lpcz;                            /* ... and the two-level sampler's sub-chain state */
    DeviceBuf wide_differs;                                          /* [K] proposal-differs-from-state flags */
    DeviceBuf wide_terms;                                            /* small ensembles: the evaluation's residual terms */
    DeviceBuf wide_lvl_x, wide_lvl_lp; /* the nest on the wide path: states (plain | tiled) and P | Q of levels 0 .. E-2 */
    /* egress (ymc_run_egress_*): a copy stream, two slots of page-locked staging */
    struct EgressSlot {
        PinnedBuf traj, acc, lp;
        DeviceBuf untiled;          /* tile-kernel record: rows untiled on the device before they travel */
        hipEvent_t ready = nullptr; /* compute stream: everything issued before _begin */
        hipEvent_t done = nullptr;  /* copy stream: the slot's copies */
        bool pending = false;
        uint64_t first = 0, rows = 0;
        uint32_t what = 0;
    } egress[YMC_EGRESS_SLOTS];
    hipStream_t copy_stream = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;  /* launches since the last query */
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
};

namespace {

int copy_array(std::vector<double> &dst, const double *src, size_t n, const char *what)
{
    if (n == 0) {
        dst.clear();
        return YMC_OK;
    }
    if (!src) return fail(YMC_ERR_INVALID, "density description: %s is null", what);
    dst.assign(src, src + n);
    for (double v : dst)
        if (!std::isfinite(v)) return fail(YMC_ERR_INVALID, "density description: %s is not finite", what);
    return YMC_OK;
}

bool needs_tile(const ymc_density_desc &in)
{
    return in.kind == YMC_DENSITY_POSTERIOR_LINEAR && (in.dim > YMC_LANE_DMAX || in.data_dim > YMC_LANE_MMAX);
}

/* force_tile: lower a linear posterior for the matrix-core path even if it would fit the lane kernels (the
 * levels of a two-level sampler run in ONE kernel: both on the same path) */
/* may_stream: the sampler can take its forward matrix from L2 through the workgroup's LDS double buffer
 * (linear_tile_kernel<.., ASTREAM>: random walk with a diagonal proposal, pCN): any number of outputs whose
 * constants fit the LDS, parameter dimensions up to 256 */
int lower_density(ymc_ctx *ctx, const ymc_density_desc &in, DensityHost &out, int expect_dim, bool force_tile = false,
                  bool may_stream = false, bool may_wide = false)
{
    (void)ctx;
    out.d = in;
    const int d = in.dim;
    if (d < 1) return fail(YMC_ERR_INVALID, "parameter dimension must be positive, got %d", d);
    if (expect_dim > 0 && d != expect_dim)
        return fail(YMC_ERR_INVALID, "surrogate dimension %d differs from target dimension %d", d, expect_dim);
    const bool big = needs_tile(in) || (force_tile && in.kind == YMC_DENSITY_POSTERIOR_LINEAR);
    if (d > YMC_LANE_DMAX && !big)
        return fail(YMC_ERR_UNSUPPORTED, "parameter dimension %d exceeds the lane-per-chain limit %d", d,
                    YMC_LANE_DMAX);
    switch (in.kind) {
    case YMC_DENSITY_GAUSS_WHITENED:
        YMC_TRY(copy_array(out.mean, in.mean, d, "mean"));
        YMC_TRY(copy_array(out.whiten, in.whiten, (size_t)d * d, "whiten"));
        if (!std::isfinite(in.log_norm)) return fail(YMC_ERR_INVALID, "log_norm is not finite");
        break;
    case YMC_DENSITY_GAUSS_ISO1D:
        if (d != 1) return fail(YMC_ERR_INVALID, "GAUSS_ISO1D needs dim 1, got %d", d);
        YMC_TRY(copy_array(out.mean, in.mean, 1, "mean"));
        if (!(in.variance > 0.0) || !std::isfinite(in.variance))
            return fail(YMC_ERR_INVALID, "variance must be positive and finite");
        break;
    default: /* an extension kind (ymc_kinds.def) shares the posterior's common fields; anything else is unknown */
        if (!ymc_ext_kind(in.kind)) return fail(YMC_ERR_UNSUPPORTED, "unknown density kind %d", in.kind);
        [[fallthrough]];
    case YMC_DENSITY_POSTERIOR_LINEAR:
    case YMC_DENSITY_POSTERIOR_LV: {
        if (in.n_data < 1 || in.data_dim < 1)
            return fail(YMC_ERR_INVALID, "data must have at least one row and one column");
        const size_t nd = in.n_data, m = in.data_dim;
        YMC_TRY(copy_array(out.data, in.data, nd * m, "data"));
        YMC_TRY(copy_array(out.noise_prec, in.noise_precision, m, "noise_precision"));
        YMC_TRY(copy_array(out.prior_mean, in.prior_mean, d, "prior_mean"));
        YMC_TRY(copy_array(out.prior_prec, in.prior_precision, d, "prior_precision"));
        if (big && (in.noise_chol_inv || in.prior_chol_inv) && in.kind != YMC_DENSITY_POSTERIOR_LINEAR)
            return fail(YMC_ERR_UNSUPPORTED, "a dense covariance at this size needs a linear forward map");
        if (big && in.prior_chol_inv && in.likelihood_only)
            return fail(YMC_ERR_UNSUPPORTED, "the pCN proposal draws from a diagonal prior covariance");
        if (in.noise_chol_inv || in.prior_chol_inv) {
            auto lower_tri = [&](std::vector<double> &dst, const double *src, size_t n, const char *what) -> int {
                YMC_TRY(copy_array(dst, src, n * n, what));
                for (size_t i = 0; i < n; ++i) {
                    for (size_t j = i + 1; j < n; ++j) dst[i * n + j] = 0.0;
                    if (!(dst[i * n + i] > 0.0)) return fail(YMC_ERR_INVALID, "%s: the diagonal of an inverse Cholesky factor is positive", what);
                }
                return YMC_OK;
            };
            if (in.noise_chol_inv) YMC_TRY(lower_tri(out.noise_linv, in.noise_chol_inv, m, "noise_chol_inv"));
            if (in.prior_chol_inv) YMC_TRY(lower_tri(out.prior_linv, in.prior_chol_inv, (size_t)d, "prior_chol_inv"));
        }
        if (in.kind == YMC_DENSITY_POSTERIOR_LINEAR) {
            if (big) {
                /* the matrix-core tile kernels: d <= 128 with A staged in LDS; the single-level samplers also stream
                 * A from L2 (d <= 256, any m whose constants fit) */
                const int d_max = may_stream ? 256 : 128;
                if (d > d_max && may_wide) {
                    out.wide = true; /* any dimension, any number of outputs (dense covariances are folded in below as ever) */
                    out.KS = 4 * ((d + 15) / 16);
                    const size_t m_wide = m + (out.prior_linv.empty() ? 0 : (size_t)d);
                    out.MT = (int)((m_wide + 15) / 16);
                    out.MT += out.MT & 1;
                } else if (d > d_max)
                    return fail(YMC_ERR_UNSUPPORTED, "parameter dimension %d exceeds the matrix-core kernels' %d%s", d, d_max,
                                may_stream ? "" : " for this sampler (the random walk, pCN and delayed acceptance take 256)");
                if (!out.wide) {
                    const int ks_need = (d + 3) / 4;
                    out.KS = ks_need <= 4 ? 4 : ks_need <= 8 ? 8 : ks_need <= 16 ? 16 : ks_need <= 32 ? 32 : 64;
                    /* a dense prior covariance adds d pseudo-observations to the forward map (below) */
                    const size_t m_all = m + (out.prior_linv.empty() ? 0 : (size_t)d);
                    out.MT = (int)((m_all + 15) / 16);
                    out.MT += out.MT & 1;
                    const bool fits = out.KS <= 32 && ymc_tile_lds_bytes(out.KS, out.MT) <= kTileLdsLimit;
                    if (!fits && !(may_stream && ymc_tile_stream_lds_bytes(out.KS, out.MT) <= kTileLdsLimit))
                        return fail(YMC_ERR_UNSUPPORTED, "forward map %zu x %d%s does not fit the 160 KB LDS%s", m, d,
                                    out.prior_linv.empty() ? "" : " plus the dense prior's d x d factor",
                                    may_stream ? " (nor do its constants alone: too many outputs)"
                                               : "; only the random walk, pCN and delayed acceptance stream it from L2");
                    out.tile = true;
                }
            }
            YMC_TRY(copy_array(out.A, in.fwd_matrix, m * d, "fwd_matrix"));
            YMC_TRY(copy_array(out.b, in.fwd_shift, m, "fwd_shift"));
            if (big && !out.noise_linv.empty()) {
                /* whiten the problem (ymc_numerics.h, ymc_lower_times); the lane form of the norm is not used here */
                const std::vector<double> &Li = out.noise_linv;
                out.A_raw = out.A;
                out.b_raw = out.b;
                ymc_lower_times(Li.data(), (int32_t)m, out.A_raw.data(), d, out.A.data());
                ymc_lower_times(Li.data(), (int32_t)m, out.b_raw.data(), 1, out.b.data());
                std::vector<double> row(m);
                for (size_t i = 0; i < nd; ++i) {
                    ymc_lower_times(Li.data(), (int32_t)m, out.data.data() + i * m, 1, row.data());
                    std::copy(row.begin(), row.end(), out.data.begin() + i * m);
                }
                std::fill(out.noise_prec.begin(), out.noise_prec.end(), 1.0);
                out.noise_linv.clear();
            }
            out.dbar.resize(m);
            out.nprec.resize(m);
            ymc_linear_data_summary(out.data.data(), out.noise_prec.data(), in.n_data, in.data_dim, out.dbar.data(),
                                    out.nprec.data(), &out.lik_const);
            if (big && !out.prior_linv.empty()) {
                /* -0.5 |Lp^-1 (v - mu)|^2 is the log-likelihood of d more outputs: forward rows Lp^-1, shift 0,
                 * "data" c = Lp^-1 mu (ymc_lower_times), weight 1 (set on the summary directly: n * (1 / n) and
                 * the mean of n copies of c are not exact).  The diagonal prior is switched off (precision 0). */
                const std::vector<double> &Lp = out.prior_linv;
                std::vector<double> c((size_t)d);
                ymc_lower_times(Lp.data(), d, out.prior_mean.data(), 1, c.data());
                if (out.A_raw.empty()) {
                    out.A_raw = out.A;
                    out.b_raw = out.b;
                }
                out.m_raw = (int)m;
                out.A.insert(out.A.end(), Lp.begin(), Lp.end());
                out.b.insert(out.b.end(), (size_t)d, 0.0);
                out.dbar.insert(out.dbar.end(), c.begin(), c.end());
                out.nprec.insert(out.nprec.end(), (size_t)d, 1.0);
                std::vector<double> rows;                       /* the data and the noise precision follow the outputs */
                for (size_t i = 0; i < nd; ++i) {
                    rows.insert(rows.end(), out.data.begin() + i * m, out.data.begin() + (i + 1) * m);
                    rows.insert(rows.end(), c.begin(), c.end());
                }
                out.data.swap(rows);
                out.noise_prec.insert(out.noise_prec.end(), (size_t)d, 1.0 / (double)nd);
                std::fill(out.prior_prec.begin(), out.prior_prec.end(), 0.0);
                out.prior_linv.clear();
                out.d.data_dim = (int32_t)(m + (size_t)d);
            }
        } else if (const ymc_ext_kind_info *ext = ymc_ext_kind(in.kind)) {
            /* an extension kind: its dimension range, outputs per design row and design columns come from the registration
             * table, what else its description must satisfy from its own check (kinds/<name>.h) */
            if (d < ext->dmin || d > ext->dmax)
                return fail(YMC_ERR_INVALID, "density kind %s needs a parameter dimension in [%d, %d], got %d", ext->name,
                            ext->dmin, ext->dmax, d);
            if ((int)m != ext->outputs)
                return fail(YMC_ERR_INVALID, "density kind %s has %d forward outputs per design row, the data have %zu columns",
                            ext->name, ext->outputs, m);
            YMC_TRY(copy_array(out.design, in.design, nd * (size_t)ext->design_cols, "design"));
            if (const char *why = ext->check(in)) return fail(YMC_ERR_INVALID, "density kind %s: %s", ext->name, why);
        } else {
            if (d != 2 || m != 2) return fail(YMC_ERR_INVALID, "Lotka-Volterra needs dim 2 and data_dim 2");
            if (in.lv_rtol < 0.0 || !std::isfinite(in.lv_rtol) || (in.lv_rtol > 0.0 && !(in.lv_atol >= 0.0)) ||
                !std::isfinite(in.lv_atol))
                return fail(YMC_ERR_INVALID, "lv_rtol and lv_atol must be finite and not negative");
            if (in.lv_rtol > 0.0 && in.lv_rtol < 1e-13)
                return fail(YMC_ERR_INVALID, "lv_rtol %g is below what double precision resolves (1e-13)", in.lv_rtol);
            if (in.lv_rtol == 0.0 && in.rk4_steps < 1) return fail(YMC_ERR_INVALID, "rk4_steps must be positive");
            if (!std::isfinite(in.t_final) || !std::isfinite(in.alpha) || !std::isfinite(in.gamma))
                return fail(YMC_ERR_INVALID, "Lotka-Volterra constants must be finite");
            YMC_TRY(copy_array(out.design, in.design, nd * 2, "design"));
        }
        /* one device block for the arrays the kernels read through scalar loads */
        std::vector<double> blob;
        auto put = [&](const std::vector<double> &v, size_t &off) {
            off = blob.size();
            blob.insert(blob.end(), v.begin(), v.end());
        };
        put(out.data, out.off_data);
        put(out.noise_prec, out.off_noise);
        put(out.prior_mean, out.off_pmean);
        put(out.prior_prec, out.off_pprec);
        put(out.A, out.off_A);
        put(out.b, out.off_b);
        put(out.design, out.off_design);
        put(out.dbar, out.off_dbar);
        put(out.nprec, out.off_nprec);
        put(out.noise_linv, out.off_nlinv);
        put(out.prior_linv, out.off_plinv);
        put(out.A_raw, out.off_A_raw);
        put(out.b_raw, out.off_b_raw);
        YMC_TRY(out.dev.alloc(blob.size() * sizeof(double)));
        YMC_HIP(hipMemcpy(out.dev.p, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice));
        break;
    }
    }
    return YMC_OK;
}

template <int D>
DensityDev<D> device_view(const DensityHost &h)
{
    DensityDev<D> v;
    std::memset(&v, 0, sizeof v);
    for (int j = 0; j < D && j < (int)h.mean.size(); ++j) v.mean[j] = h.mean[j];
    for (int j = 0; j < D * D && j < (int)h.whiten.size(); ++j) v.whiten[j] = h.whiten[j];
    v.log_norm = h.d.log_norm;
    v.variance = h.d.variance;
    v.n_data = h.d.n_data;
    v.data_dim = h.d.data_dim;
    v.rk4_steps = h.d.rk4_steps;
    v.lv_rtol = h.d.lv_rtol;
    v.lv_atol = h.d.lv_rtol > 0.0 ? h.d.lv_atol : 0.0;
    v.likelihood_only = h.d.likelihood_only;
    const double *base = h.dev.as<double>();
    if (base) {
        v.data = base + h.off_data;
        v.noise_prec = base + h.off_noise;
        v.prior_mean = base + h.off_pmean;
        v.prior_prec = base + h.off_pprec;
        v.A = base + h.off_A;
        v.b = base + h.off_b;
        v.design = base + h.off_design;
        v.dbar = base + h.off_dbar;
        v.nprec = base + h.off_nprec;
        if (!h.noise_linv.empty()) v.noise_linv = base + h.off_nlinv;
        if (!h.prior_linv.empty()) v.prior_linv = base + h.off_plinv;
    }
    v.lik_const = h.lik_const;
    v.t_final = h.d.t_final;
    v.alpha = h.d.alpha;
    v.gamma = h.d.gamma;
    return v;
}

template <int D>
int eval_logdensity_dev(ymc_ctx *ctx, const ymc_problem *p, const DensityHost &h, uint64_t n, const double *x_dev,
                        double *out_dev, uint32_t *status_dev)
{
    YMC_HIP(ymc_launch_logdensity<D>(p->kind, device_view<D>(h), n, x_dev, out_dev, status_dev, ctx->stream));
    return YMC_OK;
}

#define YMC_DISPATCH_D(D_, EXPR)                                                         \
    switch (D_) {                                                                        \
    case 1: { constexpr int DD = 1; EXPR; } break;                                       \
    case 2: { constexpr int DD = 2; EXPR; } break;                                       \
    case 3: { constexpr int DD = 3; EXPR; } break;                                       \
    case 4: { constexpr int DD = 4; EXPR; } break;                                       \
    case 5: { constexpr int DD = 5; EXPR; } break;                                       \
    case 6: { constexpr int DD = 6; EXPR; } break;                                       \
    case 7: { constexpr int DD = 7; EXPR; } break;                                       \
    case 8: { constexpr int DD = 8; EXPR; } break;                                       \
    default: return fail(YMC_ERR_UNSUPPORTED, "parameter dimension %d has no kernel", D_); \
    }

/* raw_forward: the forward map as the caller described it (a whitened problem evaluates L^-1 (A v + b)) */
BigLinearArgs big_view(const DensityHost &h, bool raw_forward = false)
{
    BigLinearArgs b{};
    const double *base = h.dev.as<double>();
    const bool raw = raw_forward && !h.A_raw.empty();
    b.A = base + (raw ? h.off_A_raw : h.off_A);
    b.b = base + (raw ? h.off_b_raw : h.off_b);
    b.dbar = base + h.off_dbar;
    b.nprec = base + h.off_nprec;
    b.lik_const = h.lik_const;
    b.prior_mean = base + h.off_pmean;
    b.prior_prec = base + h.off_pprec;
    b.d = h.d.dim;
    b.m = raw && h.m_raw ? h.m_raw : h.d.data_dim;
    b.likelihood_only = h.d.likelihood_only;
    return b;
}

int eval_logdensity(ymc_ctx *ctx, const ymc_problem *p, const DensityHost &h, uint64_t n, const double *x_dev,
                    double *out_dev, uint32_t *status_dev)
{
    if (h.tile || h.wide) {
        if (status_dev) YMC_HIP(hipMemsetAsync(status_dev, 0, n * sizeof(uint32_t), ctx->stream));
        YMC_HIP(ymc_launch_linear_big_eval(big_view(h), n, x_dev, out_dev, nullptr, ctx->stream));
        return YMC_OK;
    }
    YMC_DISPATCH_D(p->D, return eval_logdensity_dev<DD>(ctx, p, h, n, x_dev, out_dev, status_dev));
    return YMC_OK;
}

/* Lotka-Volterra, single level: with fewer than ~2 waves per SIMD at one lane per
 * chain, give every design row its own lane (mrw_lv_rows_kernel).  The results
 * are identical; YMC_LV_ROWS=0|1 overrides the choice (tests run both). */
int choose_lv_rows(const ymc_run *r)
{
    const ymc_problem *p = r->problem;
    if (p->kind != YMC_DENSITY_POSTERIOR_LV || p->method != YMC_METHOD_MRW) return 0;
    const int nd = p->tgt.d.n_data;
    if (nd < 2 || nd > 64) return 0;
    if (const char *env = std::getenv("YMC_LV_ROWS")) return env[0] == '1';
    const uint64_t simds = 4ull * (uint64_t)r->ctx->prop.multiProcessorCount;
    return (r->K + 63) / 64 < simds; /* measured: 2.5x at 16 384 chains, break-even near 65 536 */
}

/* Random walk with at most one 1024-thread workgroup (256 chains) per CU, i.e. at most one wave of chains
 * per SIMD: let helper waves make the draws (mrw_split_kernel, ymc_kernels.hip).  Identical results;
 * YMC_SPLIT=0|1 overrides the choice (tests run both). */
int choose_split(const ymc_run *r)
{
    const ymc_problem *p = r->problem;
    if (p->method != YMC_METHOD_MRW || p->D > 2 || p->kind == YMC_DENSITY_POSTERIOR_LV || p->chol.p) return 0;
    int np = 3; /* producer waves per chain wave */
    if (const char *env = std::getenv("YMC_SPLIT_NP")) np = env[0] == '2' ? 2 : 3;
    if (const char *env = std::getenv("YMC_SPLIT")) return env[0] == '1' ? np : 0;
    return (r->K + 255) / 256 <= (uint64_t)r->ctx->prop.multiProcessorCount ? np : 0;
}

template <int D>
int launch_advance(ymc_run *r, uint64_t nsteps)
{
    const ymc_problem *p = r->problem;
    ChainArgs<D> a;
    std::memset(&a, 0, sizeof a);
    a.tgt = device_view<D>(p->tgt);
    if (p->method == YMC_METHOD_MLDA2) a.sur = device_view<D>(p->sur);
    for (int j = 0; j < D; ++j) a.sigma[j] = p->sigma[j];
    a.pcn_c1 = p->pcn_c1;
    a.pcn_c2 = p->pcn_c2;
    a.seed = r->seed;
    a.chain0 = r->chain0;
    a.K = r->K;
    a.t0 = r->steps_done;
    a.wf_n0 = r->steps_done;
    a.nsteps = (uint32_t)nsteps;
    a.flags = r->flags;
    a.eq_mode = p->eq_mode;
    a.S = p->S;
    a.lv_rows = choose_lv_rows(r);
    if constexpr (D == 2) {
        if (a.lv_rows && p->eq_mode == YMC_EQ_EXACT && !p->tgt.d.likelihood_only && !p->chol.p && !(p->tgt.d.lv_rtol > 0.0)) {
            /* a number of chain waves that is not a multiple of the SIMD count leaves CUs idle for the tail of the
             * launch: cut it into a whole number of equal pieces per SIMD (mrw_lv_rows_balanced_kernel).  Pieces
             * start on draw rounds; identical results; YMC_LV_PIECES=n forces n pieces (0: never), the tests do */
            const int nd = p->tgt.d.n_data, cpw = 64 / nd;
            const uint64_t G = (r->K + cpw - 1) / cpw, simds = 4ull * (uint64_t)r->ctx->prop.multiProcessorCount;
            uint64_t pieces = 0;
            if (G > simds && G % simds != 0) pieces = ((G + simds - 1) / simds) * simds;
            if (const char *env = std::getenv("YMC_LV_PIECES")) pieces = (uint64_t)std::strtoull(env, nullptr, 10);
            if (pieces > 0 && nsteps % (uint64_t)nd == 0) {
                uint64_t len = (G * nsteps + pieces - 1) / pieces;
                len = ((len + nd - 1) / nd) * nd;
                /* a piece spans at most two chain waves and waits for at most one predecessor's head */
                if (len <= nsteps && 2 * len >= nsteps && pieces < (1ull << 31)) {
                    if (r->handoff.bytes < G * sizeof(uint32_t)) YMC_TRY(r->handoff.alloc(G * sizeof(uint32_t)));
                    YMC_HIP(hipMemsetAsync(r->handoff.p, 0, G * sizeof(uint32_t), r->ctx->stream));
                    if (!r->fault.p) { /* sticky for the run's lifetime: a record with a hole never turns good again */
                        YMC_TRY(r->fault.reserve(sizeof(uint32_t)));
                        *static_cast<volatile uint32_t *>(r->fault.p) = 0u;
                    }
                    a.lv_pieces = (int32_t)pieces;
                    a.lv_piece_len = (uint32_t)len;
                    a.handoff = r->handoff.as<uint32_t>();
                    a.fault = static_cast<uint32_t *>(r->fault.p);
                    a.lv_poll_limit = 1u << 22;
                    /* test-only: YMC_LV_POLL_LIMIT lowers the limit so that the give-up path runs (tests/test_gpu_parity.py) */
                    if (const char *env = std::getenv("YMC_LV_POLL_LIMIT")) a.lv_poll_limit = (uint32_t)std::strtoul(env, nullptr, 10);
                }
            }
        }
    }
    a.rec_stride = r->stride;
    a.rec_since = (uint32_t)(r->steps_done % r->stride);
    a.split = choose_split(r);
    a.chol = p->chol.as<double>();
    a.x = r->x.as<double>();
    a.lp = r->lp.as<double>();
    a.lpc = r->lpc.as<double>();
    a.acc_count = r->acc_count.as<uint64_t>();
    a.wf_mean = r->wf_mean.as<double>();
    a.wf_m2 = r->wf_m2.as<double>();
    a.status = r->status.as<uint32_t>();
    if (r->flags & YMC_RECORD_MOMENTS) {
        if (r->recip.bytes < nsteps * sizeof(double)) YMC_TRY(r->recip.alloc(nsteps * sizeof(double)));
        YMC_HIP(ymc_launch_recip(r->steps_done, nsteps, r->recip.as<double>(), r->ctx->stream));
        a.recip = r->recip.as<double>();
    }
    const uint64_t row = r->recorded;
    if (r->flags & YMC_RECORD_TRAJECTORY) a.traj = r->traj.as<double>() + row * r->K * D;
    if (r->flags & YMC_RECORD_ACCEPT) a.acc = r->acc.as<uint8_t>() + row * r->K;
    if (r->flags & YMC_RECORD_LOGPOST) a.lp_out = r->lp_out.as<double>() + row * r->K;
    if ((r->flags & YMC_RECORD_SUB_ACCEPT) && p->method == YMC_METHOD_MLDA2)
        a.sub_acc = r->sub_acc.as<uint8_t>() + row * (uint64_t)p->S * r->K;

    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!r->pool.empty()) {
        ev = r->pool.back();
        r->pool.pop_back();
    } else {
        YMC_HIP(hipEventCreate(&ev.first));
        YMC_HIP(hipEventCreate(&ev.second));
    }
    YMC_HIP(hipEventRecord(ev.first, r->ctx->stream));
    hipError_t e = ymc_launch_chain<D>(p->kind, p->method, a, r->ctx->stream);
    if (e != hipSuccess) {
        r->pool.push_back(ev);
        return fail(YMC_ERR_DEVICE, "chain kernel launch failed: %s", hipGetErrorString(e));
    }
    YMC_HIP(hipEventRecord(ev.second, r->ctx->stream));
    r->timed.push_back(ev);
    return YMC_OK;
}

template <int D>
int launch_advance_aem(ymc_run *r, uint64_t nsteps)
{
    const ymc_problem *p = r->problem;
    AemArgs<D> a;
    std::memset(&a, 0, sizeof a);
    a.fine = device_view<D>(p->tgt);
    a.coarse = device_view<D>(p->sur);
    for (int j = 0; j < D; ++j) a.sigma[j] = p->sigma[j];
    a.seed = r->seed;
    a.chain0 = r->chain0;
    a.K = r->K;
    a.t0 = r->steps_done;
    a.wf_n0 = r->steps_done;
    a.nsteps = (uint32_t)nsteps;
    a.flags = r->flags;
    a.eq_mode = p->eq_mode;
    a.S = p->S;
    a.rec_stride = r->stride;
    a.rec_since = (uint32_t)(r->steps_done % r->stride);
    a.min_data = p->aem_min_data;
    a.use_heuristic = p->aem_heuristic;
    a.x = r->x.as<double>();
    a.lp = r->lp.as<double>();
    a.acc_count = r->acc_count.as<uint64_t>();
    a.wf_mean = r->wf_mean.as<double>();
    a.wf_m2 = r->wf_m2.as<double>();
    a.status = r->status.as<uint32_t>();
    a.err_n = r->err_n.as<uint64_t>();
    a.err_mean = r->err_mean.as<double>();
    a.err_m2 = r->err_m2.as<double>();
    a.aem_prec = r->aem_prec.as<double>();
    a.aem_flags = r->aem_flags.as<int32_t>();
    a.cache_key = r->cache_key.as<double>();
    a.cache_val = r->cache_val.as<double>();
    if (r->flags & YMC_RECORD_MOMENTS) {
        if (r->recip.bytes < nsteps * sizeof(double)) YMC_TRY(r->recip.alloc(nsteps * sizeof(double)));
        YMC_HIP(ymc_launch_recip(r->steps_done, nsteps, r->recip.as<double>(), r->ctx->stream));
        a.recip = r->recip.as<double>();
    }
    const uint64_t row = r->recorded;
    if (r->flags & YMC_RECORD_TRAJECTORY) a.traj = r->traj.as<double>() + row * r->K * D;
    if (r->flags & YMC_RECORD_ACCEPT) a.acc = r->acc.as<uint8_t>() + row * r->K;
    if (r->flags & YMC_RECORD_LOGPOST) a.lp_out = r->lp_out.as<double>() + row * r->K;
    if (r->flags & YMC_RECORD_SUB_ACCEPT) a.sub_acc = r->sub_acc.as<uint8_t>() + row * (uint64_t)p->S * r->K;

    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!r->pool.empty()) {
        ev = r->pool.back();
        r->pool.pop_back();
    } else {
        YMC_HIP(hipEventCreate(&ev.first));
        YMC_HIP(hipEventCreate(&ev.second));
    }
    YMC_HIP(hipEventRecord(ev.first, r->ctx->stream));
    hipError_t e = ymc_launch_aem<D>(a, r->ctx->stream);
    if (e != hipSuccess) {
        r->pool.push_back(ev);
        return fail(YMC_ERR_DEVICE, "adaptive-error-model kernel launch failed: %s", hipGetErrorString(e));
    }
    YMC_HIP(hipEventRecord(ev.second, r->ctx->stream));
    r->timed.push_back(ev);
    return YMC_OK;
}

/* the nest's lowered densities as two device arrays (they exceed the kernel-argument segment) */
template <int D>
int upload_levels(ymc_problem *p)
{
    std::vector<DensityDev<D>> P(p->E), Q(p->E);
    std::memset(P.data(), 0, P.size() * sizeof(DensityDev<D>));
    std::memset(Q.data(), 0, Q.size() * sizeof(DensityDev<D>));
    for (int e = 0; e < p->E; ++e) {
        P[e] = device_view<D>(e == p->E - 1 ? p->tgt : p->lvlP[e]);
        if (e >= 1) Q[e] = device_view<D>(p->lvlQ[e]);
    }
    YMC_TRY(p->lvlP_dev.alloc(P.size() * sizeof(DensityDev<D>)));
    YMC_TRY(p->lvlQ_dev.alloc(Q.size() * sizeof(DensityDev<D>)));
    YMC_HIP(hipMemcpy(p->lvlP_dev.p, P.data(), p->lvlP_dev.bytes, hipMemcpyHostToDevice));
    YMC_HIP(hipMemcpy(p->lvlQ_dev.p, Q.data(), p->lvlQ_dev.bytes, hipMemcpyHostToDevice));
    return YMC_OK;
}

template <int D>
int launch_advance_nested(ymc_run *r, uint64_t nsteps)
{
    const ymc_problem *p = r->problem;
    NestedArgs<D> a;
    std::memset(&a, 0, sizeof a);
    a.P = p->lvlP_dev.as<DensityDev<D>>();
    a.Q = p->lvlQ_dev.as<DensityDev<D>>();
    for (int e = 0; e < YMC_MAX_LEVELS; ++e) a.steps[e] = p->lvl_steps[e];
    a.E = p->E;
    a.eq_mode = p->eq_mode;
    for (int j = 0; j < D; ++j) a.sigma[j] = p->sigma[j];
    a.seed = r->seed;
    a.chain0 = r->chain0;
    a.K = r->K;
    a.t0 = r->steps_done;
    a.cost_top = p->lvl_cost[p->E - 1];
    a.rec_stride = r->stride;
    a.rec_since = (uint32_t)(r->steps_done % r->stride);
    a.wf_n0 = r->steps_done;
    a.nsteps = (uint32_t)nsteps;
    a.flags = r->flags;
    a.x = r->x.as<double>();
    a.lp = r->lp.as<double>();
    a.lpq = r->lpc.as<double>();
    a.acc_count = r->acc_count.as<uint64_t>();
    a.level_acc = r->level_acc.as<uint64_t>();
    a.wf_mean = r->wf_mean.as<double>();
    a.wf_m2 = r->wf_m2.as<double>();
    a.status = r->status.as<uint32_t>();
    if (r->flags & YMC_RECORD_MOMENTS) {
        if (r->recip.bytes < nsteps * sizeof(double)) YMC_TRY(r->recip.alloc(nsteps * sizeof(double)));
        YMC_HIP(ymc_launch_recip(r->steps_done, nsteps, r->recip.as<double>(), r->ctx->stream));
        a.recip = r->recip.as<double>();
    }
    const uint64_t row = r->recorded;
    if (r->flags & YMC_RECORD_TRAJECTORY) a.traj = r->traj.as<double>() + row * r->K * D;
    if (r->flags & YMC_RECORD_ACCEPT) a.acc = r->acc.as<uint8_t>() + row * r->K;
    if (r->flags & YMC_RECORD_LOGPOST) a.lp_out = r->lp_out.as<double>() + row * r->K;

    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!r->pool.empty()) {
        ev = r->pool.back();
        r->pool.pop_back();
    } else {
        YMC_HIP(hipEventCreate(&ev.first));
        YMC_HIP(hipEventCreate(&ev.second));
    }
    YMC_HIP(hipEventRecord(ev.first, r->ctx->stream));
    hipError_t e = a.E > 4 ? ymc_launch_nested_deep<D>(p->kind, a, r->ctx->stream) : ymc_launch_nested<D>(p->kind, a, r->ctx->stream);
    if (e != hipSuccess) {
        r->pool.push_back(ev);
        return fail(YMC_ERR_DEVICE, "nested chain kernel launch failed: %s", hipGetErrorString(e));
    }
    YMC_HIP(hipEventRecord(ev.second, r->ctx->stream));
    r->timed.push_back(ev);
    return YMC_OK;
}

/* A in MFMA A-operand fragment order + the padded constant arrays, one block */
int build_tile_blob(ymc_problem *p, DensityHost &h)
{
    const int KS = h.KS, MT = h.MT, d = h.d.dim, m = h.d.data_dim;
    const size_t m_pad = 16 * (size_t)MT, d_pad = 4 * (size_t)KS;
    std::vector<double> blob((size_t)MT * KS * 64 + m_pad * 3 + 3 * d_pad, 0.0);
    for (int mt = 0; mt < MT; ++mt)
        for (int s = 0; s < KS; ++s)
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * mt + (l & 15), col = 4 * s + (l >> 4);
                if (row < m && col < d) blob[((size_t)mt * KS + s) * 64 + l] = h.A[(size_t)row * d + col];
            }
    double *bias = blob.data() + (size_t)MT * KS * 64;
    double *nprec = bias + m_pad, *dbar = nprec + m_pad;
    double *pmean = dbar + m_pad, *pprec = pmean + d_pad, *sigma = pprec + d_pad;
    for (int j = 0; j < m; ++j) {
        bias[j] = h.b[j];
        nprec[j] = h.nprec[j];
        dbar[j] = h.dbar[j];
    }
    for (int j = 0; j < d; ++j) {
        pmean[j] = h.prior_mean[j];
        pprec[j] = h.prior_prec[j];
        sigma[j] = p->sigma[j];
    }
    YMC_TRY(h.tile_blob.alloc(blob.size() * sizeof(double)));
    YMC_HIP(hipMemcpy(h.tile_blob.p, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice));
    return YMC_OK;
}

/* the coarse level's raw noise precision and data rows, padded to the tile kernel's output count */
int build_aem_rows(DensityHost &h)
{
    const size_t m = h.d.data_dim, m_pad = 16 * (size_t)h.MT, nd = h.d.n_data;
    std::vector<double> rows(m_pad * (1 + nd), 0.0);
    for (size_t j = 0; j < m; ++j) rows[j] = h.noise_prec[j];
    for (size_t i = 0; i < nd; ++i)
        for (size_t j = 0; j < m; ++j) rows[m_pad * (1 + i) + j] = h.data[i * m + j];
    YMC_TRY(h.aem_rows.alloc(rows.size() * sizeof(double)));
    YMC_HIP(hipMemcpy(h.aem_rows.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
    return YMC_OK;
}

/* fine tiles 2p, 2p + 1 | coarse tiles 2p, 2p + 1 for every pair p: what the fine pass of linear_tile_aem2_lds_kernel streams */
int build_aem_pair_blob(ymc_problem *p)
{
    const DensityHost &f = p->tgt, &c = p->sur;
    if (f.MT != c.MT || f.KS != c.KS || (f.MT & 1)) return YMC_OK; /* the kernel then keeps the coarse matrix in LDS */
    const size_t pair = 2 * (size_t)f.KS * 64 * sizeof(double);
    YMC_TRY(p->aem_pair_blob.alloc(2 * (size_t)f.MT * f.KS * 64 * sizeof(double)));
    for (int q = 0; q < f.MT / 2; ++q) {
        char *dst = static_cast<char *>(p->aem_pair_blob.p) + 2 * (size_t)q * pair;
        YMC_HIP(hipMemcpy(dst, static_cast<const char *>(f.tile_blob.p) + (size_t)q * pair, pair, hipMemcpyDeviceToDevice));
        YMC_HIP(hipMemcpy(dst + pair, static_cast<const char *>(c.tile_blob.p) + (size_t)q * pair, pair, hipMemcpyDeviceToDevice));
    }
    return YMC_OK;
}

uint64_t tiled_row_doubles(const ymc_run *r)
{
    return ((r->K + 15) / 16) * (uint64_t)r->problem->tgt.KS * 64;
}

/* the wide path's kernel for a run (launch_advance_wide): the persistent ones where the shape, the ensemble and what the run
 * records allow -- moments are replayed from the record, which must then hold every transition */
ymc_run::WideKind pick_wide_kind(const ymc_run *r)
{
    const ymc_problem *p = r->problem;
    if (!p->wide || p->method == YMC_METHOD_MLDA_NESTED) return ymc_run::WIDE_STAGED;
    const bool two = p->method == YMC_METHOD_MLDA2;
    const int cus = r->ctx->prop.multiProcessorCount;
    const bool moments = (r->flags & YMC_RECORD_MOMENTS) != 0;
    if (moments && !((r->flags & YMC_RECORD_TRAJECTORY) && r->stride == 1)) return ymc_run::WIDE_STAGED;
    if (two)
        return (!p->chol_frag.p && p->tgt.KS == p->sur.KS && ymc_wide_fused2_lds_bytes(p->tgt.KS, p->sur.MT, p->tgt.MT, r->K, cus))
                   ? ymc_run::WIDE_FUSED2 : ymc_run::WIDE_STAGED;
    if (p->chol_frag.p)
        return (p->method != YMC_METHOD_PCN && ymc_wide_fused_dense_lds_bytes(p->tgt.KS, p->tgt.MT, p->MT_L)) ? ymc_run::WIDE_DENSE
                                                                                                              : ymc_run::WIDE_STAGED;
    if (ymc_wide_split_group(p->tgt.KS, p->tgt.MT, r->K, cus)) return ymc_run::WIDE_GROUP;
    return ymc_wide_fused_lds_bytes(p->tgt.KS, p->tgt.MT, r->K, cus) ? ymc_run::WIDE_FUSED : ymc_run::WIDE_STAGED;
}

/* Beyond 256 parameters (random walk with a diagonal proposal, pCN over a linear model): a transition is five plain
 * launches over [K][d] arrays -- the draws, the proposal, the canonical-order evaluation (linear_big_eval_kernel, the
 * kernel behind evaluate_log at these sizes), the decision per chain, the move with record and moments.  The same
 * arithmetic in the same order as the matrix-core kernels (which are bit-identical to the checker), at any d and m. */
int launch_advance_wide(ymc_run *r, uint64_t nsteps)
{
    const ymc_problem *p = r->problem;
    hipStream_t s = r->ctx->stream;
    const uint64_t K = r->K, d = (uint64_t)p->D;
    const bool two = p->method == YMC_METHOD_MLDA2;
    const uint64_t S = two ? (uint64_t)p->S : 0;
    WideArgs a{};
    a.K = K;
    a.d = p->D;
    a.pcn = p->method == YMC_METHOD_PCN;
    a.pcn_c1 = p->pcn_c1;
    a.pcn_c2 = p->pcn_c2;
    a.sigma = p->wide_consts.as<double>();
    a.pmean = a.sigma + d;
    a.xi = r->wide_xi.as<double>();
    a.u = r->wide_u.as<double>();
    a.x = r->x.as<double>();
    a.y = r->wide_y.as<double>();
    a.y_tiled = r->wide_yt.as<double>();
    a.KS = p->tgt.KS;
    a.lp = r->lp.as<double>();
    a.lpy = r->wide_lpy.as<double>();
    a.acc_count = r->acc_count.as<uint64_t>();
    a.status = r->status.as<uint32_t>();
    a.accflag = r->wide_flag.as<uint8_t>();
    a.differs = r->wide_differs.as<uint32_t>();
    auto eval = [&](const DensityHost &h, const double *tiled, double *out) -> int {
        YMC_HIP(ymc_launch_wide_eval(h.tile_blob.as<double>(), h.KS, h.MT, h.lik_const, h.d.likelihood_only, tiled, K, out, s,
                                     r->wide_terms.as<double>()));
        return YMC_OK;
    };

    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!r->pool.empty()) {
        ev = r->pool.back();
        r->pool.pop_back();
    } else {
        YMC_HIP(hipEventCreate(&ev.first));
        YMC_HIP(hipEventCreate(&ev.second));
    }
    YMC_HIP(hipEventRecord(ev.first, s));
    /* the random walk with a diagonal proposal and pCN: the whole launch as ONE persistent kernel with the chains' state
     * on the chip (linear_wide_fused_kernel) where the proposal tile fits the LDS -- identical results; moments (they would
     * be two more slices of registers per chain) are replayed from the record afterwards, as the tile kernels' are */
    const int cus = r->ctx->prop.multiProcessorCount;
    const bool moments = (r->flags & YMC_RECORD_MOMENTS) != 0;
    /* moments: replayed from the record the launch writes -- which must then hold every transition */
    if (r->wide_kind != ymc_run::WIDE_STAGED && nsteps >= (1ull << 32))
        return fail(YMC_ERR_INVALID, "more than 2^32 transitions in one launch");
    const bool dense_fused = r->wide_kind == ymc_run::WIDE_DENSE;
    const int split_G = r->wide_kind == ymc_run::WIDE_GROUP ? ymc_wide_split_group(p->tgt.KS, p->tgt.MT, K, cus) : 0;
    if (r->wide_kind == ymc_run::WIDE_GROUP && !split_G)
        return fail(YMC_ERR_INVALID, "the kernel selection changed under a live run (YMC_WIDE_* are read when the run is created)");
    if (r->wide_kind == ymc_run::WIDE_FUSED || r->wide_kind == ymc_run::WIDE_DENSE || r->wide_kind == ymc_run::WIDE_GROUP) {
        if (moments) {
            if (!r->x_before.p) YMC_TRY(r->x_before.alloc(r->x.bytes));
            YMC_HIP(hipMemcpyAsync(r->x_before.p, r->x.p, r->x.bytes, hipMemcpyDeviceToDevice, s));
        }
        WideFusedArgs f{};
        f.blob = p->tgt.tile_blob.as<double>();
        f.lik_const = p->tgt.lik_const;
        f.KS = p->tgt.KS, f.MT = p->tgt.MT, f.d = p->D;
        f.likelihood_only = p->tgt.d.likelihood_only;
        f.pcn = a.pcn, f.pcn_c1 = a.pcn_c1, f.pcn_c2 = a.pcn_c2;
        f.seed = r->seed, f.chain0 = r->chain0, f.K = K, f.t0 = r->steps_done;
        f.nsteps = (uint32_t)nsteps;
        f.rec_stride = r->stride;
        f.rec_since = (uint32_t)(r->steps_done % r->stride);
        f.x = r->x.as<double>(), f.lp = r->lp.as<double>();
        f.acc_count = r->acc_count.as<uint64_t>(), f.status = r->status.as<uint32_t>();
        const uint64_t row = r->recorded;
        if (r->flags & YMC_RECORD_TRAJECTORY) f.traj = r->traj.as<double>() + row * tiled_row_doubles(r); /* tiled rows */
        if (r->flags & YMC_RECORD_ACCEPT) f.acc = r->acc.as<uint8_t>() + row * K;
        if (r->flags & YMC_RECORD_LOGPOST) f.lp_out = r->lp_out.as<double>() + row * K;
        hipError_t e;
        if (dense_fused) {
            f.chol_frag = p->chol_frag.as<double>();
            f.MT_L = p->MT_L;
            e = ymc_launch_wide_fused_dense(f, s);
        } else if (split_G) {
            WideSplitArgs sp{};
            sp.f = f;
            const size_t xb = ymc_wide_split_exchange_bytes(p->tgt.MT, K), cb = ymc_wide_split_counter_bytes(K);
            if (r->wide_xch.bytes < xb) YMC_TRY(r->wide_xch.alloc(xb));
            if (r->wide_arrived.bytes < cb) YMC_TRY(r->wide_arrived.alloc(cb));
            if (!r->fault.p) { /* sticky for the run's lifetime, as the hand-over pieces' */
                YMC_TRY(r->fault.reserve(sizeof(uint32_t)));
                *static_cast<volatile uint32_t *>(r->fault.p) = 0u;
            }
            sp.xch = r->wide_xch.as<double>();
            sp.arrived = r->wide_arrived.as<uint32_t>();
            sp.fault = static_cast<uint32_t *>(r->fault.p);
            sp.poll_limit = 1u << 22;
            if (const char *env = std::getenv("YMC_LV_POLL_LIMIT")) sp.poll_limit = (uint32_t)std::strtoul(env, nullptr, 10);
            e = ymc_launch_wide_split(sp, cus, s);
        } else {
            e = ymc_launch_wide_fused(f, cus, s);
        }
        if (e == hipSuccess && moments)
            e = ymc_launch_welford_replay(f.traj, r->x_before.as<double>(), nsteps, K, p->D, p->tgt.KS, r->steps_done,
                                          r->wf_mean.as<double>(), r->wf_m2.as<double>(), s);
        if (e != hipSuccess) {
            r->pool.push_back(ev);
            return fail(YMC_ERR_DEVICE, "wide fused kernel launch failed: %s", hipGetErrorString(e));
        }
        YMC_HIP(hipEventRecord(ev.second, s));
        r->timed.push_back(ev);
        return YMC_OK;
    }
    /* two-level delayed acceptance the same way: one persistent launch, x and z on the chip (linear_wide_fused2_kernel) */
    if (r->wide_kind == ymc_run::WIDE_FUSED2) {
        if (moments) {
            if (!r->x_before.p) YMC_TRY(r->x_before.alloc(r->x.bytes));
            YMC_HIP(hipMemcpyAsync(r->x_before.p, r->x.p, r->x.bytes, hipMemcpyDeviceToDevice, s));
        }
        WideFused2Args f{};
        f.blob_c = p->sur.tile_blob.as<double>(), f.blob_f = p->tgt.tile_blob.as<double>();
        f.lik_const_c = p->sur.lik_const, f.lik_const_f = p->tgt.lik_const;
        f.KS = p->tgt.KS, f.MT_c = p->sur.MT, f.MT_f = p->tgt.MT, f.d = p->D, f.S = p->S;
        f.seed = r->seed, f.chain0 = r->chain0, f.K = K, f.t0 = r->steps_done;
        f.nsteps = (uint32_t)nsteps;
        f.rec_stride = r->stride;
        f.rec_since = (uint32_t)(r->steps_done % r->stride);
        f.x = r->x.as<double>(), f.lp = r->lp.as<double>(), f.lpc = r->lpc.as<double>();
        if (r->x_tiled.bytes < tiled_row_doubles(r) * sizeof(double)) YMC_TRY(r->x_tiled.alloc(tiled_row_doubles(r) * sizeof(double)));
        f.x_park = r->x_tiled.as<double>();
        f.acc_count = r->acc_count.as<uint64_t>(), f.status = r->status.as<uint32_t>();
        const uint64_t row = r->recorded;
        if (r->flags & YMC_RECORD_TRAJECTORY) f.traj = r->traj.as<double>() + row * tiled_row_doubles(r); /* tiled rows */
        if (r->flags & YMC_RECORD_ACCEPT) f.acc = r->acc.as<uint8_t>() + row * K;
        if (r->flags & YMC_RECORD_LOGPOST) f.lp_out = r->lp_out.as<double>() + row * K;
        if (r->flags & YMC_RECORD_SUB_ACCEPT) f.sub_acc = r->sub_acc.as<uint8_t>() + row * S * K;
        hipError_t e = ymc_launch_wide_fused2(f, cus, s);
        if (e == hipSuccess && moments)
            e = ymc_launch_welford_replay(f.traj, r->x_before.as<double>(), nsteps, K, p->D, p->tgt.KS, r->steps_done,
                                          r->wf_mean.as<double>(), r->wf_m2.as<double>(), s);
        if (e != hipSuccess) {
            r->pool.push_back(ev);
            return fail(YMC_ERR_DEVICE, "wide fused two-level kernel launch failed: %s", hipGetErrorString(e));
        }
        YMC_HIP(hipEventRecord(ev.second, s));
        r->timed.push_back(ev);
        return YMC_OK;
    }
    uint64_t rows = 0;
    for (uint64_t n = 0; n < nsteps; ++n) {
        const uint64_t step = r->steps_done + n;
        const bool due = (step + 1) % r->stride == 0;
        const uint64_t row = r->recorded + rows;
        uint8_t *acc_row = (r->flags & YMC_RECORD_ACCEPT) ? r->acc.as<uint8_t>() + row * K : nullptr; /* stride 1 when recorded */
        double *traj_row = (due && (r->flags & YMC_RECORD_TRAJECTORY)) ? r->traj.as<double>() + row * K * d : nullptr;
        double *lp_row = (due && (r->flags & YMC_RECORD_LOGPOST)) ? r->lp_out.as<double>() + row * K : nullptr;
        double *wf_mean = nullptr, *wf_m2 = nullptr;
        if (r->flags & YMC_RECORD_MOMENTS) {
            wf_mean = r->wf_mean.as<double>();
            wf_m2 = r->wf_m2.as<double>();
            a.wf_dn = (double)(step + 1);
            a.wf_rdn = 1.0 / a.wf_dn;
        }
        if (!two) { /* one transition of the random walk / pCN */
            a.from = a.x, a.cur_lp = a.lp, a.dst = a.x, a.dst_tiled = nullptr, a.count = a.acc_count;
            a.acc_row = acc_row, a.traj_row = traj_row, a.lp_row = lp_row, a.wf_mean = wf_mean, a.wf_m2 = wf_m2;
            if (p->chol_frag.p) {
                /* dense proposal covariance (covariance.py:91-92): xi <- L xi on the matrix cores, then y = x + 1 * (L xi) */
                YMC_HIP(ymc_launch_wide_draws(a, r->seed, r->chain0, step, r->wide_yt.as<double>(), s));
                YMC_HIP(ymc_launch_wide_rows(p->chol_frag.as<double>(), p->tgt.KS, p->MT_L, p->D, r->wide_yt.as<double>(), K,
                                             r->wide_xi.as<double>(), s));
                YMC_HIP(ymc_launch_wide_propose(a, s));
            } else {
                YMC_HIP(ymc_launch_wide_draw_propose(a, r->seed, r->chain0, step, s));
            }
            YMC_TRY(eval(p->tgt, a.y_tiled, r->wide_lpy.as<double>()));
            YMC_HIP(ymc_launch_wide_decide(a, s));
            YMC_HIP(ymc_launch_wide_apply(a, s));
        } else {
            /* one fine step of two-level delayed acceptance (mlda.py:100-110,146-154, as mlda2_body): the sub-chain of S
             * coarse random-walk transitions from z = x, then the fine decision; fine step n owns transitions n (S + 1) + s */
            double *z = r->wide_z.as<double>(), *zt = r->wide_zt.as<double>(), *lpcz = r->wide_lpcz.as<double>();
            const uint64_t tbase = step * (S + 1);
            YMC_HIP(ymc_launch_wide_copy(a, a.x, z, zt, s));
            YMC_HIP(hipMemcpyAsync(lpcz, r->lpc.p, K * sizeof(double), hipMemcpyDeviceToDevice, s));
            a.from = z, a.cur_lp = lpcz, a.dst = z, a.dst_tiled = zt, a.count = nullptr;
            a.traj_row = nullptr, a.lp_row = nullptr, a.wf_mean = nullptr, a.wf_m2 = nullptr;
            for (uint64_t sub = 0; sub < S; ++sub) {
                a.acc_row = (r->flags & YMC_RECORD_SUB_ACCEPT) ? r->sub_acc.as<uint8_t>() + (row * S + sub) * K : nullptr;
                YMC_HIP(ymc_launch_wide_draw_propose(a, r->seed, r->chain0, tbase + sub, s));
                YMC_TRY(eval(p->sur, a.y_tiled, r->wide_lpy.as<double>()));
                YMC_HIP(ymc_launch_wide_decide(a, s));
                YMC_HIP(ymc_launch_wide_apply(a, s));
            }
            YMC_HIP(ymc_launch_rng_fill(r->seed, r->chain0, K, tbase + S, 1, p->D, nullptr, r->wide_u.as<double>(), s));
            YMC_TRY(eval(p->tgt, zt, r->wide_lpy.as<double>()));
            a.z = z, a.lpfz = r->wide_lpy.as<double>(), a.lpc = r->lpc.as<double>(), a.lpcz = lpcz;
            a.acc_row = acc_row, a.traj_row = traj_row, a.lp_row = lp_row, a.wf_mean = wf_mean, a.wf_m2 = wf_m2;
            a.dst_tiled = nullptr;
            YMC_HIP(ymc_launch_wide_differs(a, z, a.x, s));
            YMC_HIP(ymc_launch_wide_fine_decide(a, s));
            YMC_HIP(ymc_launch_wide_fine_apply(a, s));
        }
        if (due) ++rows;
    }
    YMC_HIP(hipEventRecord(ev.second, s));
    r->timed.push_back(ev);
    return YMC_OK;
}

/* The nest of levels (YMC_METHOD_MLDA_NESTED) beyond 256 parameters: the walk of ymc_nested.hip / the checker's
 * ymo_nest_run as host recursion over the wide path's kernels.  Level l < E - 1 keeps the state of its run in
 * wide_lvl_x[l] (plain and matrix-core layout) and P / Q of that state in wide_lvl_lp; every decision takes the next
 * transition index in execution order. */
int launch_advance_wide_nested(ymc_run *r, uint64_t nsteps)
{
    const ymc_problem *p = r->problem;
    hipStream_t s = r->ctx->stream;
    const uint64_t K = r->K, d = (uint64_t)p->D;
    const int E = p->E;
    WideArgs a{};
    a.K = K;
    a.d = p->D;
    a.sigma = p->wide_consts.as<double>();
    a.pmean = a.sigma + d;
    a.xi = r->wide_xi.as<double>();
    a.u = r->wide_u.as<double>();
    a.y = r->wide_y.as<double>();
    a.y_tiled = r->wide_yt.as<double>();
    a.KS = p->tgt.KS;
    a.lpy = r->wide_lpy.as<double>();
    a.status = r->status.as<uint32_t>();
    a.accflag = r->wide_flag.as<uint8_t>();
    a.differs = r->wide_differs.as<uint32_t>();
    const uint64_t tiled = ((K + 15) / 16) * (uint64_t)p->tgt.KS * 64;
    auto X = [&](int l) { return r->wide_lvl_x.as<double>() + (uint64_t)l * (K * d + tiled); };
    auto Xt = [&](int l) { return X(l) + K * d; };
    auto lpP = [&](int l) { return r->wide_lvl_lp.as<double>() + (uint64_t)(2 * l) * K; };
    auto lpQ = [&](int l) { return r->wide_lvl_lp.as<double>() + (uint64_t)(2 * l + 1) * K; };
    double *tmpQ = r->wide_lpcz.as<double>();
    auto eval = [&](const DensityHost &h, const double *pts, double *out) -> int {
        YMC_HIP(ymc_launch_wide_eval(h.tile_blob.as<double>(), h.KS, h.MT, h.lik_const, h.d.likelihood_only, pts, K, out, s,
                                     r->wide_terms.as<double>()));
        return YMC_OK;
    };
    auto Pof = [&](int l) -> const DensityHost & { return l == E - 1 ? p->tgt : p->lvlP[l]; };
    uint64_t t = 0;
    /* _accept_reject of level l >= 1 (mlda.py:35-43; the top's: :146-154): the proposal Y against the level's state */
    auto decide = [&](int l, double *x, double *xt, const double *Y, const double *Yt, double *P, double *Q, uint64_t *count) -> int {
        YMC_HIP(ymc_launch_rng_fill(r->seed, r->chain0, K, t, 1, p->D, nullptr, r->wide_u.as<double>(), s));
        YMC_TRY(eval(Pof(l), Yt, r->wide_lpy.as<double>()));
        YMC_TRY(eval(p->lvlQ[l], Yt, tmpQ));
        a.x = x, a.z = Y, a.lpfz = r->wide_lpy.as<double>(), a.lpcz = tmpQ, a.lp = P, a.lpc = Q, a.acc_count = count;
        a.dst_tiled = xt;
        YMC_HIP(ymc_launch_wide_differs(a, Y, x, s));
        YMC_HIP(ymc_launch_wide_fine_decide(a, s));
        YMC_HIP(ymc_launch_wide_fine_apply(a, s));
        ++t;
        return YMC_OK;
    };
    /* MetropolisHastings.run of level l from the state in X(l) */
    std::function<int(int)> run_level = [&](int l) -> int {
        a.acc_row = nullptr, a.traj_row = nullptr, a.lp_row = nullptr, a.wf_mean = nullptr, a.wf_m2 = nullptr;
        if (l == 0) {
            YMC_TRY(eval(p->lvlP[0], Xt(0), lpP(0)));
            for (int st = 0; st < p->lvl_steps[0]; ++st) {
                a.from = X(0), a.cur_lp = lpP(0), a.dst = X(0), a.dst_tiled = Xt(0), a.count = r->level_acc.as<uint64_t>();
                YMC_HIP(ymc_launch_wide_draw_propose(a, r->seed, r->chain0, t, s));
                YMC_TRY(eval(p->lvlP[0], a.y_tiled, r->wide_lpy.as<double>()));
                YMC_HIP(ymc_launch_wide_decide(a, s));
                YMC_HIP(ymc_launch_wide_apply(a, s));
                ++t;
            }
            return YMC_OK;
        }
        YMC_TRY(eval(p->lvlP[l], Xt(l), lpP(l)));
        YMC_TRY(eval(p->lvlQ[l], Xt(l), lpQ(l)));
        for (int st = 0; st < p->lvl_steps[l]; ++st) {
            YMC_HIP(ymc_launch_wide_copy(a, X(l), X(l - 1), Xt(l - 1), s));
            YMC_TRY(run_level(l - 1));
            a.acc_row = nullptr, a.traj_row = nullptr, a.lp_row = nullptr, a.wf_mean = nullptr, a.wf_m2 = nullptr;
            YMC_TRY(decide(l, X(l), Xt(l), X(l - 1), Xt(l - 1), lpP(l), lpQ(l), r->level_acc.as<uint64_t>() + (uint64_t)l * K));
        }
        return YMC_OK;
    };

    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!r->pool.empty()) {
        ev = r->pool.back();
        r->pool.pop_back();
    } else {
        YMC_HIP(hipEventCreate(&ev.first));
        YMC_HIP(hipEventCreate(&ev.second));
    }
    YMC_HIP(hipEventRecord(ev.first, s));
    uint64_t rows = 0;
    for (uint64_t n = 0; n < nsteps; ++n) {
        const uint64_t step = r->steps_done + n;
        const bool due = (step + 1) % r->stride == 0;
        const uint64_t row = r->recorded + rows;
        t = step * p->lvl_cost[E - 1];
        YMC_HIP(ymc_launch_wide_copy(a, r->x.as<double>(), X(E - 2), Xt(E - 2), s));
        YMC_TRY(run_level(E - 2));                                                   /* mlda.py:100-116 */
        a.acc_row = (r->flags & YMC_RECORD_ACCEPT) ? r->acc.as<uint8_t>() + row * K : nullptr;
        a.traj_row = (due && (r->flags & YMC_RECORD_TRAJECTORY)) ? r->traj.as<double>() + row * K * d : nullptr;
        a.lp_row = (due && (r->flags & YMC_RECORD_LOGPOST)) ? r->lp_out.as<double>() + row * K : nullptr;
        a.wf_mean = nullptr, a.wf_m2 = nullptr;
        if (r->flags & YMC_RECORD_MOMENTS) {
            a.wf_mean = r->wf_mean.as<double>();
            a.wf_m2 = r->wf_m2.as<double>();
            a.wf_dn = (double)(step + 1);
            a.wf_rdn = 1.0 / a.wf_dn;
        }
        YMC_TRY(decide(E - 1, r->x.as<double>(), nullptr, X(E - 2), Xt(E - 2), r->lp.as<double>(), r->lpc.as<double>(),
                       r->acc_count.as<uint64_t>()));                                /* mlda.py:146-154 */
        if (due) ++rows;
    }
    YMC_HIP(hipEventRecord(ev.second, s));
    r->timed.push_back(ev);
    return YMC_OK;
}

int launch_advance_tile(ymc_run *r, uint64_t nsteps)
{
    const ymc_problem *p = r->problem;
    const bool aem2 = p->method == YMC_METHOD_AEM2;
    const bool mlda2 = p->method == YMC_METHOD_MLDA2 || aem2;
    const bool nested = p->method == YMC_METHOD_MLDA_NESTED;
    const DensityHost &h = mlda2 ? p->sur : nested ? p->lvlP[0] : p->tgt; /* the level staged in LDS */
    TileArgs a{};
    a.blob = h.tile_blob.as<double>();
    a.lik_const = h.lik_const;
    a.d = p->D;
    a.m = h.d.data_dim;
    a.MT = h.MT;
    if (mlda2) {
        a.blob_f = p->tgt.tile_blob.as<double>();
        a.lik_const_f = p->tgt.lik_const;
        a.m_f = p->tgt.d.data_dim;
        a.MT_f = p->tgt.MT;
        a.S = p->S;
        a.lpc = r->lpc.as<double>();
        a.x_tiled = r->x_tiled.as<double>();
        if (r->flags & YMC_RECORD_SUB_ACCEPT) a.sub_acc = r->sub_acc.as<uint8_t>() + r->recorded * (uint64_t)p->S * r->K;
    }
    if (nested) {
        a.levels = p->tile_levels.as<TileLevel>();
        a.E = p->E;
        for (int e = 0; e < YMC_MAX_LEVELS; ++e) a.steps[e] = p->lvl_steps[e];
        a.cost_top = p->lvl_cost[p->E - 1];
        a.level_acc = r->level_acc.as<uint64_t>();
        a.lpc = r->lpc.as<double>();
        a.x_tiled = r->x_tiled.as<double>();
    }
    if (p->chol_frag.p) {
        a.chol_frag = p->chol_frag.as<double>();
        a.w_tiled = r->w_tiled.as<double>();
        a.MT_L = p->MT_L;
    }
    if (aem2) {
        a.n_data = h.d.n_data;
        a.min_data = p->aem_min_data;
        a.use_heuristic = p->aem_heuristic;
        a.aem_rows = h.aem_rows.as<double>();
        a.blob_fc = p->aem_pair_blob.as<double>();
        a.err_n = r->err_n.as<uint64_t>();
        a.err_mean_t = r->err_mean.as<double>();
        a.err_m2_t = r->err_m2.as<double>();
        a.err_mean_alt_t = r->err_mean_alt.as<double>();
        a.err_m2_alt_t = r->err_m2_alt.as<double>();
        a.prec_t = r->aem_prec.as<double>();
        a.aem_flags = r->aem_flags.as<int32_t>();
        a.cache_id = r->cache_key.as<uint64_t>();
        a.cache_val = r->cache_val.as<double>();
    }
    a.seed = r->seed;
    a.chain0 = r->chain0;
    a.K = r->K;
    a.t0 = r->steps_done;
    a.nsteps = (uint32_t)nsteps;
    a.flags = r->flags;
    a.rec_stride = r->stride;
    a.rec_since = (uint32_t)(r->steps_done % r->stride);
    a.pcn = p->method == YMC_METHOD_PCN;
    a.pcn_c1 = p->pcn_c1;
    a.pcn_c2 = p->pcn_c2;
    a.x = r->x.as<double>();
    a.lp = r->lp.as<double>();
    a.acc_count = r->acc_count.as<uint64_t>();
    a.status = r->status.as<uint32_t>();
    const uint64_t row = r->recorded;
    if (r->flags & YMC_RECORD_TRAJECTORY) a.traj = r->traj.as<double>() + row * tiled_row_doubles(r);
    if (r->flags & YMC_RECORD_ACCEPT) a.acc = r->acc.as<uint8_t>() + row * r->K;
    if (r->flags & YMC_RECORD_LOGPOST) a.lp_out = r->lp_out.as<double>() + row * r->K;
    const bool moments = (r->flags & YMC_RECORD_MOMENTS) != 0;
    if (moments) {
        if (!r->x_before.p) YMC_TRY(r->x_before.alloc(r->x.bytes));
        YMC_HIP(hipMemcpyAsync(r->x_before.p, r->x.p, r->x.bytes, hipMemcpyDeviceToDevice, r->ctx->stream));
    }
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!r->pool.empty()) {
        ev = r->pool.back();
        r->pool.pop_back();
    } else {
        YMC_HIP(hipEventCreate(&ev.first));
        YMC_HIP(hipEventCreate(&ev.second));
    }
    YMC_HIP(hipEventRecord(ev.first, r->ctx->stream));
    hipError_t e = aem2     ? ymc_launch_linear_tile_aem2(h.KS, a, r->ctx->stream)
                   : mlda2  ? ymc_launch_linear_tile_mlda2(h.KS, a, r->ctx->stream)
                   : nested ? ymc_launch_linear_tile_nested(h.KS, a, r->ctx->stream)
                            : ymc_launch_linear_tile(h.KS, a, r->ctx->stream);
    if (e == hipSuccess && moments)
        e = ymc_launch_welford_replay(a.traj, r->x_before.as<double>(), nsteps, r->K, p->D, h.KS, r->steps_done,
                                      r->wf_mean.as<double>(), r->wf_m2.as<double>(), r->ctx->stream);
    if (e != hipSuccess) {
        r->pool.push_back(ev);
        return fail(YMC_ERR_DEVICE, "tile kernel launch failed: %s", hipGetErrorString(e));
    }
    YMC_HIP(hipEventRecord(ev.second, r->ctx->stream));
    r->timed.push_back(ev);
    return YMC_OK;
}

bool records_anything(uint32_t flags)
{
    return (flags & (YMC_RECORD_TRAJECTORY | YMC_RECORD_ACCEPT | YMC_RECORD_LOGPOST | YMC_RECORD_SUB_ACCEPT)) != 0;
}

/* An internal device error must not be delivered as data (the reference aborts a chain whose evaluation failed with
 * an exception, model/forwardModel.py:15-19): after the stream has drained, a run whose launches were cut into
 * hand-over pieces looks at its fault word (page-locked host memory; runs that never used pieces have none). */
int check_internal_faults(ymc_run *r)
{
    if (!r->fault.p) return YMC_OK;
    const uint32_t n = *static_cast<volatile const uint32_t *>(r->fault.p);
    if (n)
        return fail(YMC_ERR_DEVICE, "internal device error: %u waves gave up waiting for a hand-over "
                                    "(YMC_CHAIN_HANDOFF): transitions are missing from the record of this run", n);
    return YMC_OK;
}

int fetch_rows(ymc_run *r, const DeviceBuf &buf, uint32_t need_flag, const char *what, uint64_t first, uint64_t n,
               size_t row_bytes, void *dst, size_t dst_bytes)
{
    if (!r || !dst) return fail(YMC_ERR_INVALID, "null argument");
    if (!(r->flags & need_flag)) return fail(YMC_ERR_INVALID, "the run does not record %s", what);
    if (first > r->recorded || n > r->recorded - first) /* overflow-safe: a negative count arrives as a huge one */
        return fail(YMC_ERR_INVALID, "rows [%llu, %llu) requested but only %llu recorded", (unsigned long long)first,
                    (unsigned long long)(first + n), (unsigned long long)r->recorded);
    if (dst_bytes < n * row_bytes) return fail(YMC_ERR_INVALID, "destination too small for %s", what);
    YMC_HIP(hipSetDevice(r->ctx->device));
    YMC_HIP(hipStreamSynchronize(r->ctx->stream));
    YMC_TRY(check_internal_faults(r));
    if (n) YMC_HIP(hipMemcpy(dst, static_cast<const char *>(buf.p) + first * row_bytes, n * row_bytes, hipMemcpyDeviceToHost));
    return YMC_OK;
}

}  // namespace

extern "C" {

int ymc_abi_version(void) { return YMC_ABI_VERSION; }

const char *ymc_last_error(void) { return g_error.c_str(); }

int ymc_device_count(int *count)
{
    if (!count) return fail(YMC_ERR_INVALID, "null argument");
    YMC_HIP(hipGetDeviceCount(count));
    return YMC_OK;
}

int ymc_ctx_create(int device, ymc_ctx **out)
{
    if (!out) return fail(YMC_ERR_INVALID, "null argument");
    *out = nullptr;
    int n = 0;
    YMC_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(YMC_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    ymc_ctx *c = new (std::nothrow) ymc_ctx;
    if (!c) return fail(YMC_ERR_DEVICE, "out of host memory");
    c->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&c->prop, device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(YMC_ERR_DEVICE, "context creation failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return YMC_OK;
}

static void ctx_free(ymc_ctx *ctx)
{
    (void)hipSetDevice(ctx->device);
    if (ctx->probe_stream) {
        (void)hipStreamSynchronize(ctx->probe_stream);
        (void)hipStreamDestroy(ctx->probe_stream);
    }
    if (ctx->probe_out) (void)hipFree(ctx->probe_out);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static void ctx_drop_user(ymc_ctx *ctx)
{
    if (--ctx->users == 0 && ctx->released) ctx_free(ctx);
}

int ymc_ctx_destroy(ymc_ctx *ctx)
{
    if (!ctx || ctx->released) return YMC_OK;
    if (ctx->users > 0) {
        ctx->released = true; /* freed by its last problem / run / communicator */
        return YMC_OK;
    }
    ctx_free(ctx);
    return YMC_OK;
}



int ymc_ctx_device_info(const ymc_ctx *ctx, ymc_device_info *out)
{
    if (!ctx || !out) return fail(YMC_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof *out);
    std::snprintf(out->name, sizeof out->name, "%s", ctx->prop.name);
    std::snprintf(out->arch, sizeof out->arch, "%s", ctx->prop.gcnArchName);
    out->compute_units = ctx->prop.multiProcessorCount;
    out->clock_mhz = ctx->prop.clockRate / 1000;
    out->hbm_bytes = ctx->prop.totalGlobalMem;
    return YMC_OK;
}

int ymc_ctx_clock_probe_start(ymc_ctx *ctx, double duration_ms)
{
    if (!ctx) return fail(YMC_ERR_INVALID, "null argument");
    if (!(duration_ms > 0.0) || duration_ms > 20000.0)
        return fail(YMC_ERR_INVALID, "probe duration must be in (0, 20000] ms, got %g", duration_ms);
    if (ctx->probe_pending) return fail(YMC_ERR_INVALID, "a clock probe is already running; read it first");
    YMC_HIP(hipSetDevice(ctx->device));
    if (!ctx->probe_stream) YMC_HIP(hipStreamCreateWithFlags(&ctx->probe_stream, hipStreamNonBlocking));
    if (!ctx->probe_out) YMC_HIP(hipMalloc((void **)&ctx->probe_out, 2 * sizeof(uint64_t)));
    YMC_HIP(ymc_launch_clock_probe((uint64_t)(duration_ms * 1e5), ctx->probe_out, ctx->probe_stream));
    ctx->probe_pending = true;
    return YMC_OK;
}

int ymc_ctx_clock_probe_read(ymc_ctx *ctx, double *mhz, double *measured_ms)
{
    if (!ctx || !mhz) return fail(YMC_ERR_INVALID, "null argument");
    if (!ctx->probe_pending) return fail(YMC_ERR_INVALID, "no clock probe was started");
    YMC_HIP(hipSetDevice(ctx->device));
    uint64_t host[2] = {0, 0};
    YMC_HIP(hipMemcpyAsync(host, ctx->probe_out, sizeof host, hipMemcpyDeviceToHost, ctx->probe_stream));
    YMC_HIP(hipStreamSynchronize(ctx->probe_stream));
    ctx->probe_pending = false;
    if (host[1] == 0) return fail(YMC_ERR_DEVICE, "the clock probe reported no elapsed time");
    *mhz = (double)host[0] / (double)host[1] * 100.0; /* cycles per tick of the 100 MHz reference */
    if (measured_ms) *measured_ms = (double)host[1] * 1e-5;
    return YMC_OK;
}

int ymc_problem_create(ymc_ctx *ctx, const ymc_problem_desc *desc, ymc_problem **out)
{
    if (!ctx || !desc || !out) return fail(YMC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (desc->method != YMC_METHOD_MRW && desc->method != YMC_METHOD_MLDA2 && desc->method != YMC_METHOD_PCN &&
        desc->method != YMC_METHOD_MLDA_NESTED && desc->method != YMC_METHOD_AEM2)
        return fail(YMC_ERR_UNSUPPORTED, "unknown method %d", desc->method);
    if (desc->state_eq != YMC_EQ_EXACT && desc->state_eq != YMC_EQ_ISCLOSE)
        return fail(YMC_ERR_INVALID, "unknown state equality rule %d", desc->state_eq);
    YMC_HIP(hipSetDevice(ctx->device));
    ymc_problem *p = new (std::nothrow) ymc_problem;
    if (!p) return fail(YMC_ERR_DEVICE, "out of host memory");
    p->ctx = ctx;
    p->method = desc->method;
    p->eq_mode = desc->state_eq;
    const bool two_level = desc->method == YMC_METHOD_MLDA2 || desc->method == YMC_METHOD_AEM2;
    bool joint_tile = two_level && desc->surrogate.kind == desc->target.kind &&
                      (needs_tile(desc->target) || needs_tile(desc->surrogate));
    if (desc->method == YMC_METHOD_MLDA_NESTED && desc->n_levels >= 2 && desc->n_levels <= YMC_MAX_LEVELS &&
        desc->level_target && desc->level_correction) {
        /* all levels run in one kernel: if one posterior needs the matrix-core path, all take it */
        joint_tile = needs_tile(desc->target);
        for (int e = 0; e < desc->n_levels - 1; ++e)
            joint_tile = joint_tile || needs_tile(desc->level_target[e]) || needs_tile(desc->level_correction[e]);
    }
    /* the random walk, pCN and delayed acceptance (two-level or nested) stream a forward matrix that does not fit the
     * LDS from L2 (d <= 256) */
    const bool may_stream = desc->method == YMC_METHOD_MRW || desc->method == YMC_METHOD_PCN || desc->method == YMC_METHOD_MLDA2 ||
                            desc->method == YMC_METHOD_MLDA_NESTED;
    /* beyond 256 parameters the random walk with a diagonal proposal and pCN run as a sequence of plain kernels per
     * transition (launch_advance_wide) */
    const bool may_wide = desc->method == YMC_METHOD_MRW || desc->method == YMC_METHOD_PCN ||
                          desc->method == YMC_METHOD_MLDA2 || desc->method == YMC_METHOD_MLDA_NESTED;
    int rc = lower_density(ctx, desc->target, p->tgt, 0, joint_tile, may_stream, may_wide);
    if (rc == YMC_OK) {
        p->wide = p->tgt.wide;
        p->D = desc->target.dim;
        p->kind = desc->target.kind;
        if (p->eq_mode == YMC_EQ_ISCLOSE && p->D != 1)
            rc = fail(YMC_ERR_INVALID, "the isclose equality rule is the scalar parameter's; dim is %d", p->D);
    }
    if (rc == YMC_OK && desc->method == YMC_METHOD_AEM2) {
        /* AEMBuilder._validate_parameters (aem.py:66-79), AEMLikelihood.__init__ (likelihood.py:99-100) */
        if (desc->target.kind != YMC_DENSITY_POSTERIOR_LINEAR || desc->surrogate.kind != YMC_DENSITY_POSTERIOR_LINEAR)
            rc = fail(YMC_ERR_INVALID, "Adaptive error correction only makes sense if the target emerges from a "
                                       "Bayesian model.");
        else if (desc->aem_min_data_size < 2)
            rc = fail(YMC_ERR_INVALID, "Smallest senisible data size for AEM is 2.");
        else if (desc->target.likelihood_only || desc->surrogate.likelihood_only)
            rc = fail(YMC_ERR_INVALID, "the adaptive error model samples posteriors");
        else if (desc->surrogate.data_dim != desc->target.data_dim)
            rc = fail(YMC_ERR_INVALID, "the model error needs forward maps of equal output dimension (%d vs %d)",
                      desc->target.data_dim, desc->surrogate.data_dim);
        if (rc == YMC_OK && (desc->target.noise_chol_inv || desc->surrogate.noise_chol_inv))
            rc = fail(YMC_ERR_UNSUPPORTED, "Currently, AEM is only implemented for independent measurement noise.");
        if (rc == YMC_OK && p->tgt.tile && (desc->target.prior_chol_inv || desc->surrogate.prior_chol_inv))
            rc = fail(YMC_ERR_UNSUPPORTED, "at this dimension the adaptive error model takes diagonal prior covariances "
                                           "(a dense one is carried as extra outputs of the forward map, which the "
                                           "error model would learn from)");
        p->aem_min_data = desc->aem_min_data_size;
        p->aem_heuristic = desc->aem_use_noise_heuristic != 0;
    }
    if (rc == YMC_OK && (desc->method == YMC_METHOD_MLDA2 || desc->method == YMC_METHOD_AEM2)) {
        /* MLDABuilder._validate_parameters, mlda.py:218-269 */
        if (desc->sub_chain_length < 1) rc = fail(YMC_ERR_INVALID, "Subchain lengths not set for MLDA");
        if (rc == YMC_OK && desc->surrogate.kind != desc->target.kind)
            rc = fail(YMC_ERR_UNSUPPORTED, "surrogate density kind %d differs from target kind %d",
                      desc->surrogate.kind, desc->target.kind);
        if (rc == YMC_OK)
            rc = lower_density(ctx, desc->surrogate, p->sur, p->D, joint_tile, may_stream, may_wide && desc->method == YMC_METHOD_MLDA2);
        /* both levels run in one kernel: on the lane path each holds at most YMC_LANE_MMAX forward outputs, on
         * the matrix-core path the coarse matrix and both levels' constants share the LDS */
        if (rc == YMC_OK && p->sur.tile != p->tgt.tile)
            rc = fail(YMC_ERR_UNSUPPORTED, "the two levels do not fit one kernel (target %d x %d, surrogate %d x %d)",
                      desc->target.data_dim, p->D, desc->surrogate.data_dim, p->D);
        if (rc == YMC_OK && p->tgt.tile && desc->method == YMC_METHOD_AEM2 &&
            ymc_tile_aem2_lds_bytes(p->sur.KS, p->sur.MT, desc->surrogate.n_data) > kTileLdsLimit)
            rc = fail(YMC_ERR_UNSUPPORTED, "the coarse forward map %d x %d, both levels' constants and the data rows do "
                                           "not fit the 160 KB LDS", desc->surrogate.data_dim, p->D);
        if (rc == YMC_OK && p->tgt.tile && desc->method == YMC_METHOD_MLDA2 &&
            ymc_tile_mlda2_stream_lds_bytes(p->sur.KS, p->sur.MT, p->tgt.MT) > kTileLdsLimit)
            rc = fail(YMC_ERR_UNSUPPORTED, "both levels' constants (%d and %d outputs) do not fit the 160 KB LDS",
                      desc->surrogate.data_dim, desc->target.data_dim);
        p->S = desc->sub_chain_length;
    }
    if (rc == YMC_OK && desc->method == YMC_METHOD_MLDA_NESTED) {
        const int E = desc->n_levels;
        if (E < 2 || E > YMC_MAX_LEVELS)
            rc = fail(YMC_ERR_UNSUPPORTED, "nested delayed acceptance has 2 to %d effective levels, got %d",
                      YMC_MAX_LEVELS, E);
        else if (!desc->level_target || !desc->level_correction || !desc->level_steps)
            rc = fail(YMC_ERR_INVALID, "Surrogate targets not set for MLDA");
        p->E = E;
        for (int e = 0; rc == YMC_OK && e < E - 1; ++e) {
            const ymc_density_desc &P = desc->level_target[e], &Q = desc->level_correction[e];
            if (desc->level_steps[e] < 1) rc = fail(YMC_ERR_INVALID, "Subchain lengths not set for MLDA");
            if (rc == YMC_OK && (P.kind != desc->target.kind || Q.kind != desc->target.kind))
                rc = fail(YMC_ERR_UNSUPPORTED, "surrogate density kinds %d / %d differ from target kind %d", P.kind,
                          Q.kind, desc->target.kind);
            if (rc == YMC_OK) rc = lower_density(ctx, P, p->lvlP[e], p->D, joint_tile, may_stream, may_wide);
            if (rc == YMC_OK) rc = lower_density(ctx, Q, p->lvlQ[e + 1], p->D, joint_tile, may_stream, may_wide);
            if (rc == YMC_OK && (p->lvlP[e].tile != p->tgt.tile || p->lvlQ[e + 1].tile != p->tgt.tile))
                rc = fail(YMC_ERR_UNSUPPORTED, "the levels do not fit one kernel (mixed forward-map sizes)");
            p->lvl_steps[e] = desc->level_steps[e];
            p->lvl_cost[e] = e == 0 ? 1 : (uint64_t)p->lvl_steps[e - 1] * p->lvl_cost[e - 1] + 1;
        }
        if (rc == YMC_OK) p->lvl_cost[E - 1] = (uint64_t)p->lvl_steps[E - 2] * p->lvl_cost[E - 2] + 1;
        /* the transition index (64 bits) addresses the draws: a step of the recorded level may span 2^40 of them */
        for (int e = 1; rc == YMC_OK && e < E; ++e)
            if (p->lvl_cost[e] > (1ull << 40) || p->lvl_cost[e] < p->lvl_cost[e - 1])
                rc = fail(YMC_ERR_UNSUPPORTED, "one step of level %d spans more than 2^40 transitions", e);
    }
    if (rc == YMC_OK) {
        /* MRWBuilder._validate_parameters, mrw.py:88-91 */
        if (!desc->proposal_sigma) rc = fail(YMC_ERR_INVALID, "Proposal Covariance not set");
        if (rc == YMC_OK) rc = copy_array(p->sigma, desc->proposal_sigma, p->D, "proposal_sigma");
    }
    if (rc == YMC_OK && desc->proposal_chol) {
        if (desc->method != YMC_METHOD_MRW)
            rc = fail(YMC_ERR_UNSUPPORTED, "a dense proposal covariance is available for the random walk");
        else {
            std::vector<double> L((size_t)p->D * p->D, 0.0);
            for (int i = 0; rc == YMC_OK && i < p->D; ++i)
                for (int j = 0; j <= i; ++j) {
                    L[(size_t)i * p->D + j] = desc->proposal_chol[(size_t)i * p->D + j];
                    if (!std::isfinite(L[(size_t)i * p->D + j]))
                        rc = fail(YMC_ERR_INVALID, "proposal_chol is not finite");
                }
            auto up = [&]() -> int {
                YMC_TRY(p->chol.alloc(L.size() * sizeof(double)));
                YMC_HIP(hipMemcpy(p->chol.p, L.data(), p->chol.bytes, hipMemcpyHostToDevice));
                if (p->tgt.tile || p->wide) { /* fragments for the matrix cores: rows = components, padded to whole tile pairs */
                    const int KS = p->tgt.KS, d = p->D;
                    p->MT_L = (d + 15) / 16;
                    p->MT_L += p->MT_L & 1;
                    /* + one ring of k-steps of zeros: the fused kernel's padded k loop reads (and multiplies by zero) past KS */
                    std::vector<double> frag((size_t)p->MT_L * KS * 64 + 16 * 64, 0.0);
                    for (int mt = 0; mt < p->MT_L; ++mt)
                        for (int s = 0; s < KS; ++s)
                            for (int l = 0; l < 64; ++l) {
                                const int row = 16 * mt + (l & 15), col = 4 * s + (l >> 4);
                                if (row < d && col < d) frag[((size_t)mt * KS + s) * 64 + l] = L[(size_t)row * d + col];
                            }
                    YMC_TRY(p->chol_frag.alloc(frag.size() * sizeof(double)));
                    YMC_HIP(hipMemcpy(p->chol_frag.p, frag.data(), p->chol_frag.bytes, hipMemcpyHostToDevice));
                }
                return YMC_OK;
            };
            if (rc == YMC_OK) rc = up();
        }
    }
    if (rc == YMC_OK && desc->method == YMC_METHOD_PCN) {
        /* PreconditionedCrankNicolson.__init__, pcn.py:40-49; PCNBuilder, pcn.py:77-83 */
        if (p->kind != YMC_DENSITY_POSTERIOR_LINEAR && p->kind != YMC_DENSITY_POSTERIOR_LV && !ymc_ext_kind(p->kind))
            rc = fail(YMC_ERR_INVALID, "PCN is only defined in relation to a Bayesian model");
        else if (desc->target.prior_chol_inv)
            rc = fail(YMC_ERR_UNSUPPORTED, "the pCN proposal draws from a diagonal prior covariance");
        else if (!(desc->pcn_step_size > 0.0 && desc->pcn_step_size <= 0.5))
            rc = fail(YMC_ERR_INVALID, "PCN step size must be in (0, 0.5], got %g", desc->pcn_step_size);
        else {
            const double t = 2.0 * desc->pcn_step_size;
            p->pcn_c1 = std::sqrt(1.0 - t);
            p->pcn_c2 = std::sqrt(t);
        }
    }
    if (rc == YMC_OK && p->wide) {
        auto up = [&]() -> int {
            std::vector<double> c(p->sigma);
            if (p->chol.p) std::fill(c.begin(), c.end(), 1.0); /* dense proposal: the increment is L xi itself */
            c.insert(c.end(), p->tgt.prior_mean.begin(), p->tgt.prior_mean.end());
            c.resize(2 * (size_t)p->D, 0.0);
            YMC_TRY(p->wide_consts.alloc(c.size() * sizeof(double)));
            YMC_HIP(hipMemcpy(p->wide_consts.p, c.data(), c.size() * sizeof(double), hipMemcpyHostToDevice));
            YMC_TRY(build_tile_blob(p, p->tgt)); /* the matrix-core layout of the model for linear_wide_eval_kernel */
            if (p->method == YMC_METHOD_MLDA2) YMC_TRY(build_tile_blob(p, p->sur));
            if (p->method == YMC_METHOD_MLDA_NESTED)
                for (int e = 0; e < p->E - 1; ++e) {
                    YMC_TRY(build_tile_blob(p, p->lvlP[e]));
                    YMC_TRY(build_tile_blob(p, p->lvlQ[e + 1]));
                }
            return YMC_OK;
        };
        rc = up();
    }
    if (rc == YMC_OK && p->tgt.tile) {
        const bool two = p->method == YMC_METHOD_MLDA2 || p->method == YMC_METHOD_AEM2;
        if (rc == YMC_OK) rc = build_tile_blob(p, p->tgt);
        if (rc == YMC_OK && two) rc = build_tile_blob(p, p->sur);
        if (rc == YMC_OK && p->method == YMC_METHOD_AEM2) rc = build_aem_rows(p->sur);
        if (rc == YMC_OK && p->method == YMC_METHOD_AEM2) rc = build_aem_pair_blob(p);
        if (rc == YMC_OK && p->method == YMC_METHOD_MLDA_NESTED) {
            /* level 0's posterior is staged in LDS whole where it fits (its matrix streamed otherwise); the posteriors
             * of the levels above are read from global memory through a table of blocks */
            auto up = [&]() -> int {
                std::vector<TileLevel> lv(2 * (size_t)p->E, TileLevel{nullptr, 0.0, 0, 0});
                for (int e = 0; e < p->E; ++e) {
                    DensityHost &P = e == p->E - 1 ? p->tgt : p->lvlP[e];
                    if (e < p->E - 1) YMC_TRY(build_tile_blob(p, P));
                    lv[2 * e] = TileLevel{P.tile_blob.as<double>(), P.lik_const, P.MT, 0};
                    if (e >= 1) {
                        YMC_TRY(build_tile_blob(p, p->lvlQ[e]));
                        lv[2 * e + 1] = TileLevel{p->lvlQ[e].tile_blob.as<double>(), p->lvlQ[e].lik_const, p->lvlQ[e].MT, 0};
                    }
                }
                YMC_TRY(p->tile_levels.alloc(lv.size() * sizeof(TileLevel)));
                YMC_HIP(hipMemcpy(p->tile_levels.p, lv.data(), p->tile_levels.bytes, hipMemcpyHostToDevice));
                return YMC_OK;
            };
            rc = up();
        }
        p->tile = true;
    }
    if (rc == YMC_OK && p->method == YMC_METHOD_MLDA_NESTED && !p->tile && !p->wide) {
        auto up = [&]() -> int {
            YMC_DISPATCH_D(p->D, return upload_levels<DD>(p));
            return YMC_OK;
        };
        rc = up();
    }
    if (rc != YMC_OK) {
        delete p;
        return rc;
    }
    ctx->users += 1;
    *out = p;
    return YMC_OK;
}

static void problem_free(const ymc_problem *problem)
{
    ymc_ctx *ctx = problem->ctx;
    (void)hipSetDevice(ctx->device);
    delete problem;
    ctx_drop_user(ctx);
}

int ymc_problem_destroy(ymc_problem *problem)
{
    if (!problem || problem->released) return YMC_OK;
    if (problem->users > 0) {
        problem->released = true; /* freed by its last run */
        return YMC_OK;
    }
    problem_free(problem);
    return YMC_OK;
}

int ymc_run_create(ymc_ctx *ctx, const ymc_problem *problem, const ymc_run_desc *desc, ymc_run **out)
{
    if (!ctx || !problem || !desc || !out) return fail(YMC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (problem->ctx != ctx) return fail(YMC_ERR_INVALID, "problem belongs to another context");
    if (desc->n_chains == 0) return fail(YMC_ERR_INVALID, "n_chains must be positive");
    if (desc->chain_offset + desc->n_chains > (1ull << 32))
        return fail(YMC_ERR_INVALID, "global chain ids must stay below 2^32");
    if (!desc->initial_state) return fail(YMC_ERR_INVALID, "initial_state is null");
    if (desc->x0_rows != 1 && desc->x0_rows != desc->n_chains)
        return fail(YMC_ERR_INVALID, "initial_state must have 1 or n_chains rows");
    if (records_anything(desc->record_flags) && desc->record_capacity == 0)
        return fail(YMC_ERR_INVALID, "record_capacity must be positive when something is recorded");
    if (desc->record_stride > 1 && (desc->record_flags & (YMC_RECORD_ACCEPT | YMC_RECORD_SUB_ACCEPT)))
        return fail(YMC_ERR_INVALID, "record_stride %u thins the trajectory; the per-transition decision records "
                                     "(YMC_RECORD_ACCEPT / _SUB_ACCEPT) are full-rate data and cannot be combined with it",
                    desc->record_stride);
    if (desc->record_stride > 1 && problem->tile && (desc->record_flags & YMC_RECORD_MOMENTS))
        return fail(YMC_ERR_UNSUPPORTED, "at this dimension the Welford moments are replayed from the FULL record: "
                                         "record_stride must be 1 with YMC_RECORD_MOMENTS");
    if (problem->tile && (desc->record_flags & YMC_RECORD_MOMENTS) && !(desc->record_flags & YMC_RECORD_TRAJECTORY))
        return fail(YMC_ERR_UNSUPPORTED,
                    "at this dimension the Welford moments are replayed from the record: record the trajectory too");
    YMC_HIP(hipSetDevice(ctx->device));
    ymc_run *r = new (std::nothrow) ymc_run;
    if (!r) return fail(YMC_ERR_DEVICE, "out of host memory");
    r->ctx = ctx;
    r->problem = problem;
    r->seed = desc->seed;
    r->chain0 = desc->chain_offset;
    r->K = desc->n_chains;
    r->flags = desc->record_flags;
    r->stride = desc->record_stride > 1 ? desc->record_stride : 1;
    r->capacity = records_anything(r->flags) ? desc->record_capacity : 0;
    const uint64_t K = r->K;
    const int D = problem->D;
    const int S = problem->S;
    auto body = [&]() -> int {
        std::vector<double> x0((size_t)K * D);
        for (uint64_t k = 0; k < K; ++k) {
            const double *src = desc->initial_state + (desc->x0_rows == 1 ? 0 : k * D);
            for (int j = 0; j < D; ++j) {
                if (!std::isfinite(src[j])) return fail(YMC_ERR_INVALID, "initial_state is not finite");
                x0[k * D + j] = src[j];
            }
        }
        YMC_TRY(r->x.alloc(K * D * sizeof(double)));
        YMC_TRY(r->lp.alloc(K * sizeof(double)));
        YMC_TRY(r->lpc.alloc(K * sizeof(double)));
        YMC_TRY(r->acc_count.alloc(K * sizeof(uint64_t)));
        YMC_TRY(r->wf_mean.alloc(K * D * sizeof(double)));
        YMC_TRY(r->wf_m2.alloc(K * D * sizeof(double)));
        YMC_TRY(r->status.alloc(K * sizeof(uint32_t)));
        YMC_HIP(hipMemcpy(r->x.p, x0.data(), x0.size() * sizeof(double), hipMemcpyHostToDevice));
        YMC_HIP(hipMemsetAsync(r->lpc.p, 0, r->lpc.bytes, ctx->stream));
        YMC_HIP(hipMemsetAsync(r->acc_count.p, 0, r->acc_count.bytes, ctx->stream));
        YMC_HIP(hipMemsetAsync(r->wf_mean.p, 0, r->wf_mean.bytes, ctx->stream));
        YMC_HIP(hipMemsetAsync(r->wf_m2.p, 0, r->wf_m2.bytes, ctx->stream));
        YMC_HIP(hipMemsetAsync(r->status.p, 0, r->status.bytes, ctx->stream));
        r->wide_kind = pick_wide_kind(r);
        if (r->flags & YMC_RECORD_TRAJECTORY)
            YMC_TRY(r->traj.alloc(r->capacity * (r->traj_tiled() ? tiled_row_doubles(r) : K * D) * sizeof(double)));
        if (r->flags & YMC_RECORD_ACCEPT) YMC_TRY(r->acc.alloc(r->capacity * K));
        if (r->flags & YMC_RECORD_LOGPOST) YMC_TRY(r->lp_out.alloc(r->capacity * K * sizeof(double)));
        if ((r->flags & YMC_RECORD_SUB_ACCEPT) &&
            (problem->method == YMC_METHOD_MLDA2 || problem->method == YMC_METHOD_AEM2))
            YMC_TRY(r->sub_acc.alloc(r->capacity * (uint64_t)S * K));
        if (problem->tile && (problem->method == YMC_METHOD_MLDA2 || problem->method == YMC_METHOD_AEM2))
            YMC_TRY(r->x_tiled.alloc(tiled_row_doubles(r) * sizeof(double)));
        if (problem->tile && problem->method == YMC_METHOD_MLDA_NESTED) /* the parked states of levels 1 .. E-1 */
            YMC_TRY(r->x_tiled.alloc((uint64_t)(problem->E - 1) * tiled_row_doubles(r) * sizeof(double)));
        if (problem->tile && problem->chol_frag.p) YMC_TRY(r->w_tiled.alloc(tiled_row_doubles(r) * sizeof(double)));
        if (problem->wide) {
            if (problem->chol_frag.p) YMC_TRY(r->wide_xi.alloc(K * D * sizeof(double))); /* L xi of the dense proposal */
            YMC_TRY(r->wide_y.alloc(K * D * sizeof(double)));
            YMC_TRY(r->wide_yt.alloc(((K + 15) / 16) * (uint64_t)problem->tgt.KS * 64 * sizeof(double)));
            YMC_HIP(hipMemsetAsync(r->wide_yt.p, 0, r->wide_yt.bytes, ctx->stream)); /* the padding stays zero */
            if (problem->method == YMC_METHOD_MLDA_NESTED) {
                const uint64_t per = K * D * sizeof(double) + r->wide_yt.bytes;
                YMC_TRY(r->wide_lvl_x.alloc((uint64_t)(problem->E - 1) * per));
                YMC_HIP(hipMemsetAsync(r->wide_lvl_x.p, 0, r->wide_lvl_x.bytes, ctx->stream)); /* tiled padding zero */
                YMC_TRY(r->wide_lvl_lp.alloc((uint64_t)(problem->E - 1) * 2 * K * sizeof(double)));
                YMC_TRY(r->wide_lpcz.alloc(K * sizeof(double)));
            }
            if (problem->method == YMC_METHOD_MLDA2) {
                YMC_TRY(r->wide_z.alloc(K * D * sizeof(double)));
                YMC_TRY(r->wide_zt.alloc(r->wide_yt.bytes));
                YMC_HIP(hipMemsetAsync(r->wide_zt.p, 0, r->wide_zt.bytes, ctx->stream));
                YMC_TRY(r->wide_lpcz.alloc(K * sizeof(double)));
            }
            YMC_TRY(r->wide_u.alloc(K * sizeof(double)));
            YMC_TRY(r->wide_lpy.alloc(K * sizeof(double)));
            YMC_TRY(r->wide_flag.alloc(K));
            {   /* the largest output count among the levels sizes the small-ensemble scratch */
                int mt_max = problem->tgt.MT;
                if (problem->method == YMC_METHOD_MLDA2) mt_max = std::max(mt_max, problem->sur.MT);
                if (problem->method == YMC_METHOD_MLDA_NESTED)
                    for (int e = 0; e < problem->E - 1; ++e) mt_max = std::max(mt_max, std::max(problem->lvlP[e].MT, problem->lvlQ[e + 1].MT));
                const size_t terms = ymc_wide_eval_terms(mt_max, K);
                if (terms) YMC_TRY(r->wide_terms.alloc(terms * sizeof(double)));
            }
            YMC_TRY(r->wide_differs.alloc(K * sizeof(uint32_t)));
            YMC_HIP(hipMemsetAsync(r->wide_differs.p, 0, r->wide_differs.bytes, ctx->stream));
        }
        if (problem->method == YMC_METHOD_AEM2 && problem->tile) {
            /* per-output arrays in the kernel's tiled layout [ceil(K/16)][4 MT][64]; ids of the cached points */
            const uint64_t per_out = ((K + 15) / 16) * 4ull * (uint64_t)problem->sur.MT * 64ull;
            YMC_TRY(r->err_mean_alt.alloc(per_out * sizeof(double)));
            YMC_TRY(r->err_m2_alt.alloc(per_out * sizeof(double)));
            DeviceBuf *bufs[] = {&r->err_n, &r->err_mean, &r->err_m2, &r->aem_prec, &r->aem_flags, &r->cache_key,
                                 &r->cache_val};
            const size_t sizes[] = {K * sizeof(uint64_t), per_out * sizeof(double), per_out * sizeof(double),
                                    per_out * sizeof(double), K * 2 * sizeof(int32_t), K * 4 * sizeof(uint64_t),
                                    K * YMC_AEM_CACHE * sizeof(double)};
            for (int i = 0; i < 7; ++i) {
                YMC_TRY(bufs[i]->alloc(sizes[i]));
                YMC_HIP(hipMemsetAsync(bufs[i]->p, 0, sizes[i], ctx->stream));
            }
        } else if (problem->method == YMC_METHOD_AEM2) {
            const uint64_t m = (uint64_t)problem->sur.d.data_dim;
            DeviceBuf *bufs[] = {&r->err_n, &r->err_mean, &r->err_m2, &r->aem_prec, &r->aem_flags, &r->cache_key,
                                 &r->cache_val};
            const size_t sizes[] = {K * sizeof(uint64_t), K * m * sizeof(double), K * m * sizeof(double),
                                    K * m * sizeof(double), K * 2 * sizeof(int32_t),
                                    K * YMC_AEM_CACHE * D * sizeof(double), K * YMC_AEM_CACHE * sizeof(double)};
            for (int i = 0; i < 7; ++i) {
                YMC_TRY(bufs[i]->alloc(sizes[i]));
                YMC_HIP(hipMemsetAsync(bufs[i]->p, 0, sizes[i], ctx->stream));
            }
        }
        /* logpi of the initial states (metropolisHastings.py:110 + first evaluate_log of mrw.py:54) */
        YMC_TRY(eval_logdensity(ctx, problem, problem->tgt, K, r->x.as<double>(), r->lp.as<double>(), nullptr));
        if (problem->method == YMC_METHOD_MLDA2)
            YMC_TRY(eval_logdensity(ctx, problem, problem->sur, K, r->x.as<double>(), r->lpc.as<double>(), nullptr));
        if (problem->method == YMC_METHOD_MLDA_NESTED) {
            YMC_TRY(eval_logdensity(ctx, problem, problem->lvlQ[problem->E - 1], K, r->x.as<double>(),
                                    r->lpc.as<double>(), nullptr));
            YMC_TRY(r->level_acc.alloc((uint64_t)(problem->E - 1) * K * sizeof(uint64_t)));
            YMC_HIP(hipMemsetAsync(r->level_acc.p, 0, r->level_acc.bytes, ctx->stream));
        }
        YMC_HIP(hipStreamSynchronize(ctx->stream));
        return YMC_OK;
    };
    int rc = body();
    if (rc != YMC_OK) {
        delete r;
        return rc;
    }
    problem->users += 1;
    *out = r;
    return YMC_OK;
}

int ymc_run_destroy(ymc_run *run)
{
    if (!run) return YMC_OK;
    (void)hipSetDevice(run->ctx->device);
    (void)hipStreamSynchronize(run->ctx->stream);
    if (run->copy_stream) {
        (void)hipStreamSynchronize(run->copy_stream);
        (void)hipStreamDestroy(run->copy_stream);
    }
    for (auto &slot : run->egress) {
        if (slot.ready) (void)hipEventDestroy(slot.ready);
        if (slot.done) (void)hipEventDestroy(slot.done);
    }
    for (auto &v : {run->timed, run->pool})
        for (auto &ev : v) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    const ymc_problem *problem = run->problem;
    delete run;
    if (--problem->users == 0 && problem->released) problem_free(problem);
    return YMC_OK;
}

int ymc_run_advance(ymc_run *run, uint64_t nsteps)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    if (nsteps == 0) return YMC_OK;
    /* rows this call adds: the transitions t (0-based, counted over the run) with (t + 1) % stride == 0 */
    const uint64_t new_rows = (run->steps_done + nsteps) / run->stride - run->steps_done / run->stride;
    if (records_anything(run->flags) && run->recorded + new_rows > run->capacity)
        return fail(YMC_ERR_CAPACITY, "record holds %llu of %llu rows; %llu more do not fit",
                    (unsigned long long)run->recorded, (unsigned long long)run->capacity,
                    (unsigned long long)new_rows);
    YMC_HIP(hipSetDevice(run->ctx->device));
    /* rows about to be written must not be on their way to the host any more */
    for (auto &slot : run->egress)
        if (slot.pending && run->recorded < slot.first + slot.rows && slot.first < run->recorded + new_rows)
            YMC_HIP(hipStreamWaitEvent(run->ctx->stream, slot.done, 0));
    const uint64_t chunk_max = 1u << 24;
    while (nsteps) {
        const uint64_t chunk = nsteps < chunk_max ? nsteps : chunk_max;
        if (run->problem->tile) {
            YMC_TRY(launch_advance_tile(run, chunk));
        } else if (run->problem->wide && run->problem->method == YMC_METHOD_MLDA_NESTED) {
            YMC_TRY(launch_advance_wide_nested(run, chunk));
        } else if (run->problem->wide) {
            YMC_TRY(launch_advance_wide(run, chunk));
        } else if (run->problem->method == YMC_METHOD_MLDA_NESTED) {
            YMC_DISPATCH_D(run->problem->D, YMC_TRY(launch_advance_nested<DD>(run, chunk)));
        } else if (run->problem->method == YMC_METHOD_AEM2) {
            YMC_DISPATCH_D(run->problem->D, YMC_TRY(launch_advance_aem<DD>(run, chunk)));
        } else {
            YMC_DISPATCH_D(run->problem->D, YMC_TRY(launch_advance<DD>(run, chunk)));
        }
        if (records_anything(run->flags))
            run->recorded += (run->steps_done + chunk) / run->stride - run->steps_done / run->stride;
        run->steps_done += chunk;
        nsteps -= chunk;
    }
    return YMC_OK;
}

int ymc_run_sync(ymc_run *run)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    YMC_HIP(hipSetDevice(run->ctx->device));
    YMC_HIP(hipStreamSynchronize(run->ctx->stream));
    return check_internal_faults(run);
}

int ymc_run_kernel_time_ms(ymc_run *run, double *avg_ms, uint64_t *launches)
{
    if (!run || !avg_ms) return fail(YMC_ERR_INVALID, "null argument");
    YMC_HIP(hipSetDevice(run->ctx->device));
    YMC_HIP(hipStreamSynchronize(run->ctx->stream));
    double total = 0.0;
    for (auto &ev : run->timed) {
        float ms = 0.f;
        YMC_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
        total += ms;
    }
    const uint64_t n = run->timed.size();
    *avg_ms = n ? total / (double)n : 0.0;
    if (launches) *launches = n;
    run->pool.insert(run->pool.end(), run->timed.begin(), run->timed.end());
    run->timed.clear();
    return YMC_OK;
}

int ymc_run_rewind_record(ymc_run *run)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    run->recorded = 0;
    return YMC_OK;
}

int ymc_run_steps_done(const ymc_run *run, uint64_t *total, uint64_t *recorded)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    if (total) *total = run->steps_done;
    if (recorded) *recorded = run->recorded;
    return YMC_OK;
}

int ymc_run_fetch_trajectory(ymc_run *run, uint64_t first, uint64_t n, double *dst, size_t dst_bytes)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    if (run->traj_tiled()) {
        /* the record is in the kernel's tiled layout: untile on the device, chunk by chunk */
        if (!dst) return fail(YMC_ERR_INVALID, "null argument");
        if (!(run->flags & YMC_RECORD_TRAJECTORY)) return fail(YMC_ERR_INVALID, "the run does not record the trajectory");
        if (first > run->recorded || n > run->recorded - first)
            return fail(YMC_ERR_INVALID, "rows beyond the record requested");
        const uint64_t row_doubles = run->K * (uint64_t)run->problem->D;
        if (dst_bytes < n * row_doubles * sizeof(double)) return fail(YMC_ERR_INVALID, "destination too small");
        YMC_HIP(hipSetDevice(run->ctx->device));
        if (run->fault.p) { /* an internal device error is never delivered as data (fetch_rows does the same) */
            YMC_HIP(hipStreamSynchronize(run->ctx->stream));
            YMC_TRY(check_internal_faults(run));
        }
        const uint64_t chunk_rows = std::max<uint64_t>(1, (256ull << 20) / (row_doubles * sizeof(double)));
        DeviceBuf stage;
        YMC_TRY(stage.alloc(std::min(chunk_rows, std::max<uint64_t>(n, 1)) * row_doubles * sizeof(double)));
        for (uint64_t r0 = 0; r0 < n; r0 += chunk_rows) {
            const uint64_t rows = std::min(chunk_rows, n - r0);
            YMC_HIP(ymc_launch_untile(run->traj.as<double>(), first + r0, rows, run->K, run->problem->D,
                                      run->problem->tgt.KS, stage.as<double>(), run->ctx->stream));
            YMC_HIP(hipMemcpyAsync(dst + r0 * row_doubles, stage.p, rows * row_doubles * sizeof(double),
                                   hipMemcpyDeviceToHost, run->ctx->stream));
            YMC_HIP(hipStreamSynchronize(run->ctx->stream));
        }
        return YMC_OK;
    }
    return fetch_rows(run, run->traj, YMC_RECORD_TRAJECTORY, "the trajectory", first, n,
                      run->K * run->problem->D * sizeof(double), dst, dst_bytes);
}

int ymc_run_fetch_accept(ymc_run *run, uint64_t first, uint64_t n, uint8_t *dst, size_t dst_bytes)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    return fetch_rows(run, run->acc, YMC_RECORD_ACCEPT, "accept decisions", first, n, run->K, dst, dst_bytes);
}

int ymc_run_fetch_logpost(ymc_run *run, uint64_t first, uint64_t n, double *dst, size_t dst_bytes)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    return fetch_rows(run, run->lp_out, YMC_RECORD_LOGPOST, "log-densities", first, n, run->K * sizeof(double), dst,
                      dst_bytes);
}

int ymc_run_fetch_sub_accept(ymc_run *run, uint64_t first, uint64_t n, uint8_t *dst, size_t dst_bytes)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    if (run->problem->method != YMC_METHOD_MLDA2 && run->problem->method != YMC_METHOD_AEM2)
        return fail(YMC_ERR_INVALID, "not a two-level run");
    return fetch_rows(run, run->sub_acc, YMC_RECORD_SUB_ACCEPT, "sub-chain decisions", first, n,
                      (uint64_t)run->problem->S * run->K, dst, dst_bytes);
}

int ymc_run_fetch_error_model(ymc_run *run, uint64_t *n, double *mean, double *m2, double *precision,
                              int32_t *has_precision)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    if (run->problem->method != YMC_METHOD_AEM2) return fail(YMC_ERR_INVALID, "not an adaptive-error-model run");
    YMC_HIP(hipSetDevice(run->ctx->device));
    YMC_HIP(hipStreamSynchronize(run->ctx->stream));
    if (n) YMC_HIP(hipMemcpy(n, run->err_n.p, run->err_n.bytes, hipMemcpyDeviceToHost));
    if (run->problem->tile) {
        /* the matrix-core kernel keeps the per-output arrays in its own layout [tile][4 MT][64] */
        const uint64_t K = run->K, m = (uint64_t)run->problem->sur.d.data_dim, Q = 4ull * run->problem->sur.MT;
        std::vector<double> tiled(run->err_mean.bytes / sizeof(double));
        auto untile = [&](const DeviceBuf &src, double *dst) -> int {
            if (!dst) return YMC_OK;
            YMC_HIP(hipMemcpy(tiled.data(), src.p, src.bytes, hipMemcpyDeviceToHost));
            for (uint64_t k = 0; k < K; ++k)
                for (uint64_t j = 0; j < m; ++j) {
                    const uint64_t q = 4 * (j / 16) + (j % 16) / 4, lane = 16 * (j % 4) + k % 16;
                    dst[k * m + j] = tiled[((k / 16) * Q + q) * 64 + lane];
                }
            return YMC_OK;
        };
        YMC_TRY(untile(run->err_mean, mean));
        YMC_TRY(untile(run->err_m2, m2));
        YMC_TRY(untile(run->aem_prec, precision));
    } else {
        if (mean) YMC_HIP(hipMemcpy(mean, run->err_mean.p, run->err_mean.bytes, hipMemcpyDeviceToHost));
        if (m2) YMC_HIP(hipMemcpy(m2, run->err_m2.p, run->err_m2.bytes, hipMemcpyDeviceToHost));
        if (precision) YMC_HIP(hipMemcpy(precision, run->aem_prec.p, run->aem_prec.bytes, hipMemcpyDeviceToHost));
    }
    if (has_precision) {
        std::vector<int32_t> flags(2 * run->K);
        YMC_HIP(hipMemcpy(flags.data(), run->aem_flags.p, run->aem_flags.bytes, hipMemcpyDeviceToHost));
        for (uint64_t k = 0; k < run->K; ++k) has_precision[k] = flags[2 * k];
    }
    return YMC_OK;
}

int ymc_run_fetch_level_accept(ymc_run *run, uint64_t *dst, size_t dst_bytes, uint64_t *made)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    const ymc_problem *p = run->problem;
    if (p->method != YMC_METHOD_MLDA_NESTED) return fail(YMC_ERR_INVALID, "not a nested run");
    YMC_HIP(hipSetDevice(run->ctx->device));
    YMC_HIP(hipStreamSynchronize(run->ctx->stream));
    if (dst) {
        if (dst_bytes < run->level_acc.bytes)
            return fail(YMC_ERR_INVALID, "destination holds %zu bytes, %zu needed", dst_bytes, run->level_acc.bytes);
        YMC_HIP(hipMemcpy(dst, run->level_acc.p, run->level_acc.bytes, hipMemcpyDeviceToHost));
    }
    if (made) {
        /* level e makes lvl_steps[e] transitions per step of level e+1 */
        uint64_t above = run->steps_done;
        for (int e = p->E - 2; e >= 0; --e) {
            above *= (uint64_t)p->lvl_steps[e];
            made[e] = above;
        }
    }
    return YMC_OK;
}

int ymc_run_fetch_state(ymc_run *run, double *state, double *logpost)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    YMC_HIP(hipSetDevice(run->ctx->device));
    YMC_HIP(hipStreamSynchronize(run->ctx->stream));
    YMC_TRY(check_internal_faults(run));
    if (state) YMC_HIP(hipMemcpy(state, run->x.p, run->x.bytes, hipMemcpyDeviceToHost));
    if (logpost) YMC_HIP(hipMemcpy(logpost, run->lp.p, run->lp.bytes, hipMemcpyDeviceToHost));
    return YMC_OK;
}

int ymc_run_fetch_summary(ymc_run *run, uint64_t *accepted, uint64_t *wf_n, double *wf_mean, double *wf_m2,
                          uint32_t *status)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    YMC_HIP(hipSetDevice(run->ctx->device));
    YMC_HIP(hipStreamSynchronize(run->ctx->stream));
    if (accepted) YMC_HIP(hipMemcpy(accepted, run->acc_count.p, run->acc_count.bytes, hipMemcpyDeviceToHost));
    if (wf_n) {
        const uint64_t n = (run->flags & YMC_RECORD_MOMENTS) ? run->steps_done : 0;
        for (uint64_t k = 0; k < run->K; ++k) wf_n[k] = n;
    }
    if (wf_mean) YMC_HIP(hipMemcpy(wf_mean, run->wf_mean.p, run->wf_mean.bytes, hipMemcpyDeviceToHost));
    if (wf_m2) YMC_HIP(hipMemcpy(wf_m2, run->wf_m2.p, run->wf_m2.bytes, hipMemcpyDeviceToHost));
    if (status) YMC_HIP(hipMemcpy(status, run->status.p, run->status.bytes, hipMemcpyDeviceToHost));
    return YMC_OK;
}

int ymc_run_egress_begin(ymc_run *run, int slot_index, uint64_t first, uint64_t n, uint32_t what)
{
    if (!run) return fail(YMC_ERR_INVALID, "null argument");
    if (slot_index < 0 || slot_index >= YMC_EGRESS_SLOTS)
        return fail(YMC_ERR_INVALID, "egress slot %d out of range (0 .. %d)", slot_index, YMC_EGRESS_SLOTS - 1);
    const uint32_t known = YMC_RECORD_TRAJECTORY | YMC_RECORD_ACCEPT | YMC_RECORD_LOGPOST;
    if (what == 0 || (what & ~known)) return fail(YMC_ERR_INVALID, "egress of record kinds 0x%x is not possible", what);
    if (what & ~run->flags) return fail(YMC_ERR_INVALID, "the run does not record what was asked for (0x%x of 0x%x)", what, run->flags);
    if (n == 0 || first > run->recorded || n > run->recorded - first)
        return fail(YMC_ERR_INVALID, "rows [%llu, %llu) requested but only %llu recorded", (unsigned long long)first,
                    (unsigned long long)(first + n), (unsigned long long)run->recorded);
    YMC_HIP(hipSetDevice(run->ctx->device));
    ymc_run::EgressSlot &slot = run->egress[slot_index];
    if (!run->copy_stream) YMC_HIP(hipStreamCreateWithFlags(&run->copy_stream, hipStreamNonBlocking));
    if (!slot.ready) YMC_HIP(hipEventCreateWithFlags(&slot.ready, hipEventDisableTiming));
    if (!slot.done) YMC_HIP(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    if (slot.pending) YMC_HIP(hipEventSynchronize(slot.done)); /* the buffers are about to be reused */
    const uint64_t K = run->K;
    const int D = run->problem->D;
    const size_t traj_row = K * (size_t)D * sizeof(double);
    if (what & YMC_RECORD_TRAJECTORY) YMC_TRY(slot.traj.reserve(n * traj_row));
    if (what & YMC_RECORD_ACCEPT) YMC_TRY(slot.acc.reserve(n * K));
    if (what & YMC_RECORD_LOGPOST) YMC_TRY(slot.lp.reserve(n * K * sizeof(double)));
    const double *traj_src = nullptr;
    if (what & YMC_RECORD_TRAJECTORY) {
        if (run->traj_tiled()) {
            if (slot.untiled.bytes < n * traj_row) YMC_TRY(slot.untiled.alloc(n * traj_row));
            YMC_HIP(ymc_launch_untile(run->traj.as<double>(), first, n, K, D, run->problem->tgt.KS,
                                      slot.untiled.as<double>(), run->ctx->stream));
            traj_src = slot.untiled.as<double>();
        } else {
            traj_src = run->traj.as<double>() + first * K * D;
        }
    }
    YMC_HIP(hipEventRecord(slot.ready, run->ctx->stream));
    YMC_HIP(hipStreamWaitEvent(run->copy_stream, slot.ready, 0));
    if (what & YMC_RECORD_TRAJECTORY)
        YMC_HIP(hipMemcpyAsync(slot.traj.p, traj_src, n * traj_row, hipMemcpyDeviceToHost, run->copy_stream));
    if (what & YMC_RECORD_ACCEPT)
        YMC_HIP(hipMemcpyAsync(slot.acc.p, run->acc.as<uint8_t>() + first * K, n * K, hipMemcpyDeviceToHost, run->copy_stream));
    if (what & YMC_RECORD_LOGPOST)
        YMC_HIP(hipMemcpyAsync(slot.lp.p, run->lp_out.as<double>() + first * K, n * K * sizeof(double),
                               hipMemcpyDeviceToHost, run->copy_stream));
    YMC_HIP(hipEventRecord(slot.done, run->copy_stream));
    slot.pending = true;
    slot.first = first;
    slot.rows = n;
    slot.what = what;
    return YMC_OK;
}

int ymc_run_egress_wait(ymc_run *run, int slot_index, ymc_egress_view *out)
{
    if (!run || !out) return fail(YMC_ERR_INVALID, "null argument");
    if (slot_index < 0 || slot_index >= YMC_EGRESS_SLOTS)
        return fail(YMC_ERR_INVALID, "egress slot %d out of range (0 .. %d)", slot_index, YMC_EGRESS_SLOTS - 1);
    ymc_run::EgressSlot &slot = run->egress[slot_index];
    if (!slot.done || slot.rows == 0) return fail(YMC_ERR_INVALID, "nothing was begun on egress slot %d", slot_index);
    YMC_HIP(hipSetDevice(run->ctx->device));
    YMC_HIP(hipEventSynchronize(slot.done));
    slot.pending = false;
    YMC_TRY(check_internal_faults(run));
    out->trajectory = (slot.what & YMC_RECORD_TRAJECTORY) ? static_cast<const double *>(slot.traj.p) : nullptr;
    out->accept = (slot.what & YMC_RECORD_ACCEPT) ? static_cast<const uint8_t *>(slot.acc.p) : nullptr;
    out->logpost = (slot.what & YMC_RECORD_LOGPOST) ? static_cast<const double *>(slot.lp.p) : nullptr;
    out->first = slot.first;
    out->rows = slot.rows;
    return YMC_OK;
}

int ymc_run_fetch_packed_summary(ymc_run *run, uint64_t *dst, size_t dst_bytes)
{
    if (!run || !dst) return fail(YMC_ERR_INVALID, "null argument");
    uint64_t *packed = nullptr, words = 0;
    hipStream_t stream = nullptr;
    int device = 0;
    YMC_TRY(ymc_internal_pack_summary(run, &packed, &words, &stream, &device));
    const size_t total = (size_t)run->K * words * sizeof(uint64_t);
    if (dst_bytes < total) return fail(YMC_ERR_INVALID, "destination holds %zu bytes, %zu needed", dst_bytes, total);
    YMC_HIP(hipMemcpyAsync(dst, packed, total, hipMemcpyDeviceToHost, stream));
    YMC_HIP(hipStreamSynchronize(stream));
    return YMC_OK;
}

int ymc_logdensity_eval(ymc_ctx *ctx, const ymc_problem *problem, int which, uint64_t n, const double *x,
                        double *out, uint32_t *status)
{
    if (!ctx || !problem || !x || !out) return fail(YMC_ERR_INVALID, "null argument");
    if (which != 0 && !(which == 1 && (problem->method == YMC_METHOD_MLDA2 || problem->method == YMC_METHOD_AEM2)))
        return fail(YMC_ERR_INVALID, "density %d does not exist in this problem", which);
    if (n == 0) return YMC_OK;
    YMC_HIP(hipSetDevice(ctx->device));
    const int D = problem->D;
    DeviceBuf xd, od, sd;
    YMC_TRY(xd.alloc(n * D * sizeof(double)));
    YMC_TRY(od.alloc(n * sizeof(double)));
    YMC_TRY(sd.alloc(n * sizeof(uint32_t)));
    YMC_HIP(hipMemcpyAsync(xd.p, x, xd.bytes, hipMemcpyHostToDevice, ctx->stream));
    YMC_TRY(eval_logdensity(ctx, problem, which ? problem->sur : problem->tgt, n, xd.as<double>(), od.as<double>(),
                            sd.as<uint32_t>()));
    YMC_HIP(hipMemcpyAsync(out, od.p, od.bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (status) YMC_HIP(hipMemcpyAsync(status, sd.p, sd.bytes, hipMemcpyDeviceToHost, ctx->stream));
    YMC_HIP(hipStreamSynchronize(ctx->stream));
    return YMC_OK;
}

int ymc_forward_eval(ymc_ctx *ctx, const ymc_problem *problem, int which, uint64_t n, const double *x, double *out)
{
    if (!ctx || !problem || !x || !out) return fail(YMC_ERR_INVALID, "null argument");
    if (which != 0 && !(which == 1 && (problem->method == YMC_METHOD_MLDA2 || problem->method == YMC_METHOD_AEM2)))
        return fail(YMC_ERR_INVALID, "density %d does not exist in this problem", which);
    const DensityHost &h = which ? problem->sur : problem->tgt;
    uint64_t m = 0;
    if (problem->kind == YMC_DENSITY_POSTERIOR_LINEAR) m = h.m_raw ? h.m_raw : h.d.data_dim;
    else if (problem->kind == YMC_DENSITY_POSTERIOR_LV) m = 2ull * h.d.n_data;
    else if (const ymc_ext_kind_info *ext = ymc_ext_kind(problem->kind)) m = (uint64_t)ext->outputs * (uint64_t)h.d.n_data;
    else return fail(YMC_ERR_INVALID, "this density has no forward model");
    if (n == 0) return YMC_OK;
    YMC_HIP(hipSetDevice(ctx->device));
    const int D = problem->D;
    DeviceBuf xd, od;
    YMC_TRY(xd.alloc(n * D * sizeof(double)));
    YMC_TRY(od.alloc(n * m * sizeof(double)));
    YMC_HIP(hipMemcpyAsync(xd.p, x, xd.bytes, hipMemcpyHostToDevice, ctx->stream));
    if (h.tile || h.wide) {
        YMC_HIP(ymc_launch_linear_big_eval(big_view(h, true), n, xd.as<double>(), nullptr, od.as<double>(), ctx->stream));
    } else {
        YMC_DISPATCH_D(D, YMC_HIP(ymc_launch_forward<DD>(problem->kind, device_view<DD>(h), n, xd.as<double>(),
                                                          od.as<double>(), ctx->stream)));
    }
    YMC_HIP(hipMemcpyAsync(out, od.p, od.bytes, hipMemcpyDeviceToHost, ctx->stream));
    YMC_HIP(hipStreamSynchronize(ctx->stream));
    return YMC_OK;
}

} /* extern "C" */

namespace {

/* IAT of every chain from device-resident states traj[rows][K][d], rows [first, first + n) */
int iat_from_device_states(ymc_ctx *ctx, const double *traj, uint64_t first, uint64_t n, uint64_t K, int d, int method,
                           double sokal_const, int32_t *iat, int tile_KS = 0)
{
    if (method != 0 && method != 1)
        return fail(YMC_ERR_INVALID, "Invalid IAT - Type: %d. Options are 'mean' (0) and 'max' (1).", method);
    if (!(sokal_const > 0.0)) return fail(YMC_ERR_INVALID, "the Sokal constant must be positive");
    if (n < 2) return fail(YMC_ERR_INVALID, "a series needs at least two entries");
    const uint64_t per_chain = method == 0 ? 1 : (uint64_t)d, S = K * per_chain;
    DeviceBuf &series = ctx->post_series, &out = ctx->post_out;
    if (out.bytes < S * sizeof(int32_t)) YMC_TRY(out.alloc(S * sizeof(int32_t)));
    hipStream_t st = ctx->stream;
    if (series.bytes < S * n * sizeof(double)) YMC_TRY(series.alloc(S * n * sizeof(double)));
    YMC_HIP(ymc_launch_series_from_record(traj, first, n, K, d, tile_KS, method == 0, series.as<double>(), st));
    YMC_HIP(ymc_launch_iat(series.as<double>(), S, n, sokal_const, out.as<int32_t>(), nullptr, st));
    std::vector<int32_t> host(S);
    YMC_HIP(hipMemcpyAsync(host.data(), out.p, S * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    YMC_HIP(hipStreamSynchronize(st));
    for (uint64_t k = 0; k < K; ++k) {
        int32_t best = host[k * per_chain]; /* 'max': autocorrelation.py:131-138 */
        bool undefined = best == YMC_IAT_UNDEFINED;
        for (uint64_t j = 1; j < per_chain; ++j) {
            undefined = undefined || host[k * per_chain + j] == YMC_IAT_UNDEFINED;
            best = std::max(best, host[k * per_chain + j]);
        }
        iat[k] = undefined ? YMC_IAT_UNDEFINED : best;
    }
    return YMC_OK;
}

}  // namespace

extern "C" {

int ymc_run_iat(ymc_run *run, uint64_t first, uint64_t n, int method, double sokal_const, int32_t *iat)
{
    if (!run || !iat) return fail(YMC_ERR_INVALID, "null argument");
    if (!(run->flags & YMC_RECORD_TRAJECTORY)) return fail(YMC_ERR_INVALID, "the run does not record the trajectory");
    if (first > run->recorded || n > run->recorded - first)
        return fail(YMC_ERR_INVALID, "rows [%llu, %llu + %llu) requested but %llu are recorded (burn-in beyond the record?)",
                    (unsigned long long)first, (unsigned long long)first, (unsigned long long)n,
                    (unsigned long long)run->recorded);
    YMC_HIP(hipSetDevice(run->ctx->device));
    if (run->fault.p) {
        YMC_HIP(hipStreamSynchronize(run->ctx->stream));
        YMC_TRY(check_internal_faults(run));
    }
    return iat_from_device_states(run->ctx, run->traj.as<double>(), first, n, run->K, run->problem->D, method,
                                  sokal_const, iat, run->traj_tiled() ? run->problem->tgt.KS : 0);
}

int ymc_iat_states(ymc_ctx *ctx, const double *states, uint64_t n, uint64_t n_chains, int32_t d, int method,
                   double sokal_const, int32_t *iat)
{
    if (!ctx || !states || !iat) return fail(YMC_ERR_INVALID, "null argument");
    if (n_chains == 0 || d < 1) return fail(YMC_ERR_INVALID, "empty states");
    YMC_HIP(hipSetDevice(ctx->device));
    DeviceBuf dev;
    YMC_TRY(dev.alloc(n * n_chains * (uint64_t)d * sizeof(double)));
    YMC_HIP(hipMemcpyAsync(dev.p, states, dev.bytes, hipMemcpyHostToDevice, ctx->stream));
    return iat_from_device_states(ctx, dev.as<double>(), 0, n, n_chains, d, method, sokal_const, iat);
}

int ymc_iat_series(ymc_ctx *ctx, const double *series, uint64_t n_series, uint64_t length, double sokal_const,
                   int32_t *iat, double *acf, uint64_t acf_lags)
{
    if (!ctx || !series || !iat) return fail(YMC_ERR_INVALID, "null argument");
    if (!(sokal_const > 0.0)) return fail(YMC_ERR_INVALID, "the Sokal constant must be positive");
    if (n_series == 0) return YMC_OK;
    if (length < 2) return fail(YMC_ERR_INVALID, "a series needs at least two entries");
    YMC_HIP(hipSetDevice(ctx->device));
    DeviceBuf dev, out, acf_dev;
    YMC_TRY(dev.alloc(n_series * length * sizeof(double)));
    YMC_TRY(out.alloc(n_series * sizeof(int32_t)));
    YMC_HIP(hipMemcpyAsync(dev.p, series, n_series * length * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    YMC_HIP(ymc_launch_iat(dev.as<double>(), n_series, length, sokal_const, out.as<int32_t>(), nullptr, ctx->stream));
    YMC_HIP(hipMemcpyAsync(iat, out.p, out.bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (acf && acf_lags) {
        YMC_TRY(acf_dev.alloc(n_series * acf_lags * sizeof(double)));
        YMC_HIP(hipMemsetAsync(acf_dev.p, 0, acf_dev.bytes, ctx->stream));
        YMC_HIP(ymc_launch_acf(dev.as<double>(), n_series, length, acf_lags, acf_dev.as<double>(), ctx->stream));
        YMC_HIP(hipMemcpyAsync(acf, acf_dev.p, acf_dev.bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    YMC_HIP(hipStreamSynchronize(ctx->stream));
    return YMC_OK;
}

int ymc_rng_fill(ymc_ctx *ctx, uint64_t seed, uint64_t chain0, uint64_t K, uint64_t t0, uint64_t nt, int32_t d,
                 double *xi, double *u)
{
    if (!ctx) return fail(YMC_ERR_INVALID, "null argument");
    if (d < 1) return fail(YMC_ERR_INVALID, "dimension must be positive");
    if (K == 0 || nt == 0) return YMC_OK;
    YMC_HIP(hipSetDevice(ctx->device));
    DeviceBuf xd, ud;
    if (xi) YMC_TRY(xd.alloc(K * nt * (uint64_t)d * sizeof(double)));
    if (u) YMC_TRY(ud.alloc(K * nt * sizeof(double)));
    YMC_HIP(ymc_launch_rng_fill(seed, chain0, K, t0, nt, d, xd.as<double>(), ud.as<double>(), ctx->stream));
    if (xi) YMC_HIP(hipMemcpyAsync(xi, xd.p, xd.bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (u) YMC_HIP(hipMemcpyAsync(u, ud.p, ud.bytes, hipMemcpyDeviceToHost, ctx->stream));
    YMC_HIP(hipStreamSynchronize(ctx->stream));
    return YMC_OK;
}

int ymc_numerics_eval(ymc_ctx *ctx, int fn, uint64_t n, const double *x, double *y0, double *y1)
{
    if (!ctx || !x || !y0) return fail(YMC_ERR_INVALID, "null argument");
    if (fn < 0 || (fn > 4 && fn != 6)) return fail(YMC_ERR_INVALID, "unknown function %d", fn);
    if ((fn == 2 || fn == 4) && !y1) return fail(YMC_ERR_INVALID, "y1 is required for this function");
    if (n == 0) return YMC_OK;
    YMC_HIP(hipSetDevice(ctx->device));
    DeviceBuf xd, ad, bd;
    YMC_TRY(xd.alloc(n * sizeof(double)));
    YMC_TRY(ad.alloc(n * sizeof(double)));
    YMC_TRY(bd.alloc(n * sizeof(double)));
    YMC_HIP(hipMemcpyAsync(xd.p, x, xd.bytes, hipMemcpyHostToDevice, ctx->stream));
    if (fn == 4) YMC_HIP(hipMemcpyAsync(bd.p, y1, bd.bytes, hipMemcpyHostToDevice, ctx->stream));
    YMC_HIP(ymc_launch_numerics(fn, n, xd.as<double>(), ad.as<double>(), bd.as<double>(), ctx->stream));
    YMC_HIP(hipMemcpyAsync(y0, ad.p, ad.bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (fn == 2) YMC_HIP(hipMemcpyAsync(y1, bd.p, bd.bytes, hipMemcpyDeviceToHost, ctx->stream));
    YMC_HIP(hipStreamSynchronize(ctx->stream));
    return YMC_OK;
}

int ymc_summary_words(const ymc_run *run, uint64_t *words)
{
    if (!run || !words) return fail(YMC_ERR_INVALID, "null argument");
    *words = 3ull * run->problem->D + 5ull;
    return YMC_OK;
}

}  // extern "C"

/* used by ymc_comm.hip: pack the per-chain summaries into run->summary */
int ymc_internal_pack_summary(ymc_run *run, uint64_t **dev_out, uint64_t *words, hipStream_t *stream, int *device)
{
    const int D = run->problem->D;
    const uint64_t w = 3ull * D + 5ull;
    YMC_HIP(hipSetDevice(run->ctx->device));
    if (run->summary.bytes < run->K * w * sizeof(uint64_t)) YMC_TRY(run->summary.alloc(run->K * w * sizeof(uint64_t)));
    const uint64_t wf_n = (run->flags & YMC_RECORD_MOMENTS) ? run->steps_done : 0;
    YMC_HIP(ymc_launch_pack_summary(run->K, D, wf_n, run->x.as<double>(), run->lp.as<double>(),
                                    run->acc_count.as<uint64_t>(), run->wf_mean.as<double>(),
                                    run->wf_m2.as<double>(), run->status.as<uint32_t>(),
                                    run->summary.as<uint64_t>(), run->ctx->stream));
    *dev_out = run->summary.as<uint64_t>();
    *words = w;
    *stream = run->ctx->stream;
    *device = run->ctx->device;
    return YMC_OK;
}

int ymc_internal_fail(int code, const char *msg) { return fail(code, "%s", msg); }
int ymc_internal_ctx_device(const ymc_ctx *ctx, hipStream_t *stream)
{
    *stream = ctx->stream;
    return ctx->device;
}
uint64_t ymc_internal_run_chains(const ymc_run *run) { return run->K; }
void ymc_internal_ctx_retain(ymc_ctx *ctx) { ctx->users += 1; }
void ymc_internal_ctx_release(ymc_ctx *ctx) { ctx_drop_user(ctx); }
