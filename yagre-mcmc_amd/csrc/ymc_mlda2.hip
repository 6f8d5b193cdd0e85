/*
 * ymc_mlda2.hip -- two-level delayed acceptance (mlda.py:100-110, :146-154) with the sub-chain length S
 * as a COMPILE-TIME parameter, for the lengths the reference's examples and tests use (2, 3, 4, 5, 6, 10)
 * at d <= 2 and the builders' default record (trajectory + decisions, with or without moments).
 *
 * mlda2_kernel (ymc_kernels.hip) serves every S with a run-time inner loop, a cached uniform block behind
 * a branch and `#pragma unroll 1`.  With one or two waves per SIMD (BASELINE configs[4]: 131 072 chains per
 * GPU) every uniform branch is an instruction-buffer bubble, so here a fine step is ONE basic block:
 * the S coarse sub-steps are unrolled, and the accept uniforms come from whole Philox blocks on a fixed
 * schedule -- fine step n owns transitions t = n (S + 1) + s, s = 0 .. S, and transitions 2p, 2p + 1
 * share a block, so the walk advances in units that start at an even t: one fine step when S + 1 is
 * even, two when it is odd (the block of the first step's fine decision is the block of the second
 * step's first sub-step).  A launch that starts on an odd t, or ends half-way through a unit, takes
 * those fine steps with per-transition uniforms.  Same draws, same arithmetic, same order as
 * mlda2_kernel: bit-identical results.
 */
#define YMC_COEFF_HOISTED 1
#include <type_traits>
#include <utility>

#include "ymc_kernels.h"

namespace {

/* f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): unrolling that does not depend on the
 * optimizer's thresholds (a `#pragma unroll` over 22 transitions of this size is declined) */
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int kBlock = 256;
[[maybe_unused]] constexpr unsigned kTA = YMC_RECORD_TRAJECTORY | YMC_RECORD_ACCEPT;
/* thinned record masks (ymc_run_desc.record_stride > 1): trajectory, trajectory + moments, rows by RecordClock */
[[maybe_unused]] constexpr unsigned kThin = 0x100u, kT = YMC_RECORD_TRAJECTORY;

#if defined(YMC_MLDA2_TWO_WAVES)
#define YMC_MLDA2_ATTR __attribute__((amdgpu_waves_per_eu(1, 2)))
#else
#define YMC_MLDA2_ATTR
#endif
template <int KIND, int D, unsigned REC, int S>
__global__ __launch_bounds__(kBlock) YMC_MLDA2_ATTR void mlda2_static_kernel(const ChainArgs<D> a)
{
    ymc_stage_tables();
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= a.K) return;
    const uint32_t chain = (uint32_t)(a.chain0 + k);
    constexpr int C = S + 1;               /* transitions per fine step */
    constexpr int FS = (C & 1) ? 2 : 1;    /* fine steps per unit */
    constexpr bool moments = (REC & YMC_RECORD_MOMENTS) != 0;
    constexpr bool thin = (REC & kThin) != 0;
    RecordClock clk(thin ? a.rec_stride : 1u, a.rec_since);

    double x[D], wmean[D], wm2[D], sigma[D];
    ymc_load_row<D>(a.x, k, x);
#pragma unroll
    for (int j = 0; j < D; ++j) {
        sigma[j] = a.sigma[j];
        wmean[j] = wm2[j] = 0.0;
    }
    double lpf = a.lp[k];
    double lpc = a.lpc[k];
    uint64_t accepted = a.acc_count[k];
    uint32_t st = a.status[k];
    if (moments) {
        ymc_load_row<D>(a.wf_mean, k, wmean);
        ymc_load_row<D>(a.wf_m2, k, wm2);
    }

    /* The two Gaussians' parameters (14 doubles at d = 2) in VGPR pairs: as kernel arguments they sit in SGPRs, and
     * with the draws' constants beside them the unit needs more than the 104 a wave has -- the compiler then parks
     * scalars in VGPR lanes and pays a v_readlane / v_writelane per use inside the unit (95 of its 2 400 vector
     * instructions).  The kernel has the vector registers to spare (two waves per SIMD). */
    DensityDev<D> sur = a.sur, tgt = a.tgt;
    if constexpr (KIND == YMC_DENSITY_GAUSS_WHITENED) {
        auto pin = [](double &v) { asm volatile("" : "+v"(v)); };
#pragma unroll
        for (int j = 0; j < D; ++j) {
            pin(sur.mean[j]);
            pin(tgt.mean[j]);
        }
#pragma unroll
        for (int j = 0; j < D * D; ++j) {
            pin(sur.whiten[j]);
            pin(tgt.whiten[j]);
        }
        pin(sur.log_norm);
        pin(tgt.log_norm);
    }
    /* surrogate.run(S + 1, state), one sub-step: a random-walk transition on the coarse density */
    auto coarse = [&](uint64_t t, double u, double (&z)[D], double &lpcz) {
        double y[D];
        ymc_propose<D>(z, sigma, a.seed, chain, t, y);
        const bool moved = !ymc_states_equal<D>(y, z, YMC_EQ_EXACT);
        const double lpcy = Density<KIND, D>::eval(sur, y, st);
        const bool acc = ymc_accept(lpcy - lpcz, moved, ymc_finite(lpcy), u, st);
#pragma unroll
        for (int j = 0; j < D; ++j) z[j] = acc ? y[j] : z[j];
        lpcz = acc ? lpcy : lpcz;
    };
    /* fine accept / reject (mlda.py:146-154), diagnostics and record of fine step n */
    auto fine = [&](uint32_t n, double u, const double (&z)[D], double lpcz) {
        const bool moved = !ymc_states_equal<D>(z, x, YMC_EQ_EXACT);
        const double lpfy = Density<KIND, D>::eval(tgt, z, st);
        const double log_ratio = ((lpfy + lpc) - lpcz) - lpf;
        const bool acc = ymc_accept(log_ratio, moved, ymc_finite(lpfy) && ymc_finite(lpcz), u, st);
        if (moments) ymc_welford<D>(x, (double)(a.wf_n0 + n + 1), ymc_recip_at(a.recip, n), wmean, wm2);
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = acc ? z[j] : x[j];
        lpf = acc ? lpfy : lpf;
        lpc = acc ? lpcz : lpc;
        accepted += acc ? 1u : 0u;
        if constexpr (thin) {
            if (clk.due()) {
                ymc_store_row<D>(a.traj + (uint64_t)clk.row * a.K * D, k, x);
                clk.wrote();
            }
        } else {
            ymc_store_row<D>(a.traj + (uint64_t)n * a.K * D, k, x);
            a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
        }
    };
    /* one fine step with per-transition uniforms (unit boundaries of a launch) */
    auto single = [&](uint32_t n) {
        const uint64_t tbase = (a.t0 + n) * (uint64_t)C;
        double z[D];
#pragma unroll
        for (int j = 0; j < D; ++j) z[j] = x[j];
        double lpcz = lpc;
#pragma unroll 1
        for (int s = 0; s < S; ++s) coarse(tbase + s, ymc_draw_uniform(a.seed, chain, tbase + s), z, lpcz);
        fine(n, ymc_draw_uniform(a.seed, chain, tbase + S), z, lpcz);
    };
    /* FS fine steps from an even transition: straight-line code, uniforms from whole blocks */
    auto unit = [&](uint32_t n) {
        const uint64_t tunit = (a.t0 + n) * (uint64_t)C;
        ymc_u32x4 ub;
        ub.v[0] = ub.v[1] = ub.v[2] = ub.v[3] = 0u;
        static_for<FS>([&](auto f_) {
            constexpr int f = decltype(f_)::value;
            double z[D];
#pragma unroll
            for (int j = 0; j < D; ++j) z[j] = x[j];
            double lpcz = lpc;
            static_for<S + 1>([&](auto s_) {
                constexpr int s = decltype(s_)::value, i = f * C + s;
#if defined(YMC_DRAW_SPEC_V2X)
                const double u = ymc_draw_uniform(a.seed, chain, tunit + i);
#else
                if constexpr ((i & 1) == 0) ub = ymc_uniform_block(a.seed, chain, tunit + i);
                const double u = (i & 1) ? ymc_u01_co(ub.v[2], ub.v[3]) : ymc_u01_co(ub.v[0], ub.v[1]);
#endif
                if constexpr (s < S)
                    coarse(tunit + i, u, z, lpcz);
                else
                    fine(n + f, u, z, lpcz);
            });
        });
    };

    uint32_t n = 0;
    if (FS == 2 && a.nsteps > 0 && (a.t0 & 1u)) { /* C odd: (t0 + n) C is even iff t0 + n is */
        single(0);
        n = 1;
    }
#pragma unroll 1
    for (; n + FS <= a.nsteps; n += FS) unit(n);
    if (n < a.nsteps) single(n);

    ymc_store_row<D>(a.x, k, x);
    a.lp[k] = lpf;
    a.lpc[k] = lpc;
    a.acc_count[k] = accepted;
    a.status[k] = st;
    if (moments) {
        ymc_store_row<D>(a.wf_mean, k, wmean);
        ymc_store_row<D>(a.wf_m2, k, wm2);
    }
}

template <int KIND, int D, unsigned REC>
bool launch_s(const ChainArgs<D> &a, hipStream_t s)
{
    const dim3 grid((unsigned)((a.K + kBlock - 1) / kBlock)), block(kBlock);
    switch (a.S) {
#define YMC_CASE(S_) case S_: hipLaunchKernelGGL((mlda2_static_kernel<KIND, D, REC, S_>), grid, block, 0, s, a); return true;
    YMC_CASE(2) YMC_CASE(3) YMC_CASE(4) YMC_CASE(5) YMC_CASE(6) YMC_CASE(10)
#undef YMC_CASE
    default: return false;
    }
}

/* This file is compiled twice (Makefile): once for the full-rate record masks, once with -DYMC_MLDA2_THIN for
 * the thinned ones -- 120 unrolled kernels in one translation unit take over three minutes to build. */
template <int KIND, int D>
bool launch_rec(unsigned rec, const ChainArgs<D> &a, hipStream_t s)
{
#if defined(YMC_MLDA2_THIN)
    if (rec == kT) return launch_s<KIND, D, kThin | kT>(a, s);
    if (rec == (kT | YMC_RECORD_MOMENTS)) return launch_s<KIND, D, kThin | kT | YMC_RECORD_MOMENTS>(a, s);
#else
    if (rec == kTA) return launch_s<KIND, D, kTA>(a, s);
    if (rec == (kTA | YMC_RECORD_MOMENTS)) return launch_s<KIND, D, kTA | YMC_RECORD_MOMENTS>(a, s);
#endif
    return false;
}

}  // namespace

#if defined(YMC_MLDA2_THIN)
#define YMC_MLDA2_ENTRY ymc_launch_mlda2_static_thin
#else
#define YMC_MLDA2_ENTRY ymc_launch_mlda2_static_full
#endif

/* true: a compile-time-S kernel was launched; false: the caller launches mlda2_kernel */
template <int D>
bool YMC_MLDA2_ENTRY(int kind, unsigned rec, const ChainArgs<D> &a, hipStream_t s)
{
    if constexpr (D > 2) {
        return false;
    } else {
        if (a.eq_mode != YMC_EQ_EXACT) return false;
        switch (kind) {
        case YMC_DENSITY_GAUSS_WHITENED: return launch_rec<YMC_DENSITY_GAUSS_WHITENED, D>(rec, a, s);
        case YMC_DENSITY_POSTERIOR_LINEAR: return launch_rec<YMC_DENSITY_POSTERIOR_LINEAR, D>(rec, a, s);
        case YMC_DENSITY_POSTERIOR_LV:
            if constexpr (D == 2) return launch_rec<YMC_DENSITY_POSTERIOR_LV, 2>(rec, a, s);
            return false;
        default: return false; /* the 1-D scalar target compares states with isclose: run-time kernel */
        }
    }
}

/* ... and each of the two once per dimension with kernels (-DYMC_MLDA2_D=1|2; the d = 1 objects also carry
 * the "not covered" answers of d = 3..8) */
#define YMC_INSTANTIATE(D) template bool YMC_MLDA2_ENTRY<D>(int, unsigned, const ChainArgs<D> &, hipStream_t);
#if YMC_MLDA2_D == 1
YMC_INSTANTIATE(1) YMC_INSTANTIATE(3) YMC_INSTANTIATE(4)
YMC_INSTANTIATE(5) YMC_INSTANTIATE(6) YMC_INSTANTIATE(7) YMC_INSTANTIATE(8)
#elif YMC_MLDA2_D == 2
YMC_INSTANTIATE(2)
#else
#error "compile with -DYMC_MLDA2_D=1 or 2 (see the Makefile)"
#endif
