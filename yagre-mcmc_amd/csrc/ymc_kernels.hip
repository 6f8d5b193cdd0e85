/*
 * ymc_kernels.hip -- gfx950 kernels of the chain-ensemble hot path.
 *
 * mrw_kernel   : MetropolisHastings.run's loop (metropolisHastings.py:112-120)
 *                for the Metropolised random walk (mrw.py:41-57); one lane =
 *                one chain, fused propose / evaluate / accept / record.
 * mlda2_kernel : the same loop for two-level delayed acceptance
 *                (mlda.py:100-110, :146-154).
 * Data layout in HBM (all row-major):
 *   state      x[K][d]           carried between launches
 *   logpost    lp[K] (, lpc[K])  log-density of the carried state
 *   record     traj[T][K][d], acc[T][K] (u8), lp_out[T][K], sub_acc[T][S][K]
 * so one wave writes 64*d*8 contiguous bytes per recorded state (16 B per lane
 * at d = 2) and 64 contiguous accept bytes.  Nothing is read inside the loop.
 */
/* Polynomial coefficients as loop-invariant VGPR pairs (ymc_numerics.h, ymc_fma_c): 11 % fewer
 * instructions per transition than re-materialising them, at the same or better occupancy. */
/* Experiment switches (tools/exp_occupancy.sh; the results are in profiles/r02_exp_occupancy.txt and NOTEBOOK.md (rounds 1-3,
 * section 5): -DYMC_EXP_SGPR keeps the coefficients in SGPR pairs instead, -DYMC_EXP_WAVES=4,4 / 8,8 builds the
 * random-walk kernel for four / eight resident waves per SIMD instead of at most two. */
#if defined(YMC_EXP_SGPR)
#define YMC_COEFF_IN_SGPR 1
#else
#define YMC_COEFF_HOISTED 1
#endif
#ifndef YMC_EXP_WAVES
#define YMC_EXP_WAVES 1, 2
#endif
#include "ymc_kernels.h"
#include "ymc_kinds.h"

/* This file is compiled four times (Makefile, -DYMC_KERNELS_PART=0..3): the dimensions are independent
 * instantiations and one translation unit with all of them takes two minutes to build.  Part 0 also holds the
 * kernels that are not templates on the dimension. */
#ifndef YMC_KERNELS_PART
#error "compile with -DYMC_KERNELS_PART=0..3 (see the Makefile)"
#elif YMC_KERNELS_PART == 0
#define YMC_FOR_EACH_D(X) X(1)
#elif YMC_KERNELS_PART == 1
#define YMC_FOR_EACH_D(X) X(2)
#elif YMC_KERNELS_PART == 2
#define YMC_FOR_EACH_D(X) X(3) X(4) X(5)
#else
#define YMC_FOR_EACH_D(X) X(6) X(7) X(8)
#endif

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(uint64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

/* Record masks and the state-equality rule are COMPILE-TIME parameters of the
 * chain kernels: with one wave per SIMD (BASELINE configs[1]) every uniform
 * branch in the loop costs an instruction-buffer bubble, and the loop body of
 * the common configurations is a single basic block (measured +30 % at 65 536
 * chains against the same kernel with run-time flags).  kRuntimeRec selects
 * the run-time form that serves every other flag combination. */
constexpr unsigned kRuntimeRec = ~0u;
constexpr unsigned kTA = YMC_RECORD_TRAJECTORY | YMC_RECORD_ACCEPT;
/* a compile-time record mask with this bit is THINNED (ymc_run_desc.record_stride > 1): rows are written when
 * the launch's RecordClock says so; the thinned masks are trajectory and trajectory + moments */
constexpr unsigned kThin = 0x100u;
constexpr unsigned kT = YMC_RECORD_TRAJECTORY;

template <unsigned REC>
struct Rec {
    static constexpr bool rt = REC == kRuntimeRec;
    static constexpr bool thin = rt || (REC & kThin) != 0; /* rows follow the RecordClock */
    template <int D>
    static __device__ __forceinline__ bool traj(const ChainArgs<D> &a) { return rt ? a.traj != nullptr : (REC & YMC_RECORD_TRAJECTORY) != 0; }
    template <int D>
    static __device__ __forceinline__ bool acc(const ChainArgs<D> &a) { return rt ? a.acc != nullptr : (REC & YMC_RECORD_ACCEPT) != 0; }
    template <int D>
    static __device__ __forceinline__ bool lp(const ChainArgs<D> &a) { return rt ? a.lp_out != nullptr : (REC & YMC_RECORD_LOGPOST) != 0; }
    template <int D>
    static __device__ __forceinline__ bool sub(const ChainArgs<D> &a) { return rt ? a.sub_acc != nullptr : (REC & YMC_RECORD_SUB_ACCEPT) != 0; }
    template <int D>
    static __device__ __forceinline__ bool moments(const ChainArgs<D> &a) { return rt ? (a.flags & YMC_RECORD_MOMENTS) != 0 : (REC & YMC_RECORD_MOMENTS) != 0; }
};

/* EQ < 0: rule taken from the arguments at run time */
template <int D, int EQ>
__device__ __forceinline__ bool states_equal(const double (&p)[D], const double (&q)[D], int runtime_eq)
{
    return ymc_states_equal<D>(p, q, EQ < 0 ? runtime_eq : EQ);
}

/* PROP: kPropDiag the random walk with a diagonal proposal covariance; kPropPcn the preconditioned
 * Crank-Nicolson proposal (pcn.py), same accept rule with the likelihood as target (pcn.py:51-57);
 * kPropDense the random walk with a dense proposal covariance (covariance.py:69-94) */
constexpr int kPropDiag = 0, kPropPcn = 1, kPropDense = 2;

template <int KIND, int D, unsigned REC, int EQ, int PROP = kPropDiag>
__device__ __forceinline__ void mrw_body(const ChainArgs<D> &a)
{
    ymc_stage_tables();
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= a.K) return;
    const uint32_t chain = (uint32_t)(a.chain0 + k);
    using R = Rec<REC>;

    double x[D], wmean[D], wm2[D], sigma[D];
    ymc_load_row<D>(a.x, k, x);
#pragma unroll
    for (int j = 0; j < D; ++j) sigma[j] = a.sigma[j];
    double lp = a.lp[k];
    uint64_t accepted = a.acc_count[k];
    uint32_t st = a.status[k];
    const bool moments = R::moments(a);
    if (moments) {
        ymc_load_row<D>(a.wf_mean, k, wmean);
        ymc_load_row<D>(a.wf_m2, k, wm2);
    }

    RecordClock clk(R::thin ? a.rec_stride : 1u, a.rec_since);
    /* one transition (metropolisHastings.py:114-120) given its accept uniform */
    auto transition = [&](uint32_t n, double u) {
        const uint64_t t = a.t0 + n;
        double y[D];
        if constexpr (PROP == kPropPcn)
            ymc_propose_pcn<D>(x, sigma, a.tgt.prior_mean, a.pcn_c1, a.pcn_c2, a.seed, chain, t, y);
        else if constexpr (PROP == kPropDense)
            ymc_propose_dense<D>(x, a.chol, a.seed, chain, t, y);
        else
            ymc_propose<D>(x, sigma, a.seed, chain, t, y);                      /* :116-117 */
        const bool moved = !states_equal<D, EQ>(y, x, a.eq_mode);               /* :60-61 */
        const double lpy = Density<KIND, D>::eval(a.tgt, y, st);
        const bool acc = ymc_accept(lpy - lp, moved, ymc_finite(lpy), u, st);   /* mrw.py:51-57, :68-70 */
        if (moments) ymc_welford<D>(x, (double)(a.wf_n0 + n + 1), ymc_recip_at(a.recip, n), wmean, wm2); /* diagnostics.py:91-94 */
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = acc ? y[j] : x[j];                   /* :79-92 */
        lp = acc ? lpy : lp;
        accepted += acc ? 1u : 0u;
        if (R::acc(a)) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;              /* diagnostics.py:48-54 */
        if constexpr (R::thin) {
            if (clk.due()) {
                if (R::traj(a)) ymc_store_row<D>(a.traj + (uint64_t)clk.row * a.K * D, k, x);
                if (R::lp(a)) a.lp_out[(uint64_t)clk.row * a.K + k] = lp;
                clk.wrote();
            }
        } else {
            if (R::traj(a)) ymc_store_row<D>(a.traj + (uint64_t)n * a.K * D, k, x); /* :75-76 */
            if (R::lp(a)) a.lp_out[(uint64_t)n * a.K + k] = lp;
        }
    };

    /* transitions 2p and 2p+1 share one Philox block of uniforms: walk in aligned pairs */
    uint32_t n = 0;
    if (a.nsteps > 0 && (a.t0 & 1u)) {
        transition(0, ymc_draw_uniform(a.seed, chain, a.t0));
        n = 1;
    }
    const uint32_t npairs = (a.nsteps - n) / 2;
#pragma unroll 1
    for (uint32_t i = 0; i < npairs; ++i, n += 2) {
#if defined(YMC_DRAW_SPEC_V2X)
        transition(n, ymc_draw_uniform(a.seed, chain, a.t0 + n));
        transition(n + 1, ymc_draw_uniform(a.seed, chain, a.t0 + n + 1));
#else
        const ymc_u32x4 ub = ymc_uniform_block(a.seed, chain, a.t0 + n);
        transition(n, ymc_u01_co(ub.v[0], ub.v[1]));
        transition(n + 1, ymc_u01_co(ub.v[2], ub.v[3]));
#endif
    }
    if (n < a.nsteps) transition(n, ymc_draw_uniform(a.seed, chain, a.t0 + n));

    ymc_store_row<D>(a.x, k, x);
    a.lp[k] = lp;
    a.acc_count[k] = accepted;
    a.status[k] = st;
    if (moments) {
        ymc_store_row<D>(a.wf_mean, k, wmean);
        ymc_store_row<D>(a.wf_m2, k, wm2);
    }
}

template <int KIND, int D, unsigned REC, int EQ, int PROP = kPropDiag>
__global__ __launch_bounds__(kBlock) void mrw_kernel(const ChainArgs<D> a)
{
    mrw_body<KIND, D, REC, EQ, PROP>(a);
}

/* The specialised record configurations are built for AT MOST two waves per SIMD (BASELINE
 * configs[1], 65 536 chains, is one).  Given the register budget of two waves (256 VGPRs) the
 * scheduler interleaves the independent draw / density chains of a transition pair instead of
 * minimising registers for a third wave: the kernel is VALU-bound, so what counts is that the one
 * or two resident waves never stall on their own latencies.  Same instruction count, 170 instead of
 * 142 VGPRs; measured +8 % at 65 536 chains, +3 % at 131 072, +1 % at 1 M (tools: YMC A/B in NOTEBOOK.md). */
template <int KIND, int D, unsigned REC, int EQ>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(YMC_EXP_WAVES))) void mrw_kernel_two_waves(const ChainArgs<D> a)
{
    mrw_body<KIND, D, REC, EQ, kPropDiag>(a);
}

/*
 * The random walk for ensembles of AT MOST one wave of chains per SIMD (BASELINE configs[1]: 65 536 chains =
 * 1024 waves on 1024 SIMDs).  A single wave issues one vector instruction every ~5.3 cycles whatever the
 * instruction (tools/probe_issue.hip), so with one lane per chain and nothing else on the SIMD the kernel
 * runs at the ISSUE rate of one wave.  Here a 1024-thread workgroup (16 waves: wave w runs on SIMD w % 4)
 * serves 256 chains; chain group g = w & 3 (64 chains) lives on SIMD g:
 *   wave g          the chain itself: y = x + delta, density, accept / reject, moments, record
 *                   (everything that depends on the state: ~75 instructions per transition)
 *   waves 4+g, 8+g, 12+g   the draws, which depend on (chain id, transition) only: producer p makes
 *                   delta = sigma * z and u for transitions 2p, 2p+1 of every block of six (~47 each)
 * handed over through an LDS ring of two blocks, one barrier per six transitions.  Every lane does the
 * arithmetic of mrw_body in the same order: same bits.  Measured +20 % at 65 536 chains (1.43e11 against
 * 1.18e11 transitions/s); what bounds it then is the chip's power limit (the clock drops from 2.39 to
 * ~2.0 GHz), the same wall the one-lane-per-chain kernel meets at 1 M chains (NOTEBOOK.md (rounds 1-3, section 5)).  The
 * consumers of four 256-thread workgroups would all land on SIMD 0 -- hence one big workgroup.
 */
constexpr int kSplitChains = 256;

template <int KIND, int D, unsigned REC, int EQ, int NP>
__global__ __launch_bounds__(256 * (NP + 1)) void mrw_split_kernel(const ChainArgs<D> a)
{
    static_assert(REC != kRuntimeRec, "the split kernel serves the specialised record configurations");
    ymc_stage_tables();
    constexpr int B = 2 * NP;
    __shared__ double ring[4][2][B][D + 1][64];
    using R = Rec<REC>;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = w & 3, role = w >> 2;
    const uint64_t k = (uint64_t)blockIdx.x * kSplitChains + (uint64_t)g * 64 + lane;
    const bool live = k < a.K;
    const uint64_t kk = live ? k : 0;  /* lanes past the ensemble shadow chain 0 and never store */
    const uint32_t chain = (uint32_t)(a.chain0 + kk);
    const uint32_t nb = (a.nsteps + B - 1) / B;

    if (role != 0) {
        /* producer p: transitions 2p and 2p + 1 of every block (draws past nsteps are made and never read) */
        const int p = role - 1;
        double sigma[D];
#pragma unroll
        for (int j = 0; j < D; ++j) sigma[j] = a.sigma[j];
        const bool aligned = (a.t0 & 1u) == 0; /* block starts are even offsets: (2p, 2p+1) share a uniform block */
        auto fill = [&](uint32_t b) {
            const int q = 2 * p;
            const uint64_t t = a.t0 + (uint64_t)b * B + q;
            double u0, u1;
#if defined(YMC_DRAW_SPEC_V2X)
            if (false) {
#else
            if (aligned) {
#endif
                const ymc_u32x4 ub = ymc_uniform_block(a.seed, chain, t);
                u0 = ymc_u01_co(ub.v[0], ub.v[1]);
                u1 = ymc_u01_co(ub.v[2], ub.v[3]);
            } else {
                u0 = ymc_draw_uniform(a.seed, chain, t);
                u1 = ymc_draw_uniform(a.seed, chain, t + 1);
            }
            ring[g][b & 1][q][D][lane] = u0;
            ring[g][b & 1][q + 1][D][lane] = u1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; 2 * j < D; ++j) {
                    double z0, z1;
                    ymc_draw_normal_pair(a.seed, chain, t + h, (uint32_t)j, &z0, &z1);
                    ring[g][b & 1][q + h][2 * j][lane] = sigma[2 * j] * z0;                      /* covariance.py:51-52 */
                    if (2 * j + 1 < D) ring[g][b & 1][q + h][2 * j + 1][lane] = sigma[2 * j + 1] * z1;
                }
            }
        };
        if (nb > 0) fill(0);
        __syncthreads();
#pragma unroll 1
        for (uint32_t b = 0; b < nb; ++b) {
            if (b + 1 < nb) fill(b + 1);
            __syncthreads();
        }
        return;
    }

    double x[D], wmean[D], wm2[D];
    ymc_load_row<D>(a.x, kk, x);
    double lp = a.lp[kk];
    uint64_t accepted = a.acc_count[kk];
    uint32_t st = a.status[kk];
    constexpr bool moments = (REC & YMC_RECORD_MOMENTS) != 0;
#pragma unroll
    for (int j = 0; j < D; ++j) wmean[j] = wm2[j] = 0.0;
    if (moments) {
        ymc_load_row<D>(a.wf_mean, kk, wmean);
        ymc_load_row<D>(a.wf_m2, kk, wm2);
    }
    RecordClock clk(R::thin ? a.rec_stride : 1u, a.rec_since);
    /* one transition (metropolisHastings.py:114-120) given its proposal increment and accept uniform */
    auto transition = [&](uint32_t n, const double (&delta)[D], double u) {
        double y[D];
#pragma unroll
        for (int j = 0; j < D; ++j) y[j] = x[j] + delta[j];                         /* :116-117, gaussian.py:66 */
        const bool moved = !states_equal<D, EQ>(y, x, a.eq_mode);                   /* :60-61 */
        const double lpy = Density<KIND, D>::eval(a.tgt, y, st);
        const bool acc = ymc_accept(lpy - lp, moved, ymc_finite(lpy), u, st);       /* mrw.py:51-57, :68-70 */
        if (moments) ymc_welford<D>(x, (double)(a.wf_n0 + n + 1), ymc_recip_at(a.recip, n), wmean, wm2);
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = acc ? y[j] : x[j];                       /* :79-92 */
        lp = acc ? lpy : lp;
        accepted += acc ? 1u : 0u;
        if constexpr (R::thin) {
            if (clk.due()) {
                if (live && R::traj(a)) ymc_store_row<D>(a.traj + (uint64_t)clk.row * a.K * D, k, x);
                clk.wrote();
            }
        } else if (live) {
            if (R::traj(a)) ymc_store_row<D>(a.traj + (uint64_t)n * a.K * D, k, x); /* :75-76 */
            if (R::acc(a)) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;              /* diagnostics.py:48-54 */
        }
    };
    __syncthreads();
    /* the increments and the uniform of transition s of the current block, read where they are used (the
     * scheduler lifts the reads as far ahead as the register budget of four waves per SIMD allows) */
    auto step = [&](uint32_t b, int s) {
        double delta[D];
#pragma unroll
        for (int j = 0; j < D; ++j) delta[j] = ring[g][b & 1][s][j][lane];
        transition(b * B + s, delta, ring[g][b & 1][s][D][lane]);
    };
#pragma unroll 1
    for (uint32_t b = 0; b < nb; ++b) {
        if (b * B + B <= a.nsteps) {
#pragma unroll
            for (int s = 0; s < B; ++s) step(b, s);
        } else {
#pragma unroll
            for (int s = 0; s < B; ++s)
                if (b * B + s < a.nsteps) step(b, s);
        }
        __syncthreads();
    }
    if (live) {
        ymc_store_row<D>(a.x, k, x);
        a.lp[k] = lp;
        a.acc_count[k] = accepted;
        a.status[k] = st;
        if (moments) {
            ymc_store_row<D>(a.wf_mean, k, wmean);
            ymc_store_row<D>(a.wf_m2, k, wm2);
        }
    }
}

template <int KIND, int D, unsigned REC, int EQ>
__device__ __forceinline__ void mlda2_body(const ChainArgs<D> &a)
{
    ymc_stage_tables();
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= a.K) return;
    const uint32_t chain = (uint32_t)(a.chain0 + k);
    const int S = a.S;
    using R = Rec<REC>;

    double x[D], wmean[D], wm2[D], sigma[D];
    ymc_load_row<D>(a.x, k, x);
#pragma unroll
    for (int j = 0; j < D; ++j) sigma[j] = a.sigma[j];
    double lpf = a.lp[k];
    double lpc = a.lpc[k];
    uint64_t accepted = a.acc_count[k];
    uint32_t st = a.status[k];
    const bool moments = R::moments(a);
    if (moments) {
        ymc_load_row<D>(a.wf_mean, k, wmean);
        ymc_load_row<D>(a.wf_m2, k, wm2);
    }
    UniformCache uc;
    uc.valid = false;
    uc.pair = 0;
    RecordClock clk(R::rt ? a.rec_stride : 1u, a.rec_since);

#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t tbase = (a.t0 + n) * (uint64_t)(S + 1);
        /* surrogate.run(S + 1, state): an S-step random walk on the coarse density */
        double z[D];
#pragma unroll
        for (int j = 0; j < D; ++j) z[j] = x[j];
        double lpcz = lpc;
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
            const uint64_t t = tbase + (uint64_t)s;
            double y[D];
            ymc_propose<D>(z, sigma, a.seed, chain, t, y);
            const double u = ymc_uniform_cached(uc, a.seed, chain, t);
            const bool moved = !states_equal<D, EQ>(y, z, a.eq_mode);
            const double lpcy = Density<KIND, D>::eval(a.sur, y, st);
            const bool acc = ymc_accept(lpcy - lpcz, moved, ymc_finite(lpcy), u, st);
#pragma unroll
            for (int j = 0; j < D; ++j) z[j] = acc ? y[j] : z[j];
            lpcz = acc ? lpcy : lpcz;
            if (R::sub(a)) a.sub_acc[((uint64_t)n * S + s) * a.K + k] = acc ? 1 : 0;
        }
        /* fine accept/reject, mlda.py:146-154 */
        const uint64_t tf = tbase + (uint64_t)S;
        const double u = ymc_uniform_cached(uc, a.seed, chain, tf);
        const bool moved = !states_equal<D, EQ>(z, x, a.eq_mode);
        const double lpfy = Density<KIND, D>::eval(a.tgt, z, st);
        const double log_ratio = ((lpfy + lpc) - lpcz) - lpf;
        const bool acc = ymc_accept(log_ratio, moved, ymc_finite(lpfy) && ymc_finite(lpcz), u, st);
        if (moments) ymc_welford<D>(x, (double)(a.wf_n0 + n + 1), ymc_recip_at(a.recip, n), wmean, wm2);
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = acc ? z[j] : x[j];
        lpf = acc ? lpfy : lpf;
        lpc = acc ? lpcz : lpc;
        accepted += acc ? 1u : 0u;
        if (R::acc(a)) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
        if (clk.due()) {
            if (R::traj(a)) ymc_store_row<D>(a.traj + (uint64_t)clk.row * a.K * D, k, x);
            if (R::lp(a)) a.lp_out[(uint64_t)clk.row * a.K + k] = lpf;
            clk.wrote();
        }
    }

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

template <int KIND, int D, unsigned REC, int EQ>
__global__ __launch_bounds__(kBlock) void mlda2_kernel(const ChainArgs<D> a)
{
    mlda2_body<KIND, D, REC, EQ>(a);
}

/*
 * Lotka-Volterra posterior for SMALL ensembles (BASELINE configs[3]: 16 384
 * chains per GPU = a quarter of a wave per SIMD with one lane per chain).  The
 * likelihood is n_data independent ODE solves (testSetup.py:118-141), so a chain
 * is spread over G = n_data lanes: lane r integrates design row r, the row
 * norms are gathered in row order (the canonical sequential sum).  The draws do
 * not depend on the state, so they are not repeated by the G lanes of a chain:
 * once per round of G transitions lane r makes the proposal increment and the
 * accept uniform of transition (round start + r) and leaves them in a
 * wave-private LDS slot; at every transition the group reads its slot (one
 * broadcast read).  What the lanes still repeat is the cheap state-dependent
 * rest (prior, exp, accept, moments).  64 / G chains per wave.  Same bits as
 * mrw_kernel.
 */
/* transitions [n_begin, n_end) of this launch for the cpw chains of chain wave `cwave` (state loaded from and
 * stored to the run's arrays, so a chain wave's launch can be cut into segments run by different waves) */
/* ADAPT: the lane's solve by the adaptive integrator (lv_rtol > 0) -- its own instantiation, the fixed-step kernel's
 * registers and loop are what configs[3] is measured on */
template <unsigned REC, bool ADAPT = false>
__device__ __forceinline__ void lv_rows_segment(const ChainArgs<2> &a, uint64_t cwave, uint32_t n_begin, uint32_t n_end,
                                                double (*slots)[3])
{
    constexpr int D = 2;
    using R = Rec<REC>;
    const DensityDev<2> &p = a.tgt;
    const int G = p.n_data;
    const int cpw = 64 / G;                                  /* chains per wave */
    const int lane = threadIdx.x & 63;
    const int slot = lane / G, row = lane - slot * G;
    const uint64_t k = cwave * cpw + slot;
    const bool live = slot < cpw && k < a.K;
    const uint64_t kk = live ? k : 0;                        /* idle lanes shadow chain 0 and never store */
    const uint32_t chain = (uint32_t)(a.chain0 + kk);
    const int base = slot * G;

    double x[D], wmean[D], wm2[D], sigma[D] = {a.sigma[0], a.sigma[1]};
    ymc_load_row<D>(a.x, kk, x);
    double lp = a.lp[kk];
    uint64_t accepted = a.acc_count[kk];
    uint32_t st = a.status[kk];
    const bool moments = R::moments(a);
    wmean[0] = wmean[1] = wm2[0] = wm2[1] = 0.0;
    if (moments) {
        ymc_load_row<D>(a.wf_mean, kk, wmean);
        ymc_load_row<D>(a.wf_m2, kk, wm2);
    }
    const double alpha = p.alpha, gamma = p.gamma;
    const double h = p.t_final / (double)p.rk4_steps, h2 = 0.5 * h, h6 = h / 6.0;
    const int myrow = (slot < cpw) ? row : 0;
    const double dx0 = p.design[2 * myrow], dy0 = p.design[2 * myrow + 1];
    const double obs0 = p.data[2 * myrow], obs1 = p.data[2 * myrow + 1];
    RecordClock clk(a.rec_stride, a.rec_since);
    clk.skip(n_begin);
    int in_round = 0;                                   /* transition index within the round of G (segments start on rounds) */

#pragma unroll 1
    for (uint32_t n = n_begin; n < n_end; ++n) {
        if (in_round == 0) { /* wave-uniform: lane r of every group makes the draws of transition n + r */
            const uint64_t tr = a.t0 + n + (uint64_t)row;
            double z0, z1;
            ymc_draw_normal_pair(a.seed, chain, tr, 0u, &z0, &z1);
            const double ur = ymc_draw_uniform(a.seed, chain, tr);
            __builtin_amdgcn_wave_barrier();
            slots[lane][0] = sigma[0] * z0;             /* covariance.py:51-52 */
            slots[lane][1] = sigma[1] * z1;
            slots[lane][2] = ur;
            __builtin_amdgcn_wave_barrier();
        }
        const int src = (base + in_round) & 63;
        double y[D];
        y[0] = x[0] + slots[src][0];                     /* mrw.py:32-38 -> gaussian.py:66 */
        y[1] = x[1] + slots[src][1];
        const double u = slots[src][2];
        in_round = (in_round + 1 == G) ? 0 : in_round + 1;
        const bool moved = !ymc_states_equal<D>(y, x, YMC_EQ_EXACT);
        /* this lane's ODE solve (Density<POSTERIOR_LV>::eval, one design row) */
        const double beta = ymc_exp(y[0]), delta = ymc_exp(y[1]);
        double sx = dx0, sy = dy0;
        /* five steps per loop iteration: at ~2.7 waves per SIMD the loop branch of a 30-instruction body is a
         * visible bubble (north_star's 50 steps divide evenly; any other count finishes in the remainder loop) */
        if constexpr (ADAPT) {
            ymc_lv_solve_adaptive(sx, sy, alpha, beta, gamma, delta, p.t_final, p.lv_rtol, p.lv_atol);
        } else {
            int s = 0;
#pragma unroll 1
            for (; s + 5 <= p.rk4_steps; s += 5) {
#pragma unroll
                for (int r = 0; r < 5; ++r) ymc_lv_rk4_step(&sx, &sy, alpha, beta, gamma, delta, h, h2, h6);
            }
#pragma unroll 1
            for (; s < p.rk4_steps; ++s) ymc_lv_rk4_step(&sx, &sy, alpha, beta, gamma, delta, h, h2, h6);
        }
        const int bad_l = (!ymc_finite(sx) || !ymc_finite(sy)) ? 1 : 0;
        const double res[2] = {sx - obs0, sy - obs1};
        const double term = p.noise_linv ? ymc_whitened_norm_sq_reg<2>(p.noise_linv, res, 2)
                                         : ymc_diag_norm_sq<2>(res, p.noise_prec, 2);
        /* gather the rows in order: total = ((t_0) + t_1) + ... (likelihood.py:80) */
        double total = 0.0;
        int bad = 0;
        for (int i = 0; i < G; ++i) {
            const uint64_t b = ymc_bits(term);
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)b, base + i, 64);
            const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(b >> 32), base + i, 64);
            total = total + ymc_from_bits(((uint64_t)hi << 32) | lo);
            bad |= __shfl(bad_l, base + i, 64);
        }
        double lpy = -0.5 * total + ymc_prior_log<D>(p, y);
        if (bad) {
            st |= YMC_CHAIN_NONFINITE;
            lpy = -__builtin_inf();
        }
        const bool acc = ymc_accept(lpy - lp, moved, ymc_finite(lpy), u, st);
        if (moments) ymc_welford<D>(x, (double)(a.wf_n0 + n + 1), ymc_recip_at(a.recip, n), wmean, wm2);
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = acc ? y[j] : x[j];
        lp = acc ? lpy : lp;
        accepted += acc ? 1u : 0u;
        const bool rec = clk.due();
        if (live && row == 0) {
            if (R::acc(a)) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
            if (rec && R::traj(a)) ymc_store_row<D>(a.traj + (uint64_t)clk.row * a.K * D, k, x);
            if (rec && R::lp(a)) a.lp_out[(uint64_t)clk.row * a.K + k] = lp;
        }
        if (rec) clk.wrote();
    }
    if (live && row == 0) {
        ymc_store_row<D>(a.x, k, x);
        a.lp[k] = lp;
        a.acc_count[k] = accepted;
        a.status[k] = st;
        if (moments) {
            ymc_store_row<D>(a.wf_mean, k, wmean);
            ymc_store_row<D>(a.wf_m2, k, wm2);
        }
    }
}

template <unsigned REC, bool ADAPT = false>
__global__ __launch_bounds__(kBlock) void mrw_lv_rows_kernel(const ChainArgs<2> a)
{
    ymc_stage_tables();
    __shared__ double draw_slots[kBlock / 64][64][3]; /* per wave and lane: sigma_0 z_0, sigma_1 z_1, u */
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    lv_rows_segment<REC, ADAPT>(a, wave, 0u, a.nsteps, draw_slots[threadIdx.x >> 6]);
}

/*
 * The same work cut to fit the chip.  BASELINE configs[3] gives a GPU 16 384 chains = 2 731 chain waves for 1 024
 * SIMDs: a third of the CUs carry two workgroups instead of three and idle for the last third of the launch, whatever
 * a wave's lane count (profiles/r03_c4_shape.txt).  Here the launch has a whole number of waves per SIMD (P = pieces)
 * and every wave takes an equal piece of the RIBBON of all chain waves' transitions laid end to end: the tail
 * [t0, T) of chain wave g and the head [0, h) of chain wave g + 1.  A wave runs its HEAD first -- it depends on nothing
 * -- and publishes (g + 1, h) in `handoff`; then it waits for its predecessor's publication of (g, t0), which that wave
 * made after ITS head, and runs the tail from the state the predecessor left in the run's arrays.  Pieces start on
 * draw rounds (multiples of n_data), so a chain's draws and arithmetic are those of mrw_lv_rows_kernel: same bits.
 * Waiting waves sleep; the SIMD's other waves use the slots, so a SIMD is busy for the sum of its pieces.  A wave that
 * waits beyond any plausible launch length flags its chains (YMC_CHAIN_HANDOFF) and leaves: the grid always drains.
 */
template <unsigned REC>
__global__ __launch_bounds__(kBlock) void mrw_lv_rows_balanced_kernel(const ChainArgs<2> a)
{
    ymc_stage_tables();
    __shared__ double draw_slots[kBlock / 64][64][3];
    double (*slots)[3] = draw_slots[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const uint64_t piece = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint64_t T = a.nsteps, L = a.lv_piece_len;
    const int cpw = 64 / a.tgt.n_data;
    const uint64_t G = (a.K + cpw - 1) / cpw, end = G * T;
    const uint64_t r0 = piece * L, r1 = (r0 + L < end) ? r0 + L : end;
    if (piece >= (uint64_t)a.lv_pieces || r0 >= r1) return;
    const uint64_t g0 = r0 / T;
    const uint32_t t0 = (uint32_t)(r0 - g0 * T);
    const uint32_t tail_end = (uint32_t)((r1 - r0) < (T - t0) ? t0 + (r1 - r0) : T);
    const uint32_t head_len = (uint32_t)((r1 - r0) - (tail_end - t0));

    /* phase 0: the head of chain wave g0 + 1 (no dependency); phase 1: the tail of g0, after its predecessor's head.
     * One call site of the segment (a loop, not two inlined copies: half the code, fewer live scalars) */
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
        const uint64_t g = phase == 0 ? g0 + 1 : g0;
        const uint32_t nb = phase == 0 ? 0u : t0, ne = phase == 0 ? head_len : tail_end;
        if (nb >= ne) continue;
        if (nb > 0) {
            int ok = 1;
            if (lane == 0) {
                uint32_t spins = 0;
                /* relaxed polls (an agent-scope load goes to L2; an ACQUIRE here would invalidate the CU's caches under
                 * its working waves at every poll), one acquire fence after the loop */
                while (__hip_atomic_load(a.handoff + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nb) {
                    __builtin_amdgcn_s_sleep(127);
                    if (++spins > a.lv_poll_limit) {      /* 2^22 polls = tens of seconds: something is wrong, do not hang the device */
                        ok = 0;
                        break;
                    }
                }
            }
            ok = __shfl(ok, 0, 64);
            __threadfence();                              /* nothing of the predecessor's state from a stale cache line */
            if (!ok) {
                const int nd = a.tgt.n_data, slot = lane / nd, row = lane - slot * nd;
                const uint64_t k = g * cpw + slot;
                if (slot < cpw && k < a.K && row == 0) a.status[k] |= YMC_CHAIN_HANDOFF;
                /* the tail [nb, ne) of chain wave g is not run (nobody waits for a tail): the host turns the count
                 * into a failed ymc_run_sync / fetch */
                if (lane == 0) __hip_atomic_fetch_add(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
        }
        lv_rows_segment<REC>(a, g, nb, ne, slots);
        if (ne < T) {
            __threadfence();                              /* the segment's state stores before the flag */
            if (lane == 0) __hip_atomic_store(a.handoff + g, ne, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

/* DensityInterface.evaluate_log for a batch of parameters */
template <int KIND, int D>
__global__ __launch_bounds__(kBlock) void logdensity_kernel(const DensityDev<D> p, uint64_t n,
                                                            const double *__restrict__ xin,
                                                            double *__restrict__ out,
                                                            uint32_t *__restrict__ status)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double x[D];
    ymc_load_row<D>(xin, i, x);
    uint32_t st = 0;
    out[i] = Density<KIND, D>::eval(p, x, st);
    if (status) status[i] = st;
}

/* ForwardModel.evaluate for a batch (linear map held per lane) */
template <int D>
__global__ __launch_bounds__(kBlock) void forward_linear_kernel(const DensityDev<D> p, uint64_t n,
                                                                const double *__restrict__ xin,
                                                                double *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double x[D], F[YMC_LANE_MMAX];
    ymc_load_row<D>(xin, i, x);
    Density<YMC_DENSITY_POSTERIOR_LINEAR, D>::forward(p, x, F);
#pragma unroll
    for (int m = 0; m < YMC_LANE_MMAX; ++m)
        if (m < p.data_dim) out[i * p.data_dim + m] = F[m];
}

/* ForwardModel.evaluate for a batch, extension kinds: out[i][row][output] (Density<KIND, D>::forward_row) */
template <int KIND, int D>
__global__ __launch_bounds__(kBlock) void forward_rows_kernel(const DensityDev<D> p, uint64_t n,
                                                              const double *__restrict__ xin,
                                                              double *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    constexpr int M = Density<KIND, D>::kOutputs;
    double th[D];
    ymc_load_row<D>(xin, i, th);
    for (int r = 0; r < p.n_data; ++r) {
        double F[M];
        Density<KIND, D>::forward_row(p, th, r, F);
#pragma unroll
        for (int j = 0; j < M; ++j) out[(i * p.n_data + r) * M + j] = F[j];
    }
}

template <int UNUSED = 0> /* a template only so that the parts without d = 2 do not emit it */
__global__ __launch_bounds__(kBlock) void forward_lv_kernel(const DensityDev<2> p, uint64_t n,
                                                            const double *__restrict__ xin,
                                                            double *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double beta = ymc_exp(xin[2 * i]), delta = ymc_exp(xin[2 * i + 1]);
    const double h = p.t_final / (double)p.rk4_steps, h2 = 0.5 * h, h6 = h / 6.0;
    for (int r = 0; r < p.n_data; ++r) {
        double x = p.design[2 * r], y = p.design[2 * r + 1];
        if (p.lv_rtol > 0.0) {
            ymc_lv_solve_adaptive(x, y, p.alpha, beta, p.gamma, delta, p.t_final, p.lv_rtol, p.lv_atol);
        } else {
            for (int s = 0; s < p.rk4_steps; ++s) {
                ymc_lv_rk4_step(&x, &y, p.alpha, beta, p.gamma, delta, h, h2, h6);
            }
        }
        out[(i * p.n_data + r) * 2] = x;
        out[(i * p.n_data + r) * 2 + 1] = y;
    }
}

#if YMC_KERNELS_PART == 0
/* the draws of chains chain0..chain0+K-1 in transitions t0..t0+nt-1 */
__global__ __launch_bounds__(kBlock) void rng_fill_kernel(uint64_t seed, uint64_t chain0, uint64_t K,
                                                          uint64_t t0, uint64_t nt, int d,
                                                          double *__restrict__ xi, double *__restrict__ u)
{
    ymc_stage_tables();
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= K * nt) return;
    const uint64_t t = i / K, k = i % K;
    const uint32_t chain = (uint32_t)(chain0 + k);
    if (u) u[i] = ymc_draw_uniform(seed, chain, t0 + t);
    if (xi) {
        for (int j = 0; 2 * j < d; ++j) {
            double z0, z1;
            ymc_draw_normal_pair(seed, chain, t0 + t, (uint32_t)j, &z0, &z1);
            xi[i * d + 2 * j] = z0;
            if (2 * j + 1 < d) xi[i * d + 2 * j + 1] = z1;
        }
    }
}

/* out[i] = RN(1 / (n0 + i + 1)): the Welford divisors' reciprocals of one launch */
__global__ __launch_bounds__(kBlock) void recip_kernel(uint64_t n0, uint64_t count, double *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < count) out[i] = 1.0 / (double)(n0 + i + 1);
}

__global__ __launch_bounds__(kBlock) void numerics_kernel(int fn, uint64_t n, const double *__restrict__ x,
                                                          double *__restrict__ y0, double *__restrict__ y1)
{
    ymc_stage_tables();
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    switch (fn) {
    case 0: y0[i] = ymc_log(x[i]); break;
    case 1: y0[i] = ymc_exp(x[i]); break;
    case 2: {
        double s, c;
        ymc_sincospi(x[i], &s, &c);
        y0[i] = s;
        y1[i] = c;
        break;
    }
    case 3: y0[i] = __builtin_sqrt(x[i]); break;
    case 6: y0[i] = ymc_sqrt_radius(x[i]); break;
    case 4: y0[i] = x[i] / y1[i]; break;
    default: break;
    }
}

/* One wave that sleeps next to the working kernels for `ticks` periods of the 100 MHz reference
 * counter and reports how many shader cycles went by: out[0] = s_memtime delta, out[1] =
 * s_memrealtime delta.  Always terminates: the reference counter advances whatever the chip does. */
__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t ticks, uint64_t *__restrict__ out)
{
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    uint64_t r1;
    do {
        __builtin_amdgcn_s_sleep(127);
        r1 = __builtin_amdgcn_s_memrealtime();
    } while (r1 - r0 < ticks);
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

/* per-chain summary record, 3d+5 eight-byte words:
 * state[d], logpost, accepted, wf_n, wf_mean[d], wf_m2[d], status */
__global__ __launch_bounds__(kBlock) void pack_summary_kernel(uint64_t K, int d, uint64_t wf_n,
                                                              const double *__restrict__ x,
                                                              const double *__restrict__ lp,
                                                              const uint64_t *__restrict__ accepted,
                                                              const double *__restrict__ wmean,
                                                              const double *__restrict__ wm2,
                                                              const uint32_t *__restrict__ status,
                                                              uint64_t *__restrict__ out)
{
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= K) return;
    uint64_t *o = out + k * (uint64_t)(3 * d + 5);
    for (int j = 0; j < d; ++j) o[j] = ymc_bits(x[k * d + j]);
    o[d] = ymc_bits(lp[k]);
    o[d + 1] = accepted[k];
    o[d + 2] = wf_n;
    for (int j = 0; j < d; ++j) o[d + 3 + j] = ymc_bits(wmean[k * d + j]);
    for (int j = 0; j < d; ++j) o[2 * d + 3 + j] = ymc_bits(wm2[k * d + j]);
    o[3 * d + 3] = status[k];
    o[3 * d + 4] = 0;
}


#endif /* part 0 */

/* pCN is defined in relation to a Bayesian model (pcn.py:77-83): every kind but the two explicit Gaussian targets */
template <int KIND>
constexpr bool kKindIsPosterior = KIND != YMC_DENSITY_GAUSS_WHITENED && KIND != YMC_DENSITY_GAUSS_ISO1D;

template <int KIND, int D, unsigned REC, int EQ>
void launch_one(int method, const ChainArgs<D> &a, hipStream_t s)
{
    if (method == YMC_METHOD_PCN) {
        if constexpr (REC == kRuntimeRec && kKindIsPosterior<KIND>)
            hipLaunchKernelGGL((mrw_kernel<KIND, D, REC, EQ, kPropPcn>), dim3(grid_for(a.K)), dim3(kBlock), 0, s, a);
    } else if (method == YMC_METHOD_MRW && a.chol) {
        if constexpr (REC == kRuntimeRec)
            hipLaunchKernelGGL((mrw_kernel<KIND, D, REC, EQ, kPropDense>), dim3(grid_for(a.K)), dim3(kBlock), 0, s, a);
    } else if (method == YMC_METHOD_MRW) {
        if constexpr (REC != kRuntimeRec) {
            if constexpr (D <= 2 && KIND != YMC_DENSITY_POSTERIOR_LV) { /* beyond d = 2 the consumer no longer fits four waves per SIMD without spilling */
                if (a.split) {
                    const dim3 grid((unsigned)((a.K + kSplitChains - 1) / kSplitChains));
                    if (a.split == 2)
                        hipLaunchKernelGGL((mrw_split_kernel<KIND, D, REC, EQ, 2>), grid, dim3(768), 0, s, a);
                    else
                        hipLaunchKernelGGL((mrw_split_kernel<KIND, D, REC, EQ, 3>), grid, dim3(1024), 0, s, a);
                    return;
                }
            }
            hipLaunchKernelGGL((mrw_kernel_two_waves<KIND, D, REC, EQ>), dim3(grid_for(a.K)), dim3(kBlock), 0, s, a);
        } else
            hipLaunchKernelGGL((mrw_kernel<KIND, D, REC, EQ>), dim3(grid_for(a.K)), dim3(kBlock), 0, s, a);
    } else if constexpr (REC == kRuntimeRec || (REC & kThin) == 0) { /* the thinned masks are the random walk's */
        if constexpr (REC != kRuntimeRec)
            if (ymc_launch_mlda2_static<D>(KIND, REC, a, s)) return;
        hipLaunchKernelGGL((mlda2_kernel<KIND, D, REC, EQ>), dim3(grid_for(a.K)), dim3(kBlock), 0, s, a);
    }
}

/* the specialised kernels cover what the builders produce by default (trajectory
 * + decisions, with or without FullDiagnostics' moments, exact state equality);
 * every other combination runs the run-time-flag form */
template <int KIND, int D>
hipError_t launch_chain(int method, const ChainArgs<D> &a, hipStream_t s)
{
    const unsigned rec = (a.traj ? YMC_RECORD_TRAJECTORY : 0u) | (a.acc ? YMC_RECORD_ACCEPT : 0u) |
                         (a.lp_out ? YMC_RECORD_LOGPOST : 0u) | (a.sub_acc ? YMC_RECORD_SUB_ACCEPT : 0u) |
                         (a.flags & YMC_RECORD_MOMENTS);
    const bool thin = a.rec_stride > 1, exact = a.eq_mode == YMC_EQ_EXACT;
    if (method == YMC_METHOD_PCN || a.chol)
        launch_one<KIND, D, kRuntimeRec, -1>(method, a, s);
    else if (exact && !thin && rec == kTA)
        launch_one<KIND, D, kTA, YMC_EQ_EXACT>(method, a, s);
    else if (exact && !thin && rec == (kTA | YMC_RECORD_MOMENTS))
        launch_one<KIND, D, kTA | YMC_RECORD_MOMENTS, YMC_EQ_EXACT>(method, a, s);
    else if (exact && thin && rec == kT && method == YMC_METHOD_MRW)
        launch_one<KIND, D, kThin | kT, YMC_EQ_EXACT>(method, a, s);
    else if (exact && thin && rec == (kT | YMC_RECORD_MOMENTS) && method == YMC_METHOD_MRW)
        launch_one<KIND, D, kThin | kT | YMC_RECORD_MOMENTS, YMC_EQ_EXACT>(method, a, s);
    else if (exact && thin && method == YMC_METHOD_MLDA2 && (rec == kT || rec == (kT | YMC_RECORD_MOMENTS)) &&
             ymc_launch_mlda2_static<D>(KIND, rec, a, s))
        ; /* two-level with a compile-time sub-chain length and a thinned record (ymc_mlda2.hip) */
    else
        launch_one<KIND, D, kRuntimeRec, -1>(method, a, s);
    return hipGetLastError();
}

}  // namespace


template <int D>
hipError_t ymc_launch_chain(int kind, int method, const ChainArgs<D> &a, hipStream_t s)
{
    switch (kind) {
    case YMC_DENSITY_GAUSS_WHITENED: return launch_chain<YMC_DENSITY_GAUSS_WHITENED, D>(method, a, s);
    case YMC_DENSITY_GAUSS_ISO1D:
        if constexpr (D == 1) return launch_chain<YMC_DENSITY_GAUSS_ISO1D, 1>(method, a, s);
        return hipErrorInvalidValue;
    case YMC_DENSITY_POSTERIOR_LINEAR: return launch_chain<YMC_DENSITY_POSTERIOR_LINEAR, D>(method, a, s);
    case YMC_DENSITY_POSTERIOR_LV:
        if constexpr (D == 2) {
            if (method == YMC_METHOD_MRW && a.eq_mode == YMC_EQ_EXACT && a.lv_rows && !a.tgt.likelihood_only && !a.chol) {
                /* small ensemble: n_data lanes per chain */
                const int cpw = 64 / a.tgt.n_data;
                const uint64_t waves = (a.K + cpw - 1) / cpw;
                if (a.tgt.lv_rtol > 0.0) /* solves of uneven length: no equal pieces to cut */
                    hipLaunchKernelGGL((mrw_lv_rows_kernel<kRuntimeRec, true>), dim3((unsigned)((waves * 64 + kBlock - 1) / kBlock)),
                                       dim3(kBlock), 0, s, a);
                else if (a.lv_pieces > 0) /* a whole number of waves per SIMD, equal pieces of the ribbon (see the kernel) */
                    hipLaunchKernelGGL((mrw_lv_rows_balanced_kernel<kRuntimeRec>),
                                       dim3((unsigned)(((uint64_t)a.lv_pieces * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
                else
                    hipLaunchKernelGGL((mrw_lv_rows_kernel<kRuntimeRec>), dim3((unsigned)((waves * 64 + kBlock - 1) / kBlock)),
                                       dim3(kBlock), 0, s, a);
                return hipGetLastError();
            }
            return launch_chain<YMC_DENSITY_POSTERIOR_LV, 2>(method, a, s);
        }
        return hipErrorInvalidValue;
    /* the extension kinds, generated from the registration table */
#define YMC_EXT_KIND(ENUM, NAME, DMIN, DMAX, OUT, COLS)                                        \
    case ENUM:                                                                                 \
        if constexpr (D >= DMIN && D <= DMAX) return launch_chain<ENUM, D>(method, a, s);      \
        return hipErrorInvalidValue;
#include "ymc_kinds.def"
#undef YMC_EXT_KIND
    default: return hipErrorInvalidValue;
    }
}

template <int D>
hipError_t ymc_launch_logdensity(int kind, const DensityDev<D> &p, uint64_t n, const double *x, double *out,
                                 uint32_t *status, hipStream_t s)
{
    dim3 g(grid_for(n)), b(kBlock);
    switch (kind) {
    case YMC_DENSITY_GAUSS_WHITENED:
        hipLaunchKernelGGL((logdensity_kernel<YMC_DENSITY_GAUSS_WHITENED, D>), g, b, 0, s, p, n, x, out, status);
        break;
    case YMC_DENSITY_GAUSS_ISO1D:
        if constexpr (D == 1)
            hipLaunchKernelGGL((logdensity_kernel<YMC_DENSITY_GAUSS_ISO1D, 1>), g, b, 0, s, p, n, x, out, status);
        else
            return hipErrorInvalidValue;
        break;
    case YMC_DENSITY_POSTERIOR_LINEAR:
        hipLaunchKernelGGL((logdensity_kernel<YMC_DENSITY_POSTERIOR_LINEAR, D>), g, b, 0, s, p, n, x, out, status);
        break;
    case YMC_DENSITY_POSTERIOR_LV:
        if constexpr (D == 2)
            hipLaunchKernelGGL((logdensity_kernel<YMC_DENSITY_POSTERIOR_LV, 2>), g, b, 0, s, p, n, x, out, status);
        else
            return hipErrorInvalidValue;
        break;
#define YMC_EXT_KIND(ENUM, NAME, DMIN, DMAX, OUT, COLS)                                                        \
    case ENUM:                                                                                                 \
        if constexpr (D >= DMIN && D <= DMAX)                                                                  \
            hipLaunchKernelGGL((logdensity_kernel<ENUM, D>), g, b, 0, s, p, n, x, out, status);                \
        else                                                                                                   \
            return hipErrorInvalidValue;                                                                       \
        break;
#include "ymc_kinds.def"
#undef YMC_EXT_KIND
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int D>
hipError_t ymc_launch_forward(int kind, const DensityDev<D> &p, uint64_t n, const double *x, double *out,
                              hipStream_t s)
{
    dim3 g(grid_for(n)), b(kBlock);
    if (kind == YMC_DENSITY_POSTERIOR_LINEAR) {
        hipLaunchKernelGGL((forward_linear_kernel<D>), g, b, 0, s, p, n, x, out);
    } else if (kind == YMC_DENSITY_POSTERIOR_LV) {
        if constexpr (D == 2)
            hipLaunchKernelGGL(forward_lv_kernel<>, g, b, 0, s, p, n, x, out);
        else
            return hipErrorInvalidValue;
    } else {
        switch (kind) {
#define YMC_EXT_KIND(ENUM, NAME, DMIN, DMAX, OUT, COLS)                                              \
    case ENUM:                                                                                       \
        if constexpr (D >= DMIN && D <= DMAX)                                                        \
            hipLaunchKernelGGL((forward_rows_kernel<ENUM, D>), g, b, 0, s, p, n, x, out);            \
        else                                                                                         \
            return hipErrorInvalidValue;                                                             \
        break;
#include "ymc_kinds.def"
#undef YMC_EXT_KIND
        default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

#define YMC_INSTANTIATE(D)                                                                             \
    template hipError_t ymc_launch_chain<D>(int, int, const ChainArgs<D> &, hipStream_t);              \
    template hipError_t ymc_launch_logdensity<D>(int, const DensityDev<D> &, uint64_t, const double *, \
                                                 double *, uint32_t *, hipStream_t);                   \
    template hipError_t ymc_launch_forward<D>(int, const DensityDev<D> &, uint64_t, const double *,    \
                                              double *, hipStream_t);
YMC_FOR_EACH_D(YMC_INSTANTIATE)

#if YMC_KERNELS_PART == 0
hipError_t ymc_launch_rng_fill(uint64_t seed, uint64_t chain0, uint64_t K, uint64_t t0, uint64_t nt, int d,
                               double *xi, double *u, hipStream_t s)
{
    hipLaunchKernelGGL(rng_fill_kernel, dim3(grid_for(K * nt)), dim3(kBlock), 0, s, seed, chain0, K, t0, nt, d,
                       xi, u);
    return hipGetLastError();
}

hipError_t ymc_launch_recip(uint64_t n0, uint64_t count, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(recip_kernel, dim3(grid_for(count)), dim3(kBlock), 0, s, n0, count, out);
    return hipGetLastError();
}

hipError_t ymc_launch_numerics(int fn, uint64_t n, const double *x, double *y0, double *y1, hipStream_t s)
{
    hipLaunchKernelGGL(numerics_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, fn, n, x, y0, y1);
    return hipGetLastError();
}

hipError_t ymc_launch_clock_probe(uint64_t ticks, uint64_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s, ticks, out);
    return hipGetLastError();
}

hipError_t ymc_launch_pack_summary(uint64_t K, int d, uint64_t wf_n, const double *x, const double *lp,
                                   const uint64_t *accepted, const double *wmean, const double *wm2,
                                   const uint32_t *status, uint64_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(pack_summary_kernel, dim3(grid_for(K)), dim3(kBlock), 0, s, K, d, wf_n, x, lp, accepted,
                       wmean, wm2, status, out);
    return hipGetLastError();
}
#endif /* part 0 */
