/*
 * ymc_nested.hip -- delayed acceptance over an even number of surrogates the
 * way the reference executes it (chain/method/mlda.py:12-73,100-116; see
 * YMC_METHOD_MLDA_NESTED in include/ymc.h for the derivation): a nest of E
 * effective levels.  One lane owns one chain and walks the whole nest; the
 * nesting depth is a template parameter, each level a function instance.
 *
 *   level 0   MetropolisHastings.run of the base MetropolisedRandomWalk
 *             (metropolisHastings.py:103-120, mrw.py:51-57) on P[0]
 *   level e   MetropolisHastings.run of a SurrogateTransition (mlda.py:23-43):
 *             proposal = final state of a level-(e-1) run started at the
 *             current state; ratio exp(P(y) + Q(x) - Q(y) - P(x)) (:35-43)
 *   top       MLDA (mlda.py:119-154), the recorded chain
 *
 * Every decision takes the next transition index t in execution order (draw
 * addressing, ymc_numerics.h).  Densities are pure functions, so P(x), Q(x)
 * of a level's current state are carried instead of re-evaluated.
 *
 * Nests of more than kInlineLevels effective levels (built with
 * -DYMC_NESTED_DEEP into their own object): levels 0 .. kInlineLevels-2 stay
 * the inlined instances above -- that is where the transitions are -- and the
 * levels from kInlineLevels-1 up to E-2 are walked by a loop over a run-time
 * level whose per-level state (current point, P, Q, step count) lives in
 * private memory.
 */
#define YMC_COEFF_HOISTED 1
#include "ymc_nested.h"
#include "ymc_kinds.h"

namespace {

constexpr int kBlock = 256;
constexpr int kInlineLevels = 4;                            /* E of the deepest fully inlined nest */
constexpr int kLoopLevels = YMC_MAX_LEVELS - kInlineLevels; /* levels kInlineLevels-1 .. E-2 of a deeper one */

template <int KIND, int D>
struct Walker {
    const NestedArgs<D> &a;
    uint32_t chain;
    uint64_t t;
    uint32_t st;
    UniformCache uc;
    uint64_t lacc[kInlineLevels]; /* levels 0 .. kInlineLevels-1 (compile-time indices: registers) */
    double sigma[D];

    __device__ __forceinline__ explicit Walker(const NestedArgs<D> &args) : a(args) {}

    /* one run() of level e from x: x becomes the final state */
    template <int e>
    __device__ __forceinline__ void run(double (&x)[D])
    {
        const int S = a.steps[e];
        if constexpr (e == 0) {
            double lp = Density<KIND, D>::eval(a.P[0], x, st);
#pragma unroll 1
            for (int s = 0; s < S; ++s) {
                double y[D];
                ymc_propose<D>(x, sigma, a.seed, chain, t, y);                       /* mrw.py:27-38 */
                const double u = ymc_uniform_cached(uc, a.seed, chain, t);
                const bool moved = !ymc_states_equal<D>(y, x, a.eq_mode);            /* metropolisHastings.py:60-61 */
                const double lpy = Density<KIND, D>::eval(a.P[0], y, st);
                const bool acc = ymc_accept(lpy - lp, moved, ymc_finite(lpy), u, st); /* mrw.py:51-57 */
#pragma unroll
                for (int j = 0; j < D; ++j) x[j] = acc ? y[j] : x[j];
                lp = acc ? lpy : lp;
                lacc[0] += acc ? 1u : 0u;
                ++t;
            }
        } else {
            double lpP = Density<KIND, D>::eval(a.P[e], x, st);
            double lpQ = Density<KIND, D>::eval(a.Q[e], x, st);
#pragma unroll 1
            for (int s = 0; s < S; ++s) {
                double y[D];
#pragma unroll
                for (int j = 0; j < D; ++j) y[j] = x[j];
                run<e - 1>(y);                                                        /* mlda.py:23-33 */
                const bool acc = decide(a.P[e], a.Q[e], x, y, lpP, lpQ);
                lacc[e] += acc ? 1u : 0u;
            }
        }
    }

    /* _accept_reject of a SurrogateTransition / of MLDA (mlda.py:35-43,146-154) for the proposal y;
     * on acceptance x, lpP, lpQ move to y */
    __device__ __forceinline__ bool decide(const DensityDev<D> &P, const DensityDev<D> &Q, double (&x)[D],
                                           const double (&y)[D], double &lpP, double &lpQ)
    {
        const double u = ymc_uniform_cached(uc, a.seed, chain, t);
        ++t;
        const bool moved = !ymc_states_equal<D>(y, x, a.eq_mode);
        const double lpPy = Density<KIND, D>::eval(P, y, st);
        const double lpQy = Density<KIND, D>::eval(Q, y, st);
        const double log_ratio = ((lpPy + lpQ) - lpQy) - lpP;
        const bool acc = ymc_accept(log_ratio, moved, ymc_finite(lpPy) && ymc_finite(lpQy), u, st);
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = acc ? y[j] : x[j];
        lpP = acc ? lpPy : lpP;
        lpQ = acc ? lpQy : lpQ;
        return acc;
    }

    /* one run() of level E-2 >= kInlineLevels-1 from x: the recursion of run<e> above level B = kInlineLevels-1
     * unrolled into a loop.  Level l's frame (slot l - B): its current point, P and Q there, steps taken; a level
     * whose run is complete hands its point to the level above as that one's proposal.  `hi` = accepted
     * transitions of levels B+1 .. E-2 (level B's are lacc[B]) */
    __device__ __noinline__ void run_deep(int E, double (&x)[D], uint64_t (&hi)[kLoopLevels])
    {
        constexpr int B = kInlineLevels - 1;
        double X[kLoopLevels][D], lpP[kLoopLevels], lpQ[kLoopLevels];
        int done[kLoopLevels];
        auto enter = [&](int l) { /* the head of run<l>: P and Q of the starting point */
            double v[D];
#pragma unroll
            for (int j = 0; j < D; ++j) v[j] = X[l - B][j];
            lpP[l - B] = Density<KIND, D>::eval(a.P[l], v, st);
            lpQ[l - B] = Density<KIND, D>::eval(a.Q[l], v, st);
            done[l - B] = 0;
        };
        int l = E - 2;
#pragma unroll
        for (int j = 0; j < D; ++j) X[l - B][j] = x[j];
        enter(l);
        for (;;) {
            if (done[l - B] == a.steps[l]) {
                if (l == E - 2) break;
                /* level l's run is level l+1's proposal (mlda.py:23-43) */
                double cur[D], prop[D], p = lpP[l + 1 - B], q = lpQ[l + 1 - B];
#pragma unroll
                for (int j = 0; j < D; ++j) cur[j] = X[l + 1 - B][j], prop[j] = X[l - B][j];
                const bool acc = decide(a.P[l + 1], a.Q[l + 1], cur, prop, p, q);
#pragma unroll
                for (int j = 0; j < D; ++j) X[l + 1 - B][j] = cur[j];
                lpP[l + 1 - B] = p, lpQ[l + 1 - B] = q;
                hi[l - B] += acc ? 1u : 0u;
                ++l;
                ++done[l - B];
            } else if (l == B) {
                double cur[D], prop[D], p = lpP[0], q = lpQ[0];
#pragma unroll
                for (int j = 0; j < D; ++j) cur[j] = X[0][j], prop[j] = X[0][j];
                run<B - 1>(prop);
                const bool acc = decide(a.P[B], a.Q[B], cur, prop, p, q);
#pragma unroll
                for (int j = 0; j < D; ++j) X[0][j] = cur[j];
                lpP[0] = p, lpQ[0] = q;
                lacc[B] += acc ? 1u : 0u;
                ++done[0];
            } else {
#pragma unroll
                for (int j = 0; j < D; ++j) X[l - 1 - B][j] = X[l - B][j];
                --l;
                enter(l);
            }
        }
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = X[E - 2 - B][j];
    }
};

template <int KIND, int D, int E>
__global__ __launch_bounds__(kBlock) void mlda_nested_kernel(const NestedArgs<D> a)
{
    ymc_stage_tables();
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= a.K) return;

    Walker<KIND, D> w(a);
    w.chain = (uint32_t)(a.chain0 + k);
    w.st = a.status[k];
    w.uc.valid = false;
    w.uc.pair = 0;
    /* E = 0: a nest deeper than kInlineLevels, its depth in a.E */
    constexpr bool DEEP = E == 0;
    const int depth = DEEP ? a.E : E;
#pragma unroll
    for (int e = 0; e < kInlineLevels; ++e) w.lacc[e] = e < depth - 1 ? a.level_acc[(uint64_t)e * a.K + k] : 0;
    uint64_t hi[kLoopLevels];
    if constexpr (DEEP)
        for (int e = 0; e < kLoopLevels; ++e)
            hi[e] = kInlineLevels + e < depth - 1 ? a.level_acc[(uint64_t)(kInlineLevels + e) * a.K + k] : 0;
#pragma unroll
    for (int j = 0; j < D; ++j) w.sigma[j] = a.sigma[j];

    double x[D], wmean[D], wm2[D];
    ymc_load_row<D>(a.x, k, x);
    double lpP = a.lp[k], lpQ = a.lpq[k];
    uint64_t accepted = a.acc_count[k];
    const bool moments = (a.flags & YMC_RECORD_MOMENTS) != 0;
    if (moments) {
        ymc_load_row<D>(a.wf_mean, k, wmean);
        ymc_load_row<D>(a.wf_m2, k, wm2);
    }

    RecordClock clk(a.rec_stride, a.rec_since);
#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        w.t = (a.t0 + n) * a.cost_top;
        double y[D];
#pragma unroll
        for (int j = 0; j < D; ++j) y[j] = x[j];
        if constexpr (DEEP) w.run_deep(depth, y, hi);
        else w.template run<E - 2>(y);                                                /* mlda.py:100-116 */
        if (moments) ymc_welford<D>(x, (double)(a.wf_n0 + n + 1), ymc_recip_at(a.recip, n), wmean, wm2);
        const bool acc = w.decide(a.P[depth - 1], a.Q[depth - 1], x, y, lpP, lpQ);    /* mlda.py:146-154 */
        accepted += acc ? 1u : 0u;
        if (a.acc) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
        if (clk.due()) {
            if (a.traj) ymc_store_row<D>(a.traj + (uint64_t)clk.row * a.K * D, k, x);
            if (a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + k] = lpP;
            clk.wrote();
        }
    }

    ymc_store_row<D>(a.x, k, x);
    a.lp[k] = lpP;
    a.lpq[k] = lpQ;
    a.acc_count[k] = accepted;
    a.status[k] = w.st;
#pragma unroll
    for (int e = 0; e < kInlineLevels; ++e)
        if (e < depth - 1) a.level_acc[(uint64_t)e * a.K + k] = w.lacc[e];
    if constexpr (DEEP)
        for (int e = 0; e < kLoopLevels; ++e)
            if (kInlineLevels + e < depth - 1) a.level_acc[(uint64_t)(kInlineLevels + e) * a.K + k] = hi[e];
    if (moments) {
        ymc_store_row<D>(a.wf_mean, k, wmean);
        ymc_store_row<D>(a.wf_m2, k, wm2);
    }
}

template <int KIND, int D>
hipError_t launch_kind(const NestedArgs<D> &a, hipStream_t s)
{
    const dim3 grid((unsigned)((a.K + kBlock - 1) / kBlock)), block(kBlock);
#ifdef YMC_NESTED_DEEP
    if (a.E <= kInlineLevels || a.E > YMC_MAX_LEVELS) return hipErrorInvalidValue;
    hipLaunchKernelGGL((mlda_nested_kernel<KIND, D, 0>), grid, block, 0, s, a);
#else
    static_assert(kInlineLevels == 4, "one case per inlined depth");
    switch (a.E) {
    case 2: hipLaunchKernelGGL((mlda_nested_kernel<KIND, D, 2>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((mlda_nested_kernel<KIND, D, 3>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((mlda_nested_kernel<KIND, D, 4>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
#endif
    return hipGetLastError();
}

}  // namespace

#ifdef YMC_NESTED_DEEP
#define YMC_NESTED_ENTRY ymc_launch_nested_deep
#else
#define YMC_NESTED_ENTRY ymc_launch_nested
#endif

template <int D>
hipError_t YMC_NESTED_ENTRY(int kind, const NestedArgs<D> &a, hipStream_t s)
{
    switch (kind) {
    case YMC_DENSITY_GAUSS_WHITENED: return launch_kind<YMC_DENSITY_GAUSS_WHITENED, D>(a, s);
    case YMC_DENSITY_GAUSS_ISO1D:
        if constexpr (D == 1) return launch_kind<YMC_DENSITY_GAUSS_ISO1D, 1>(a, s);
        return hipErrorInvalidValue;
    case YMC_DENSITY_POSTERIOR_LINEAR: return launch_kind<YMC_DENSITY_POSTERIOR_LINEAR, D>(a, s);
    case YMC_DENSITY_POSTERIOR_LV:
        if constexpr (D == 2) return launch_kind<YMC_DENSITY_POSTERIOR_LV, 2>(a, s);
        return hipErrorInvalidValue;
#define YMC_EXT_KIND(ENUM, NAME, DMIN, DMAX, OUT, COLS)                                \
    case ENUM:                                                                         \
        if constexpr (D >= DMIN && D <= DMAX) return launch_kind<ENUM, D>(a, s);       \
        return hipErrorInvalidValue;
#include "ymc_kinds.def"
#undef YMC_EXT_KIND
    default: return hipErrorInvalidValue;
    }
}

#define YMC_INSTANTIATE_NESTED(D) template hipError_t YMC_NESTED_ENTRY<D>(int, const NestedArgs<D> &, hipStream_t);
YMC_INSTANTIATE_NESTED(1)
YMC_INSTANTIATE_NESTED(2)
YMC_INSTANTIATE_NESTED(3)
YMC_INSTANTIATE_NESTED(4)
YMC_INSTANTIATE_NESTED(5)
YMC_INSTANTIATE_NESTED(6)
YMC_INSTANTIATE_NESTED(7)
YMC_INSTANTIATE_NESTED(8)
