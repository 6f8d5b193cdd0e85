/*
 * ymc_tile_nested.hip -- delayed acceptance over an even number of surrogates (YMC_METHOD_MLDA_NESTED,
 * include/ymc.h; chain/method/mlda.py:12-73,100-116) for linear posteriors at large dimension
 * (MLDABuilder.bayesModel = a BayesianRegressionModelHierarchy of three or more models,
 * mlda.py:315-335, modelHierarchy.py:6-53), on the matrix cores.
 *
 * Same chain-to-lane mapping as linear_tile_kernel (16 chains per wave, component 4 s + g of chain c in
 * register s of lane (g, c)); what the lane kernel of ymc_nested.hip writes as a recursion over the E
 * effective levels is an explicit walk here, because the states of the levels do not fit the registers:
 *
 *   registers   z  the state being walked at level 0 (and, on the way up, the proposal of each level)
 *               y  level 0's proposal / the parked state of the deciding level
 *   x_tiled     the current states of levels 1 .. E-1, parked in the record's coalesced layout
 *   LDS         level 0's posterior P[0] whole (it is evaluated most often): A fragments + constants
 *   L2          P[l], Q[l] of the levels above: fragments and constants read from global memory
 *
 * A level-l decision (mlda.py:35-43,146-154) compares the state z that came up from below with the parked
 * x_l, evaluates P[l](z) and Q[l](z) (two GEMVs on the matrix cores) and keeps one of the two; the counters
 * of the levels are wave-uniform, so the walk is straight-line code for the whole wave.  Decisions take
 * consecutive transition indices in execution order, exactly as in mlda_nested_kernel and the oracle.
 *
 * STAGED (as linear_tile_mlda2_kernel): the waves of a workgroup walk in lockstep, so a matrix of an upper
 * level comes from L2 once per workgroup and evaluation, through a double buffer in the LDS left beside level
 * 0's block; waves past the last tile stay in the walk for the barriers and touch no memory.
 */
#define YMC_COEFF_IN_SGPR 1
#include <cstdlib>

#include "ymc_kernels.h"
#include "ymc_tile.h"
#include "ymc_tile_dev.h"

namespace {

/* four waves per workgroup (one per SIMD; the LDS admits one workgroup per CU anyway): at KS = 32 the walk
 * needs z, y, two accumulator chains and the fragments in flight from L2 -- more than the 256 registers a
 * wave gets at two waves per SIMD */
constexpr int kNestWaves = 4;
constexpr int kNestThreads = 64 * kNestWaves;

/* one value per level, addressed by a run-time (wave-uniform) level: named members and selects, so that the
 * values stay in registers (an indexed array went to scratch) */
template <typename T>
struct PerLevel {
    T v0, v1, v2, v3, v4, v5, v6, v7;
    static_assert(YMC_MAX_LEVELS == 8, "PerLevel holds eight levels");
    /* every member is read BEFORE the selects: a conditional over the members themselves becomes a load through a
     * selected address, which at eight members leaves the whole struct in private memory */
    __device__ __forceinline__ T get(int l) const
    {
        const T a0 = v0, a1 = v1, a2 = v2, a3 = v3, a4 = v4, a5 = v5, a6 = v6, a7 = v7;
        const T lo = l == 0 ? a0 : l == 1 ? a1 : l == 2 ? a2 : a3;
        const T hi = l == 4 ? a4 : l == 5 ? a5 : l == 6 ? a6 : a7;
        return l < 4 ? lo : hi;
    }
    __device__ __forceinline__ void set(int l, T x)
    {
        const T a0 = v0, a1 = v1, a2 = v2, a3 = v3, a4 = v4, a5 = v5, a6 = v6, a7 = v7;
        v0 = l == 0 ? x : a0;
        v1 = l == 1 ? x : a1;
        v2 = l == 2 ? x : a2;
        v3 = l == 3 ? x : a3;
        v4 = l == 4 ? x : a4;
        v5 = l == 5 ? x : a5;
        v6 = l == 6 ? x : a6;
        v7 = l == 7 ? x : a7;
    }
};

/* BSTREAM: level 0's matrix streams through the shared double buffer too (its constants stay in LDS): where it does not
 * fit the LDS, and at KS = 64 (d <= 256) */
template <int KS, bool STAGED, bool BSTREAM = false>
__global__ __launch_bounds__(kNestThreads) void linear_tile_nested_kernel(const TileArgs a)
{
    static_assert(!BSTREAM || STAGED, "a streamed base matrix shares the upper levels' double buffer");
    ymc_stage_tables();
    extern __shared__ double lds[];
    const int MT0 = a.MT, m_pad0 = 16 * MT0, d_pad = 4 * KS;
    const size_t frag0 = (size_t)MT0 * KS * 64;
    const double *A0 = BSTREAM ? a.blob : lds;
    double *bias0 = lds + (BSTREAM ? 0 : frag0);
    double *nprec0 = bias0 + m_pad0;
    double *dbar0 = nprec0 + m_pad0;
    double *pmean0 = dbar0 + m_pad0;
    double *pprec0 = pmean0 + d_pad;
    double *sigma = pprec0 + d_pad;
    const size_t consts0 = (size_t)m_pad0 * 3 + 3 * (size_t)d_pad, blob0 = frag0 + consts0;
    double *stage = bias0 + consts0;                             /* STAGED: the stream's double buffer (ymc_tile_dev.h) */

    const int tid = threadIdx.x;
    if constexpr (BSTREAM) {
        for (size_t i = tid; i < consts0; i += kNestThreads) bias0[i] = a.blob[frag0 + i];
    } else {
        ymc_stage_block<kNestThreads>(lds, a.blob, blob0 / 2, tid);
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const uint64_t nct = (a.K + 15) / 16;
    const uint64_t tile = (uint64_t)blockIdx.x * kNestWaves + wave;
    const bool wave_live = tile < nct;
    if (!STAGED && !wave_live) return;
    const uint64_t k = tile * 16 + c;
    const bool live = wave_live && k < a.K;
    const uint32_t chain = (uint32_t)(a.chain0 + k);
    const int d = a.d, E = a.E;
    /* a local copy: indexing the argument block with a run-time level would put it in scratch */
    const PerLevel<int> steps = {a.steps[0], a.steps[1], a.steps[2], a.steps[3], a.steps[4], a.steps[5], a.steps[6], a.steps[7]};
    const uint32_t pair_of_g = (g == 0) ? 0u : (g == 1) ? 2u : (g == 2) ? 1u : 3u;
    /* parked state of level l >= 1: this lane's column */
    auto park = [&](int l) { return a.x_tiled + (((uint64_t)(l - 1) * nct + (wave_live ? tile : 0)) * KS) * 64 + lane; };

    double z[KS], y[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int j = 4 * s + g;
        z[s] = (live && j < d) ? a.x[k * d + j] : 0.0;
    }
    if (wave_live) {
        double *top = park(E - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) top[s * 64] = z[s];
    }
    PerLevel<double> lpP = {}, lpQ = {};
    lpP.set(E - 1, live ? a.lp[k] : 0.0);
    lpQ.set(E - 1, live ? a.lpc[k] : 0.0);
    PerLevel<uint64_t> lacc = {};
    for (int e = 0; e < E - 1; ++e)
        if (live) lacc.set(e, a.level_acc[(uint64_t)e * a.K + k]);
    uint64_t accepted = live ? a.acc_count[k] : 0;
    uint32_t st = live ? a.status[k] : 0;
    UniformCache uc;
    uc.valid = false;
    uc.pair = 0;
    RecordClock clk(a.rec_stride, a.rec_since);

    /* log-posterior of v: forward map on the matrix cores, Gaussian-noise likelihood through the row-mean
     * identity, Gaussian prior (as linear_tile_kernel); fragments and constants from LDS or from global memory */
    auto logpost = [&](const double *A, int MT, const double *bias, const double *nprec, const double *dbar,
                       const double *pmean, const double *pprec, double lik_const, const double (&v)[KS]) {
        const double logl = -0.5 * (combine4(ymc_tile_residual_sum<KS>(A, MT, bias, nprec, dbar, lane, g, v)) + lik_const);
        return logl + ymc_tile_logprior<KS>(pmean, pprec, g, v);
    };
    /* P[l] (which = 0) or Q[l] (which = 1) of a level above the base: everything from global memory */
    /* STAGED: the same value over the workgroup's shared stream of the matrix (ymc_stream_tile_pairs,
     * ymc_tile_dev.h); every wave of the workgroup evaluates the same level at the same point of the walk */
    auto logpost_staged = [&](const double *A, int MT, const double *bias, const double *nprec, const double *dbar,
                              const double *pmean, const double *pprec, double lik_const, const double (&v)[KS]) {
        double Ssum = 0.0;
        ymc_stream_tile_pairs<KS, 4, kNestThreads>(A, MT / 2, stage, tid, lane, v, [&](int pair, const d4 &acc0, const d4 &acc1) {
            ymc_tile_add_residuals(2 * pair, acc0, acc1, bias, nprec, dbar, g, Ssum);
        });
        const double logl = -0.5 * (combine4(Ssum) + lik_const);
        return logl + ymc_tile_logprior<KS>(pmean, pprec, g, v);
    };
    auto logpost0 = [&](const double (&v)[KS]) {
        if constexpr (BSTREAM) return logpost_staged(A0, MT0, bias0, nprec0, dbar0, pmean0, pprec0, a.lik_const, v);
        else return logpost(A0, MT0, bias0, nprec0, dbar0, pmean0, pprec0, a.lik_const, v);
    };
    auto logpost_of = [&](int l, int which, const double (&v)[KS]) {
        const TileLevel L = a.levels[2 * l + which];
        const double *consts = L.blob + (size_t)L.MT * KS * 64;
        const int m_pad = 16 * L.MT;
        if constexpr (STAGED)
            return logpost_staged(L.blob, L.MT, consts, consts + m_pad, consts + 2 * m_pad, consts + 3 * m_pad,
                                  consts + 3 * m_pad + d_pad, L.lik_const, v);
        else
            return logpost(L.blob, L.MT, consts, consts + m_pad, consts + 2 * m_pad, consts + 3 * m_pad,
                           consts + 3 * m_pad + d_pad, L.lik_const, v);
    };
    auto any_differs = [&](const double (&p)[KS], const double (&q)[KS]) { return ymc_tile_any_differs<KS>(p, q); };
    /* _accept_reject of level l (mlda.py:35-43; the top level's: mlda.py:146-154): the proposal is z, the
     * level's state is parked; afterwards z and the parking place hold the level's new state */
    auto decide = [&](int l, uint64_t t) {
        double *xl = park(l);
#pragma unroll
        for (int s = 0; s < KS; ++s) y[s] = wave_live ? xl[s * 64] : 0.0;
        const bool moved = any_differs(z, y);
        double lpPy = 0.0, lpQy = 0.0;
#pragma unroll 1
        for (int which = 0; which < 2; ++which) { /* one copy of the evaluation's code for both densities */
            const double val = logpost_of(l, which, z);
            lpPy = which == 0 ? val : lpPy;
            lpQy = which == 1 ? val : lpQy;
        }
        const double lpPx = lpP.get(l), lpQx = lpQ.get(l);
        const double log_ratio = ((lpPy + lpQx) - lpQy) - lpPx;
        const double u = ymc_uniform_cached(uc, a.seed, chain, t);
        const bool acc = ymc_accept(log_ratio, moved, ymc_finite(lpPy) && ymc_finite(lpQy), u, st);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            z[s] = acc ? z[s] : y[s];
            if (wave_live) xl[s * 64] = z[s];
        }
        lpP.set(l, acc ? lpPy : lpPx);
        lpQ.set(l, acc ? lpQy : lpQx);
        return acc;
    };

#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        uint64_t t = (a.t0 + n) * a.cost_top;
        PerLevel<int> cnt = {};
        int from = E - 2; /* the levels below the top are entered afresh: MLDAProposal.generate_proposal, mlda.py:100-116 */
        bool acc_top = false;
#pragma unroll 1
        for (;;) {
            /* MetropolisHastings.run of levels from .. 1 starts at z (metropolisHastings.py:103-110) */
#pragma unroll 1
            for (int l = from; l >= 1; --l) {
#pragma unroll 1
                for (int which = 0; which < 2; ++which) {
                    const double val = logpost_of(l, which, z);
                    if (which == 0) lpP.set(l, val);
                    else lpQ.set(l, val);
                }
                double *xl = park(l);
                if (wave_live) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) xl[s * 64] = z[s];
                }
                cnt.set(l, 0);
            }
            /* level 0: a random walk of steps[0] transitions on P[0] (mrw.py:41-57) */
            double lp0 = logpost0(z);
            uint64_t acc0 = 0;
#pragma unroll 1
            for (int sub = 0; sub < steps.v0; ++sub) {
#pragma unroll
                for (int q = 0; q < KS / 2; ++q) {
                    double xi0, xi1;
                    ymc_draw_normal_pair(a.seed, chain, t, 4u * (uint32_t)q + pair_of_g, &xi0, &xi1);
                    swap_odd_even_rows(xi0, xi1);
                    y[2 * q] = z[2 * q] + sigma[8 * q + g] * xi0;
                    y[2 * q + 1] = z[2 * q + 1] + sigma[8 * q + 4 + g] * xi1;
                }
                const bool moved = any_differs(y, z);
                const double lpy = logpost0(y);
                const double u = ymc_uniform_cached(uc, a.seed, chain, t);
                const bool acc = ymc_accept(lpy - lp0, moved, ymc_finite(lpy), u, st);
#pragma unroll
                for (int s = 0; s < KS; ++s) z[s] = acc ? y[s] : z[s];
                lp0 = acc ? lpy : lp0;
                acc0 += acc ? 1u : 0u;
                ++t;
            }
            lacc.v0 += acc0;
            /* upwards: every level whose run below has ended decides; a level with steps left goes down again */
            int l = 1;
            bool top = false;
#pragma unroll 1
            for (;; ++l) { /* one call site of decide(): the recorded level's decision ends the walk */
                acc_top = decide(l, t);
                ++t;
                if (l == E - 1) {
                    top = true;
                    break;
                }
                lacc.set(l, lacc.get(l) + (acc_top ? 1u : 0u));
                const int done = cnt.get(l) + 1;
                cnt.set(l, done);
                if (done < steps.get(l)) break;
            }
            if (top) break;
            from = l - 1;
        }
        const bool acc = acc_top;
        accepted += acc ? 1u : 0u;

        const bool rec = clk.due();
        if (rec && a.traj && wave_live) {
            double *row = a.traj + (((uint64_t)clk.row * nct + tile) * KS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) row[s * 64] = z[s];
        }
        if (live && g == 0) {
            if (a.acc) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
            if (rec && a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + k] = lpP.get(E - 1);
        }
        if (rec) clk.wrote();
    }

    if (live) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int j = 4 * s + g;
            if (j < d) a.x[k * d + j] = z[s];
        }
        if (g == 0) {
            a.lp[k] = lpP.get(E - 1);
            a.lpc[k] = lpQ.get(E - 1);
            a.acc_count[k] = accepted;
            a.status[k] = st;
            for (int e = 0; e < E - 1; ++e) a.level_acc[(uint64_t)e * a.K + k] = lacc.get(e);
        }
    }
}

template <int KS, bool STAGED, bool BSTREAM = false>
hipError_t launch_staged_or_not(const TileArgs &a, size_t lds, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_tile_nested_kernel<KS, STAGED, BSTREAM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t nct = (a.K + 15) / 16;
    const unsigned grid = (unsigned)((nct + kNestWaves - 1) / kNestWaves);
    hipLaunchKernelGGL((linear_tile_nested_kernel<KS, STAGED, BSTREAM>), dim3(grid), dim3(kNestThreads), lds, s, a);
    return hipGetLastError();
}

template <int KS>
hipError_t launch_as(const TileArgs &a, hipStream_t s)
{
    /* the upper levels' matrices through a shared double buffer when the LDS has room beside level 0's block
     * (identical results; YMC_TILE_STAGE=0|1 overrides, the tests run both) */
    const size_t plain = ymc_tile_lds_bytes(KS, a.MT);
    const size_t unit2 = 2 * (size_t)ymc_stream_unit<KS, 4>() * sizeof(double), staged = plain + unit2;
    /* level 0's matrix streamed where it does not fit (YMC_TILE_NESTED=stream forces it: the tests compare) */
    const char *mode = std::getenv("YMC_TILE_NESTED");
    if (KS >= 64 || plain > kTileLdsLimit || (mode && mode[0] == 's')) {
        const size_t lds = ymc_tile_nested_stream_lds_bytes(KS, a.MT);
        if (lds > kTileLdsLimit) return hipErrorInvalidValue;
        return launch_staged_or_not<KS, true, true>(a, lds, s);
    }
    if constexpr (KS < 64) {
        bool stage = staged <= kTileLdsLimit;
        if (const char *env = std::getenv("YMC_TILE_STAGE")) stage = stage && env[0] != '0';
        return stage ? launch_staged_or_not<KS, true>(a, staged, s) : launch_staged_or_not<KS, false>(a, plain, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

size_t ymc_tile_nested_stream_lds_bytes(int KS, int MT)
{
    return ymc_tile_stream_lds_bytes(KS, MT); /* level 0's constants + the double buffer */
}

hipError_t ymc_launch_linear_tile_nested(int KS, const TileArgs &a, hipStream_t s)
{
    if (a.E < 2 || a.E > YMC_MAX_LEVELS || !a.levels || !a.x_tiled || !a.level_acc || !a.lpc) return hipErrorInvalidValue;
    switch (KS) {
    case 4: return launch_as<4>(a, s);
    case 8: return launch_as<8>(a, s);
    case 16: return launch_as<16>(a, s);
    case 32: return launch_as<32>(a, s);
    case 64: return launch_as<64>(a, s);
    default: return hipErrorInvalidValue;
    }
}
