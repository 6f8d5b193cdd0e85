/*
 * ymc_wide_fused.hip -- the random walk (diagonal proposal) and pCN over linear models beyond 256 parameters as ONE
 * persistent launch per advance call (mrw.py:27-57, pcn.py:9-57, likelihood.py:74-84,
 * exampleSetup.py:42-52).  A translation unit of its own: the draws' polynomial coefficients are plain literals here
 * (ymc_numerics.h, ymc_fma_c) -- as SGPR pairs, the mode of ymc_tile.hip, they are spilled into VGPR lanes by this
 * kernel's scalar pressure and cost two v_readlane per use.
 */
#include <cstdlib>

#include "ymc_kernels.h"
#include "ymc_tile.h"
#include "ymc_tile_dev.h"

namespace {

/*
 * The wide path as a GEMM, fused over the whole launch.  linear_wide_eval_kernel above is 65 536
 * independent GEMVs -- every wave pulls every fragment of the matrix for its own 16 or 32 points, and a transition is four
 * launches that carry the proposal and the state through HBM (7 x K d 8 bytes per transition against the record's one).
 *
 * Here a workgroup of W waves owns NT tiles of 16 chains from the first transition of the launch to the last:
 *   state x      in registers, cut along k: wave w holds the k-steps of its q-range (a q = two k-steps = the eight
 *                components a lane's normal pair lands on, as in linear_tile_kernel)
 *   proposal y   in LDS, the MFMA B-operand layout [tile][k-step][lane]: every wave writes its slice, all read all of it
 *   F = A y      OUTPUT tiles are dealt to the waves in chunks of TC (chunk i to wave i mod W): a wave runs the FULL k
 *                loop for its chunk -- k ascending per accumulator: the bits of every other kernel -- with the matrix's
 *                fragments requested from L2 a block of k-steps ahead, each fragment feeding NT MFMAs
 *   sums         |F - Dbar|^2 and the prior join the chains' partial sums in the canonical order (outputs ascending,
 *                components ascending) by a hand-over from wave to wave through LDS: phase w of a round belongs to wave w
 *   decision     wave 0 (it carries log pi, the counters and the status of the chains) decides and posts the flags; every
 *                wave moves its slice and writes its part of the record row
 * Nothing but the record leaves the chip inside the launch.  Same draws, same arithmetic in the same order as the
 * launch-per-stage path: bit-identical (F22b, the d <= 1000 shapes and the wide fuzz cases run both).
 */
/* what an evaluation works with: the proposal tile and the hand-over slots in LDS, the wave's place in the workgroup */
struct WideEvalCtx {
    const double *ylds;        /* [NT][KSP][64] */
    const double *y_w;         /* ylds + this wave's first k-step + lane */
    double *handS, *handP;     /* [NT][64] each */
    volatile uint32_t *ticket; /* [2] */
    int KS, KSP, nq, lane, wave, g;
};

/* log-likelihood and prior sums of the NT tiles' proposals in LDS under ONE linear-model posterior (its matrix `blob`, MT
 * output tiles, constants bias / nprec / dbar in LDS, prior constants at this lane's first component): afterwards handS /
 * handP hold the chains' partial sums (per lane: its g), complete and visible to the LAST wave.  Every wave of the
 * workgroup calls this at the same point; tkP / tkS are the running ticket numbers (uniform, advanced here). */
template <int NT, int W, int QW, int TC, int KB, int PD>
__device__ __forceinline__ void wide_evaluate(const WideEvalCtx &cx, const double *__restrict__ blob, int MT, const double *bias,
                                              const double *nprec, const double *dbar, const double *pm_w, const double *pp_w,
                                              uint32_t &tkP, uint32_t &tkS)
{
    const int KS = cx.KS, KSP = cx.KSP, nq = cx.nq, lane = cx.lane, wave = cx.wave, g = cx.g;
    const double *ylds = cx.ylds, *y_w = cx.y_w;
    double *handS = cx.handS, *handP = cx.handP;
    volatile uint32_t *ticket = cx.ticket;
    const int n_chunks = (MT + TC - 1) / TC, rounds = (n_chunks + W - 1) / W;
    /* ---- the sums travel from wave to wave behind TICKETS in LDS (no barrier: a wave spins only for its predecessor,
     * which never waits for it): the prior's first -- components ascending, ymc_tile_logprior's order -- so that it
     * hides under the matrix products of the waves that are through ---- */
    auto wait_for = [&](int which, uint32_t value) {
        if (lane == 0)
            while (ticket[which] != value) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_wave_barrier();
    };
    {
        /* the terms first (they depend on nobody), the wait, then nothing but the additions */
        double pt[NT][2 * QW];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const double *yt = y_w + nt * KSP * 64;
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (qi < nq) {
                        const double dx = yt[(2 * qi + e) * 64] - pm_w[8 * qi + 4 * e];
                        const double px = pp_w[8 * qi + 4 * e] * dx;
                        pt[nt][2 * qi + e] = dx * px;
                    } else {
                        pt[nt][2 * qi + e] = 0.0;
                    }
                }
            }
        }
        const uint32_t tk = tkP + (uint32_t)wave; /* modulo 2^32: only equality is tested */
        double P[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) P[nt] = 0.0;
        if (wave > 0) {
            wait_for(0, tk);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) P[nt] = handP[nt * 64 + lane];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
                if (qi < nq) {
                    P[nt] = P[nt] + pt[nt][2 * qi];
                    P[nt] = P[nt] + pt[nt][2 * qi + 1];
                }
            }
            handP[nt * 64 + lane] = P[nt];
        }
        if (lane == 0) ticket[0] = tk + 1; /* LDS operations of a wave are performed in order: the sums are there */
    }

    /* ---- F = A y on the matrix cores ---- */
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const int chunk = r * W + wave, t0 = chunk * TC; /* this wave's output tiles t0 .. t0 + TC - 1 (those < MT) */
        d4 acc[NT][TC];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) acc[nt][tc] = d4{0.0, 0.0, 0.0, 0.0};
        if (t0 < MT) {
            const double *A = blob + (size_t)t0 * KS * 64 + lane;
            const double *yk = ylds + lane;
            /* a ring of PD stages of KB k-steps: stage i is multiplied while stages i + 1 .. i + PD - 1 are on their way
             * from L2.  The loop body is ONE basic block -- no guard on the k-step -- so that the compiler counts the
             * loads in flight (with guards it drains them all, `s_waitcnt vmcnt(0)`, at every iteration): the k-steps run
             * to KSP, KS rounded up to whole rings; beyond KS the proposal tile holds zeros and the fragment read is
             * whatever follows in the block (the next tile's, or the constants': finite), so the products are +0 */
            double fr[PD][KB][TC];
            auto request = [&](double (&f)[KB][TC], int s0) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const int s = s0 + kb < KSP ? s0 + kb : KSP - 1;
#pragma unroll
                    for (int tc = 0; tc < TC; ++tc) {
                        const int tt = t0 + tc < MT ? tc : 0; /* a chunk's tail repeats its first tile (never summed) */
                        f[kb][tc] = A[((size_t)tt * KS + s) * 64];
                    }
                }
            };
#pragma unroll
            for (int p = 0; p < PD; ++p) request(fr[p], p * KB);
#pragma unroll 1
            for (int s0 = 0; s0 < KSP; s0 += PD * KB) {
#pragma unroll
                for (int p = 0; p < PD; ++p) {
                    const int sb = s0 + p * KB;
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        double yb[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) yb[nt] = yk[(nt * KSP + sb + kb) * 64];
#pragma unroll
                        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[p][kb][tc], yb[nt], acc[nt][tc], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0); /* the stage's requests stay HERE, between its MFMAs and the next */
                    request(fr[p], sb + PD * KB);      /* stage's (the scheduler otherwise gathers all 32 at the loop's end) */
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        /* the terms of this wave's outputs IN PLACE of their accumulators (ymc_tile_add_residuals, one tile at a time)
         * and, in the first round, of its components' prior -- computed by all waves together, so that a phase of the
         * hand-over below is nothing but the additions */
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int tc = 0; tc < TC; ++tc)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int m = 16 * (t0 + tc < MT ? t0 + tc : 0) + 4 * rr + g;
                    const double F = acc[nt][tc][rr] + bias[m];
                    const double res = F - dbar[m];
                    const double px = nprec[m] * res;
                    acc[nt][tc][rr] = res * px;
                }
        /* the residuals' hand-over: outputs ascending, chunk r W + w follows chunk r W + w - 1 */
        {
            const uint32_t tk = tkS + (uint32_t)r * (uint32_t)W + (uint32_t)wave;
            if (chunk > 0) wait_for(1, tk);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                double S = (chunk == 0) ? 0.0 : handS[nt * 64 + lane];
#pragma unroll
                for (int tc = 0; tc < TC; ++tc) {
                    if (t0 + tc < MT) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) S = S + acc[nt][tc][rr];
                    }
                }
                handS[nt * 64 + lane] = S;
            }
            if (lane == 0) ticket[1] = tk + 1;
        }
    }

    tkP += (uint32_t)W;
    tkS += (uint32_t)rounds * (uint32_t)W;
}

template <int NT, int W, int QW, int TC, bool CG>
__global__ __launch_bounds__(64 * W) void linear_wide_fused_kernel(const WideFusedArgs a)
{
    ymc_stage_tables();
    extern __shared__ double lds[];
    constexpr int KB = TC >= 4 ? 1 : 4 / TC; /* k-steps per stage of fragments: ~4 fragments (more cost registers: spills) */
    constexpr int PD = W >= 16 ? 2 : 4;      /* stages in flight */
    const int KS = a.KS, MT = a.MT, m_pad = 16 * MT, d_pad = 4 * KS, d = a.d;
    const int KSP = ((KS + PD * KB - 1) / (PD * KB)) * (PD * KB); /* the k-steps rounded up to whole rings of stages */
    double *ylds = lds;                                   /* [NT][KSP][64], zero beyond KS */
    double *bias = ylds + (size_t)NT * KSP * 64;          /* bias | n prec | row mean | prior mean | prior prec | sigma */
    double *nprec = bias + m_pad, *dbar = nprec + m_pad;
    /* CG: the three per-component constants stay in the problem's block in global memory (L2): beyond ~900 parameters
     * they no longer fit the LDS next to the proposal tile */
    const double *pmean = CG ? a.blob + (size_t)MT * KS * 64 + 3 * (size_t)m_pad : dbar + m_pad;
    const double *pprec = pmean + d_pad, *sigma = pprec + d_pad;
    double *handS = dbar + m_pad + (CG ? 0 : 3 * d_pad);  /* [NT][64]: the residual sums on their way from wave to wave */
    double *handP = handS + NT * 64;                      /* [NT][64]: the prior's */
    int *movedf = reinterpret_cast<int *>(handP + NT * 64); /* [NT][16] */
    int *accf = movedf + NT * 16;                         /* [NT][16] */
    volatile uint32_t *ticket = reinterpret_cast<volatile uint32_t *>(accf + NT * 16);                /* [2]: hand-overs posted so far, of the prior's sums and of the residuals' */

    /* the wave's number as a SCALAR: its q-range, its chunk and its phase are uniform branches and scalar address
     * arithmetic (as tid >> 6 the compiler treats them as per-lane values: exec-masked branches around every pair's
     * draws and a vector register per hoisted LDS address) */
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4;
    {
        const double *consts = a.blob + (size_t)MT * KS * 64;
        for (int i = tid; i < 3 * m_pad + (CG ? 0 : 3 * d_pad); i += 64 * W) bias[i] = consts[i];
        for (int i = tid; i < NT * 16; i += 64 * W) movedf[i] = 0;
        if (tid < 2) ticket[tid] = 0;
        for (int i = tid; i < NT * (KSP - KS) * 64; i += 64 * W) {
            const int nt = i / ((KSP - KS) * 64), r = i % ((KSP - KS) * 64);
            ylds[(nt * KSP + KS) * 64 + r] = 0.0;
        }
    }
    const uint64_t nct = (a.K + 15) / 16;
    const uint64_t tile0 = (uint64_t)blockIdx.x * NT;
    const int Q = KS / 2, q0 = (wave * Q) / W, nq = ((wave + 1) * Q) / W - q0; /* this wave's q-range [q0, q0 + nq) */
    const uint32_t pair_of_g = (g == 0) ? 0u : (g == 1) ? 2u : (g == 2) ? 1u : 3u;
    /* everything the slice touches as ONE base per array + compile-time offsets (8 qi + 4 e components on, 2 qi + e k-steps
     * on): per-pair addresses computed from q would be a hundred loop-invariant values for the compiler to hoist and spill */
    const int j0 = 8 * q0 + g;                       /* this lane's first component */
    const double *sig_w = sigma + j0, *pm_w = pmean + j0, *pp_w = pprec + j0;
    double *y_w = ylds + (2 * q0) * 64 + lane;       /* + nt KSP 64 + (2 qi + e) 64 */
    const int left = d - j0 - 4;                      /* pairs qi with 8 qi < left have both their components below d */
    const int full_q = left > 0 ? (left + 7) / 8 : 0;

    uint64_t kk[NT];
    bool live[NT];
    double xs[NT][2 * QW];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const uint64_t k = (tile0 + nt) * 16 + c;
        live[nt] = tile0 + nt < nct && k < a.K;
        kk[nt] = live[nt] ? k : 0;
        const double *xr = a.x + kk[nt] * d + j0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                xs[nt][2 * qi + e] = (live[nt] && qi < nq && j0 + 8 * qi + 4 * e < d) ? xr[8 * qi + 4 * e] : 0.0;
        }
    }
    /* the LAST wave (the hand-overs end with it: it decides): what a chain carries besides its state */
    double lp[NT];
    uint64_t accepted[NT];
    uint32_t st[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        lp[nt] = (wave == W - 1 && live[nt]) ? a.lp[kk[nt]] : 0.0;
        accepted[nt] = (wave == W - 1 && live[nt]) ? a.acc_count[kk[nt]] : 0;
        st[nt] = (wave == W - 1 && live[nt]) ? a.status[kk[nt]] : 0;
    }
    RecordClock clk(a.rec_stride, a.rec_since);
    /* The accumulators are carried by a run-time k loop (KS is not a template parameter here), so the kernel must stay
     * within the VGPR-form MFMA budget: built for one wave per SIMD (512 registers) the compiler puts them in AGPRs and
     * copies every accumulator in and out per iteration (500 spilled registers at NT = 2, TC = 8); at two waves per SIMD
     * everything is a VGPR */
    __syncthreads();
    const WideEvalCtx cx{ylds, y_w, handS, handP, ticket, KS, KSP, nq, lane, wave, g};
    uint32_t tkP = 0, tkS = 0;

#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t t = a.t0 + n;
        /* ---- proposal (mrw.py:32-38 / pcn.py:27-36) of this wave's slice, into LDS ---- */
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint32_t chain = (uint32_t)(a.chain0 + kk[nt]);
            double *yt = y_w + nt * KSP * 64;
            bool moved_l = false;
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
                if (qi < nq) {
                    double xi0, xi1;
#if defined(YMC_WF_NODRAW)   /* timing experiments only */
                    xi0 = 1e-3 * (double)(qi + 1), xi1 = -1e-3 * (double)(chain & 7u);
#else
                    ymc_draw_normal_pair(a.seed, chain, t, 4u * (uint32_t)(q0 + qi) + pair_of_g, &xi0, &xi1);
#endif
                    swap_odd_even_rows(xi0, xi1);
                    const double x0 = xs[nt][2 * qi], x1 = xs[nt][2 * qi + 1];
                    double y0, y1;
                    if (a.pcn) {
                        const double p0 = pm_w[8 * qi] + sig_w[8 * qi] * xi0;
                        const double p1 = pm_w[8 * qi + 4] + sig_w[8 * qi + 4] * xi1;
                        y0 = a.pcn_c1 * x0 + a.pcn_c2 * p0;
                        y1 = a.pcn_c1 * x1 + a.pcn_c2 * p1;
                    } else {
                        y0 = x0 + sig_w[8 * qi] * xi0;
                        y1 = x1 + sig_w[8 * qi + 4] * xi1;
                    }
                    yt[(2 * qi) * 64] = y0;
                    yt[(2 * qi + 1) * 64] = y1;
                    moved_l = moved_l || !(y0 == x0) || !(y1 == x1);
                }
                __builtin_amdgcn_sched_barrier(0); /* one pair's draws at a time */
            }
            int mv = moved_l ? 1 : 0;
            mv |= __shfl_xor(mv, 16, 64);
            mv |= __shfl_xor(mv, 32, 64);
            if (mv && g == 0) movedf[nt * 16 + c] = 1; /* every writer stores the same 1 */
        }
        __syncthreads();

        /* ---- the evaluation: prior and residual sums handed from wave to wave behind tickets, F = A y on the matrix cores ---- */
        wide_evaluate<NT, W, QW, TC, KB, PD>(cx, a.blob, MT, bias, nprec, dbar, pm_w, pp_w, tkP, tkS);

        /* ---- the decision (metropolisHastings.py:55-73, mrw.py:51-57): the last wave, whose hand-over closed the sums ---- */
        const bool rec = clk.due();
        if (wave == W - 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const double logl = -0.5 * (combine4(handS[nt * 64 + lane]) + a.lik_const);
                const double lpy = a.likelihood_only ? logl : logl + (-0.5 * combine4(handP[nt * 64 + lane]));
                const uint32_t chain = (uint32_t)(a.chain0 + kk[nt]);
                const double u = ymc_draw_uniform(a.seed, chain, t);
                const bool moved = movedf[nt * 16 + c] != 0;
                const bool acc1 = ymc_accept(lpy - lp[nt], moved, ymc_finite(lpy), u, st[nt]);
                lp[nt] = acc1 ? lpy : lp[nt];
                accepted[nt] += acc1 ? 1u : 0u;
                if (g == 0) {
                    accf[nt * 16 + c] = acc1 ? 1 : 0;
                    movedf[nt * 16 + c] = 0;
                    if (live[nt]) {
                        if (a.acc) a.acc[(uint64_t)n * a.K + kk[nt]] = acc1 ? 1 : 0;
                        if (rec && a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + kk[nt]] = lp[nt];
                    }
                }
            }
        }
        __syncthreads();
        /* ---- the move and the record row, every wave its slice ---- */
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool acc1 = accf[nt * 16 + c] != 0;
            const double *yt = y_w + nt * KSP * 64;
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
                if (qi < nq) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const double y = yt[(2 * qi + e) * 64];
                        xs[nt][2 * qi + e] = acc1 ? y : xs[nt][2 * qi + e];
                    }
                }
            }
            if (rec && a.traj && live[nt]) { /* ONE masked region: the chains that exist */
                double *row = a.traj + ((uint64_t)clk.row * a.K + kk[nt]) * (uint64_t)d + j0;
#pragma unroll
                for (int qi = 0; qi < QW; ++qi) {
                    if (qi < nq) {
                        if (qi < full_q) { /* both components exist in every lane's row */
                            row[8 * qi] = xs[nt][2 * qi];
                            row[8 * qi + 4] = xs[nt][2 * qi + 1];
                        } else {
                            if (j0 + 8 * qi < d) row[8 * qi] = xs[nt][2 * qi];
                            if (j0 + 8 * qi + 4 < d) row[8 * qi + 4] = xs[nt][2 * qi + 1];
                        }
                    }
                }
            }
        }
        if (rec) clk.wrote();
        /* no barrier here: the next proposal writes this wave's OWN slice of the LDS tile, which nobody else reads any more
         * (every wave is through its products: the decision's barrier), and the flags it touches were read before that
         * barrier */
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (!live[nt]) continue;
        double *xr = a.x + kk[nt] * d + j0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (qi < nq && j0 + 8 * qi + 4 * e < d) xr[8 * qi + 4 * e] = xs[nt][2 * qi + e];
        }
        if (wave == W - 1 && g == 0) {
            a.lp[kk[nt]] = lp[nt];
            a.acc_count[kk[nt]] = accepted[nt];
            a.status[kk[nt]] = st[nt];
        }
    }
}

/*
 * Two-level delayed acceptance on the same plan.  One tile of 16 chains per workgroup: the fine state x and the
 * sub-chain's state z are two slices of registers per wave; a sub-step proposes from z into the LDS tile, evaluates the
 * COARSE posterior and moves z; the fine step writes z into the tile, evaluates the FINE posterior and decides with
 * ((pi_f(z) + pi_c(x)) - pi_c(z)) - pi_f(x) (mlda.py:146-154).  Fine step n owns transitions n (S + 1) + s, as everywhere.
 * The three per-component constant arrays of each level stay in global memory (L2) when they do not fit the LDS.
 */
template <int W, int QW, int TC, bool CG>
__global__ __launch_bounds__(64 * W) void linear_wide_fused2_kernel(const WideFused2Args a)
{
    ymc_stage_tables();
    extern __shared__ double lds[];
    constexpr int NT = 1;
    constexpr int KB = TC >= 4 ? 1 : 4 / TC;
    constexpr int PD = W >= 16 ? 2 : 4;
    const int KS = a.KS, d_pad = 4 * KS, d = a.d, S = a.S;
    const int MT_c = a.MT_c, MT_f = a.MT_f, mc_pad = 16 * MT_c, mf_pad = 16 * MT_f;
    const int KSP = ((KS + PD * KB - 1) / (PD * KB)) * (PD * KB);
    double *ylds = lds;                                   /* [KSP][64], zero beyond KS */
    double *bias_c = ylds + (size_t)KSP * 64;             /* coarse: bias | n prec | row mean */
    double *nprec_c = bias_c + mc_pad, *dbar_c = nprec_c + mc_pad;
    double *bias_f = dbar_c + mc_pad;                     /* fine: bias | n prec | row mean */
    double *nprec_f = bias_f + mf_pad, *dbar_f = nprec_f + mf_pad;
    double *dconst = dbar_f + mf_pad;                     /* !CG: coarse prior mean | prec | sigma | fine prior mean | prec */
    const double *cc = a.blob_c + (size_t)MT_c * KS * 64, *fc = a.blob_f + (size_t)MT_f * KS * 64;
    const double *pmean_c = CG ? cc + 3 * (size_t)mc_pad : dconst;
    const double *pprec_c = pmean_c + d_pad, *sigma = pprec_c + d_pad;
    const double *pmean_f = CG ? fc + 3 * (size_t)mf_pad : dconst + 3 * d_pad;
    const double *pprec_f = pmean_f + d_pad;
    double *handS = dconst + (CG ? 0 : 5 * d_pad);
    double *handP = handS + 64;
    int *movedf = reinterpret_cast<int *>(handP + 64);    /* [16] */
    int *accf = movedf + 16;                              /* [16] */
    volatile uint32_t *ticket = reinterpret_cast<volatile uint32_t *>(accf + 16);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4;
    {
        for (int i = tid; i < 3 * mc_pad; i += 64 * W) bias_c[i] = cc[i];
        for (int i = tid; i < 3 * mf_pad; i += 64 * W) bias_f[i] = fc[i];
        if (!CG) {
            for (int i = tid; i < 3 * d_pad; i += 64 * W) dconst[i] = cc[3 * (size_t)mc_pad + i];
            for (int i = tid; i < 2 * d_pad; i += 64 * W) dconst[3 * d_pad + i] = fc[3 * (size_t)mf_pad + i];
        }
        for (int i = tid; i < 16; i += 64 * W) movedf[i] = 0;
        if (tid < 2) ticket[tid] = 0;
        for (int i = tid; i < (KSP - KS) * 64; i += 64 * W) ylds[KS * 64 + i] = 0.0;
    }
    const uint64_t nct = (a.K + 15) / 16, tile = blockIdx.x;
    const int Q = KS / 2, q0 = (wave * Q) / W, nq = ((wave + 1) * Q) / W - q0;
    const uint32_t pair_of_g = (g == 0) ? 0u : (g == 1) ? 2u : (g == 2) ? 1u : 3u;
    const int j0 = 8 * q0 + g;
    const double *sig_w = sigma + j0;
    double *y_w = ylds + (2 * q0) * 64 + lane;
    const int left = d - j0 - 4;
    const int full_q = left > 0 ? (left + 7) / 8 : 0;

    const uint64_t k = tile * 16 + c;
    const bool live = tile < nct && k < a.K;
    const uint64_t kk = live ? k : 0;
    const uint32_t chain = (uint32_t)(a.chain0 + kk);
    double xs[2 * QW], zs[2 * QW];
    {
        const double *xr = a.x + kk * d + j0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                xs[2 * qi + e] = (live && qi < nq && j0 + 8 * qi + 4 * e < d) ? xr[8 * qi + 4 * e] : 0.0;
    }
    /* the last wave: what a chain carries besides its state */
    const bool decider = wave == W - 1;
    double lpf = (decider && live) ? a.lp[kk] : 0.0, lpc = (decider && live) ? a.lpc[kk] : 0.0;
    uint64_t accepted = (decider && live) ? a.acc_count[kk] : 0;
    uint32_t st = (decider && live) ? a.status[kk] : 0;
    RecordClock clk(a.rec_stride, a.rec_since);
    __syncthreads();
    const WideEvalCtx cx{ylds, y_w, handS, handP, ticket, KS, KSP, nq, lane, wave, g};
    uint32_t tkP = 0, tkS = 0;
    auto any_moved = [&](bool moved_l) { /* the chain's flag from the lanes' (every writer stores the same 1) */
        int mv = moved_l ? 1 : 0;
        mv |= __shfl_xor(mv, 16, 64);
        mv |= __shfl_xor(mv, 32, 64);
        if (mv && g == 0) movedf[c] = 1;
    };

#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t tbase = (a.t0 + n) * (uint64_t)(S + 1);
#pragma unroll
        for (int i = 0; i < 2 * QW; ++i) zs[i] = xs[i];
        double lpcz = lpc;
        /* surrogate.run(S + 1, state), mlda.py:102-110: S random-walk transitions of z on the coarse posterior */
#pragma unroll 1
        for (int sub = 0; sub < S; ++sub) {
            const uint64_t t = tbase + (uint64_t)sub;
            bool moved_l = false;
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
                if (qi < nq) {
                    double xi0, xi1;
                    ymc_draw_normal_pair(a.seed, chain, t, 4u * (uint32_t)(q0 + qi) + pair_of_g, &xi0, &xi1);
                    swap_odd_even_rows(xi0, xi1);
                    const double y0 = zs[2 * qi] + sig_w[8 * qi] * xi0;
                    const double y1 = zs[2 * qi + 1] + sig_w[8 * qi + 4] * xi1;
                    y_w[(2 * qi) * 64] = y0;
                    y_w[(2 * qi + 1) * 64] = y1;
                    moved_l = moved_l || !(y0 == zs[2 * qi]) || !(y1 == zs[2 * qi + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            any_moved(moved_l);
            __syncthreads();
            wide_evaluate<NT, W, QW, TC, KB, PD>(cx, a.blob_c, MT_c, bias_c, nprec_c, dbar_c, pmean_c + j0, pprec_c + j0, tkP, tkS);
            if (decider) {
                const double logl = -0.5 * (combine4(handS[lane]) + a.lik_const_c);
                const double lpcy = logl + (-0.5 * combine4(handP[lane]));
                const double u = ymc_draw_uniform(a.seed, chain, t);
                const bool moved = movedf[c] != 0;
                const bool acc1 = ymc_accept(lpcy - lpcz, moved, ymc_finite(lpcy), u, st);
                lpcz = acc1 ? lpcy : lpcz;
                if (g == 0) {
                    accf[c] = acc1 ? 1 : 0;
                    movedf[c] = 0;
                    if (live && a.sub_acc) a.sub_acc[((uint64_t)n * S + sub) * a.K + kk] = acc1 ? 1 : 0;
                }
            }
            __syncthreads();
            {
                const bool acc1 = accf[c] != 0;
#pragma unroll
                for (int qi = 0; qi < QW; ++qi) {
                    if (qi < nq) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const double y = y_w[(2 * qi + e) * 64];
                            zs[2 * qi + e] = acc1 ? y : zs[2 * qi + e];
                        }
                    }
                }
            }
        }
        /* the fine accept / reject, mlda.py:146-154: z into the tile, the fine posterior of z */
        {
            bool moved_l = false;
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
                if (qi < nq) {
                    y_w[(2 * qi) * 64] = zs[2 * qi];
                    y_w[(2 * qi + 1) * 64] = zs[2 * qi + 1];
                    moved_l = moved_l || !(zs[2 * qi] == xs[2 * qi]) || !(zs[2 * qi + 1] == xs[2 * qi + 1]);
                }
            }
            any_moved(moved_l);
        }
        __syncthreads();
        wide_evaluate<NT, W, QW, TC, KB, PD>(cx, a.blob_f, MT_f, bias_f, nprec_f, dbar_f, pmean_f + j0, pprec_f + j0, tkP, tkS);
        const bool rec = clk.due();
        if (decider) {
            const double logl = -0.5 * (combine4(handS[lane]) + a.lik_const_f);
            const double lpfz = logl + (-0.5 * combine4(handP[lane]));
            const double u = ymc_draw_uniform(a.seed, chain, tbase + (uint64_t)S);
            const bool moved = movedf[c] != 0;
            const double log_ratio = ((lpfz + lpc) - lpcz) - lpf;
            const bool acc1 = ymc_accept(log_ratio, moved, ymc_finite(lpfz) && ymc_finite(lpcz), u, st);
            lpf = acc1 ? lpfz : lpf;
            lpc = acc1 ? lpcz : lpc;
            accepted += acc1 ? 1u : 0u;
            if (g == 0) {
                accf[c] = acc1 ? 1 : 0;
                movedf[c] = 0;
                if (live) {
                    if (a.acc) a.acc[(uint64_t)n * a.K + kk] = acc1 ? 1 : 0;
                    if (rec && a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + kk] = lpf;
                }
            }
        }
        __syncthreads();
        {
            const bool acc1 = accf[c] != 0;
#pragma unroll
            for (int i = 0; i < 2 * QW; ++i) xs[i] = acc1 ? zs[i] : xs[i];
            if (rec && a.traj && live) {
                double *row = a.traj + ((uint64_t)clk.row * a.K + kk) * (uint64_t)d + j0;
#pragma unroll
                for (int qi = 0; qi < QW; ++qi) {
                    if (qi < nq) {
                        if (qi < full_q) {
                            row[8 * qi] = xs[2 * qi];
                            row[8 * qi + 4] = xs[2 * qi + 1];
                        } else {
                            if (j0 + 8 * qi < d) row[8 * qi] = xs[2 * qi];
                            if (j0 + 8 * qi + 4 < d) row[8 * qi + 4] = xs[2 * qi + 1];
                        }
                    }
                }
            }
        }
        if (rec) clk.wrote();
    }

    if (live) {
        double *xr = a.x + kk * d + j0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (qi < nq && j0 + 8 * qi + 4 * e < d) xr[8 * qi + 4 * e] = xs[2 * qi + e];
        if (decider && g == 0) {
            a.lp[kk] = lpf;
            a.lpc[kk] = lpc;
            a.acc_count[kk] = accepted;
            a.status[kk] = st;
        }
    }
}

}  // namespace

/* FullDiagnostics' moments (estimation.py:36-53, diagnostics.py:91-94) for the fused kernel: replayed after the launch from
 * the plain record [rows][K][d] it wrote, one thread per chain and component -- the online update sequence per component,
 * hence the bits of the kernels that update in place (welford_replay_kernel is the same for the tiled record).  The state
 * before the launch's first transition is the carried state saved before the launch. */
namespace {
__global__ __launch_bounds__(256) void welford_replay_plain_kernel(const double *__restrict__ traj, const double *__restrict__ x_before,
                                                                   uint64_t rows, uint64_t per_row, uint64_t wf_n0,
                                                                   double *__restrict__ wf_mean, double *__restrict__ wf_m2)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_row) return;
    double mean = wf_mean[i], m2 = wf_m2[i];
    double v = x_before[i];
    auto step = [&](uint64_t n, double next) {
        const double dn = (double)(wf_n0 + n + 1);
        const double delta = v - mean;
        mean = mean + delta / dn;
        const double delta2 = v - mean;
        m2 = m2 + delta * delta2;
        v = next;
    };
    uint64_t n = 0;
    for (; n + 8 <= rows; n += 8) { /* eight rows requested together: the updates are sequential, the loads need not be */
        double nx[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) nx[q] = traj[(n + q) * per_row + i];
#pragma unroll
        for (int q = 0; q < 8; ++q) step(n + q, nx[q]);
    }
    for (; n < rows; ++n) step(n, traj[n * per_row + i]);
    wf_mean[i] = mean;
    wf_m2[i] = m2;
}
}  // namespace

hipError_t ymc_launch_welford_replay_plain(const double *traj, const double *x_before, uint64_t rows, uint64_t K, int d,
                                           uint64_t wf_n0, double *wf_mean, double *wf_m2, hipStream_t s)
{
    const uint64_t per_row = K * (uint64_t)d;
    hipLaunchKernelGGL(welford_replay_plain_kernel, dim3((unsigned)((per_row + 255) / 256)), dim3(256), 0, s, traj, x_before, rows,
                       per_row, wf_n0, wf_mean, wf_m2);
    return hipGetLastError();
}

/* which build of linear_wide_fused_kernel an ensemble gets: large ensembles two tiles per workgroup of eight waves (two per
 * SIMD, every fragment of the matrix feeding two MFMAs; one tile beyond 576 parameters, where two do not fit the LDS),
 * small ones one tile per workgroup of sixteen (the chip is short of waves, not of registers); TC output tiles per wave
 * and round by the number of output tiles */
namespace {
struct FusedPick {
    int NT, W, QW, TC;
    bool CG; /* the per-component constants from global memory */
};

bool pick_fused(int KS, int MT, uint64_t K, int cus, FusedPick &f)
{
    const uint64_t nct = (K + 15) / 16;
    const int Q = KS / 2;
    if (nct >= 2ull * (uint64_t)cus) { /* eight waves = two per SIMD, 256 registers each */
        f.W = 8;
        f.NT = Q <= 72 ? 2 : 1;
        f.QW = f.NT == 2 ? 9 : 18;
    } else {                           /* sixteen waves = four per SIMD, 128 registers */
        f.W = 16, f.NT = 1, f.QW = 9;
    }
    if ((Q + f.W - 1) / f.W > f.QW) return false;
    const int per_wave = (MT + f.W - 1) / f.W;
    const int cap = f.W == 16 ? 1 : 2; /* more output tiles per wave and round spill (the accumulators + the ring of fragments) */
    f.TC = per_wave >= 8 ? 8 : per_wave >= 4 ? 4 : per_wave >= 2 ? 2 : 1;
    if (f.TC > cap) f.TC = cap;
    if (const char *env = std::getenv("YMC_WIDE_TC")) { /* experiments: tiles per wave and round */
        const int want = std::atoi(env);
        if ((want == 1 || want == 2 || want == 4 || want == 8) && want <= cap) f.TC = want;
    }
    return true;
}

size_t fused_lds(const FusedPick &f, int KS, int MT)
{
    const size_t m_pad = 16 * (size_t)MT, d_pad = 4 * (size_t)KS;
    const int kb = f.TC >= 4 ? 1 : 4 / f.TC, pd = f.W >= 16 ? 2 : 4, ring = kb * pd;
    const size_t ksp = (size_t)((KS + ring - 1) / ring) * ring; /* as the kernel rounds its k loop */
    return ((size_t)f.NT * ksp * 64 + 3 * m_pad + (f.CG ? 0 : 3 * d_pad) + 2 * (size_t)f.NT * 64) * sizeof(double) +
           (2 * (size_t)f.NT * 16 + 4) * sizeof(int);
}

/* the build and its LDS: constants in LDS if everything fits, else the per-component ones from global memory; 0 = no */
size_t fused_fit(int KS, int MT, uint64_t K, int cus, FusedPick &f)
{
    if (!pick_fused(KS, MT, K, cus, f)) return 0;
    f.CG = false;
    size_t lds = fused_lds(f, KS, MT);
    if (lds > kTileLdsLimit) {
        f.CG = true;
        lds = fused_lds(f, KS, MT);
    }
    return lds <= kTileLdsLimit ? lds : 0;
}

template <int NT, int W, int QW, int TC, bool CG>
hipError_t launch_fused(const WideFusedArgs &a, size_t lds, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_wide_fused_kernel<NT, W, QW, TC, CG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t nct = (a.K + 15) / 16;
    hipLaunchKernelGGL((linear_wide_fused_kernel<NT, W, QW, TC, CG>), dim3((unsigned)((nct + NT - 1) / NT)), dim3(64 * W), lds, s, a);
    return hipGetLastError();
}
}  // namespace

size_t ymc_wide_fused_lds_bytes(int KS, int MT, uint64_t K, int cus)
{
    if (const char *env = std::getenv("YMC_WIDE_FUSED"))
        if (env[0] == '0') return 0;
    FusedPick f;
    return fused_fit(KS, MT, K, cus, f);
}

hipError_t ymc_launch_wide_fused(const WideFusedArgs &a, int cus, hipStream_t s)
{
    FusedPick f;
    const size_t lds = fused_fit(a.KS, a.MT, a.K, cus, f);
    if (!lds) return hipErrorInvalidValue;
#define YMC_FUSED(NT_, W_, QW_, TC_)                                                                      \
    if (f.NT == NT_ && f.W == W_ && f.TC == TC_)                                                          \
        return f.CG ? launch_fused<NT_, W_, QW_, TC_, true>(a, lds, s) : launch_fused<NT_, W_, QW_, TC_, false>(a, lds, s);
    YMC_FUSED(2, 8, 9, 2) YMC_FUSED(2, 8, 9, 1)
    YMC_FUSED(1, 8, 18, 2) YMC_FUSED(1, 8, 18, 1)
    YMC_FUSED(1, 16, 9, 1)
#undef YMC_FUSED
    return hipErrorInvalidValue;
}

/* ---- two-level ---- */
namespace {
struct Fused2Pick {
    int W, QW, TC;
    bool CG;
};

size_t fused2_lds(const Fused2Pick &f, int KS, int MT_c, int MT_f)
{
    const size_t mc = 16 * (size_t)MT_c, mf = 16 * (size_t)MT_f, d_pad = 4 * (size_t)KS;
    const int kb = f.TC >= 4 ? 1 : 4 / f.TC, pd = f.W >= 16 ? 2 : 4, ring = kb * pd;
    const size_t ksp = (size_t)((KS + ring - 1) / ring) * ring;
    return (ksp * 64 + 3 * mc + 3 * mf + (f.CG ? 0 : 5 * d_pad) + 2 * 64) * sizeof(double) + (2 * 16 + 4) * sizeof(int);
}

size_t fused2_fit(int KS, int MT_c, int MT_f, uint64_t K, int cus, Fused2Pick &f)
{
    const uint64_t nct = (K + 15) / 16;
    const int Q = KS / 2;
    if (nct >= (uint64_t)cus) f.W = 8, f.QW = (Q + 7) / 8 <= 9 ? 9 : 18; /* two slices of state: the smaller one where it does */
    else f.W = 16, f.QW = (Q + 15) / 16 <= 5 ? 5 : 9;
    if ((Q + f.W - 1) / f.W > f.QW) return 0;
    const int per_wave = ((MT_f > MT_c ? MT_f : MT_c) + f.W - 1) / f.W;
    f.TC = f.W == 16 ? 1 : (per_wave >= 2 ? 2 : 1);
    f.CG = false;
    size_t lds = fused2_lds(f, KS, MT_c, MT_f);
    if (lds > kTileLdsLimit) {
        f.CG = true;
        lds = fused2_lds(f, KS, MT_c, MT_f);
    }
    return lds <= kTileLdsLimit ? lds : 0;
}

template <int W, int QW, int TC, bool CG>
hipError_t launch_fused2(const WideFused2Args &a, size_t lds, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_wide_fused2_kernel<W, QW, TC, CG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t nct = (a.K + 15) / 16;
    hipLaunchKernelGGL((linear_wide_fused2_kernel<W, QW, TC, CG>), dim3((unsigned)nct), dim3(64 * W), lds, s, a);
    return hipGetLastError();
}
}  // namespace

size_t ymc_wide_fused2_lds_bytes(int KS, int MT_c, int MT_f, uint64_t K, int cus)
{
    if (const char *env = std::getenv("YMC_WIDE_FUSED"))
        if (env[0] == '0') return 0;
    Fused2Pick f;
    return fused2_fit(KS, MT_c, MT_f, K, cus, f);
}

hipError_t ymc_launch_wide_fused2(const WideFused2Args &a, int cus, hipStream_t s)
{
    Fused2Pick f;
    const size_t lds = fused2_fit(a.KS, a.MT_c, a.MT_f, a.K, cus, f);
    if (!lds) return hipErrorInvalidValue;
#define YMC_FUSED2(W_, QW_, TC_)                     \
    if (f.W == W_ && f.QW == QW_ && f.TC == TC_)     \
        return f.CG ? launch_fused2<W_, QW_, TC_, true>(a, lds, s) : launch_fused2<W_, QW_, TC_, false>(a, lds, s);
    YMC_FUSED2(8, 9, 2) YMC_FUSED2(8, 9, 1) YMC_FUSED2(8, 18, 2) YMC_FUSED2(8, 18, 1) YMC_FUSED2(16, 5, 1) YMC_FUSED2(16, 9, 1)
#undef YMC_FUSED2
    return hipErrorInvalidValue;
}
