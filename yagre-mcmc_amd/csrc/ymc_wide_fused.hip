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

/* a wave spins only for its predecessor, which never waits for it (lane 0 polls, the wave follows) */
__device__ __forceinline__ void wide_wait_ticket(volatile uint32_t *ticket, int which, uint32_t value, int lane)
{
    if (lane == 0)
        while (ticket[which] != value) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_wave_barrier();
}

/* The prior's sums of the NT tiles' proposals in LDS, travelling from wave to wave behind TICKETS in LDS (no barrier) --
 * components ascending, ymc_tile_logprior's order -- so that they hide under the matrix products of the waves that are
 * through: afterwards handP holds the chains' partial sums, complete and visible to the LAST wave.  pm_w / pp_w: the prior's
 * mean and precision at this lane's first component. */
template <int NT, int W, int QW>
__device__ __forceinline__ void wide_prior_chain(const WideEvalCtx &cx, const double *pm_w, const double *pp_w, uint32_t &tkP)
{
    const int KSP = cx.KSP, nq = cx.nq, lane = cx.lane, wave = cx.wave;
    const double *y_w = cx.y_w;
    double *handP = cx.handP;
    volatile uint32_t *ticket = cx.ticket;
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
        wide_wait_ticket(ticket, 0, tk, lane);
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
    tkP += (uint32_t)W;
}

/* F = A y on the matrix cores for the TC output tiles from t0 (those below MT) of the NT tiles' proposals in LDS: the FULL k
 * loop, k ascending per accumulator. */
template <int NT, int TC, int KB, int PD>
__device__ __forceinline__ void wide_products(const WideEvalCtx &cx, const double *__restrict__ blob, int MT, int t0, d4 (&acc)[NT][TC],
                                              int k_end = 0)
{
    /* k_end: whole rings of k-steps to run (0 = all KSP): a lower-triangular factor's rows have nothing but zeros beyond their
     * diagonal block, and a zero product added to an accumulator changes no bit of it */
    const int KS = cx.KS, KSP = k_end ? k_end : cx.KSP, lane = cx.lane;
    const double *A = blob + (size_t)t0 * KS * 64 + lane;
    const double *yk = cx.ylds + lane;
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

/* the terms of the outputs IN PLACE of their accumulators (ymc_tile_add_residuals, one tile at a time) */
template <int NT, int TC>
__device__ __forceinline__ void wide_terms(d4 (&acc)[NT][TC], int MT, int t0, int g, const double *bias, const double *nprec,
                                           const double *dbar)
{
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
}

/* log-likelihood and prior sums of the NT tiles' proposals in LDS under ONE linear-model posterior (its matrix `blob`, MT
 * output tiles, constants bias / nprec / dbar in LDS, prior constants at this lane's first component): afterwards handS /
 * handP hold the chains' partial sums (per lane: its g), complete and visible to the LAST wave.  Every wave of the
 * workgroup calls this at the same point; tkP / tkS are the running ticket numbers (uniform, advanced here). */
template <int NT, int W, int QW, int TC, int KB, int PD>
__device__ __forceinline__ void wide_evaluate(const WideEvalCtx &cx, const double *__restrict__ blob, int MT, const double *bias,
                                              const double *nprec, const double *dbar, const double *pm_w, const double *pp_w,
                                              uint32_t &tkP, uint32_t &tkS)
{
    const int lane = cx.lane, wave = cx.wave, g = cx.g;
    double *handS = cx.handS;
    volatile uint32_t *ticket = cx.ticket;
    const int n_chunks = (MT + TC - 1) / TC, rounds = (n_chunks + W - 1) / W;
#if !defined(YMC_WF_NOPRIOR)
    wide_prior_chain<NT, W, QW>(cx, pm_w, pp_w, tkP);
#endif

    /* ---- F = A y on the matrix cores ---- */
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        const int chunk = r * W + wave, t0 = chunk * TC; /* this wave's output tiles t0 .. t0 + TC - 1 (those < MT) */
        d4 acc[NT][TC];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) acc[nt][tc] = d4{0.0, 0.0, 0.0, 0.0};
#if !defined(YMC_WF_NOPROD)   /* timing experiments only (tools/exp_wide_stages.sh) */
        if (t0 < MT) wide_products<NT, TC, KB, PD>(cx, blob, MT, t0, acc);
#endif
        /* computed by all waves together, so that a phase of the hand-over below is nothing but the additions */
        wide_terms<NT, TC>(acc, MT, t0, g, bias, nprec, dbar);
        /* the residuals' hand-over: outputs ascending, chunk c follows chunk c - 1; a wave whose chunk holds no output tile
         * takes no part (with 64 outputs on eight waves that is half the hops) */
        if (chunk < n_chunks) {
            const uint32_t tk = tkS + (uint32_t)chunk;
            if (chunk > 0) wide_wait_ticket(ticket, 1, tk, lane);
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
    /* the LAST wave reads the sums next (it decides): unless the last chunk was its own it waits for that chunk's wave */
    if (wave == W - 1 && rounds * W > n_chunks) wide_wait_ticket(ticket, 1, tkS + (uint32_t)n_chunks, lane);
    tkS += (uint32_t)n_chunks;
}

/* DENSE: the random walk with a dense proposal covariance (covariance.py:91-92), y = x + 1 (L xi): the normals go into a
 * second LDS tile, L xi is one more product on the matrix cores (the factor's rows as output tiles, dealt to the waves like
 * the forward map's; an output tile's accumulator IS the B-operand fragment of its four k-steps: lane (c, g), register rr
 * = row 16 t + 4 rr + g = k-step 4 t + rr, component g), the slices' owners add their state. */
template <int NT, int W, int QW, int TC, bool CG, bool DENSE = false>
__global__ __launch_bounds__(64 * W) void linear_wide_fused_kernel(const WideFusedArgs a)
{
    static_assert(!DENSE || NT == 1, "the dense proposal's build holds one tile");
    ymc_stage_tables();
    extern __shared__ double lds[];
    constexpr int KB = TC >= 4 ? 1 : 4 / TC; /* k-steps per stage of fragments: ~4 fragments (more cost registers: spills) */
    constexpr int PD = W >= 16 ? 2 : 4;      /* stages in flight */
    const int KS = a.KS, MT = a.MT, m_pad = 16 * MT, d_pad = 4 * KS, d = a.d;
    const int KSP = ((KS + PD * KB - 1) / (PD * KB)) * (PD * KB); /* the k-steps rounded up to whole rings of stages */
    double *ylds = lds;                                   /* [NT][KSP][64], zero beyond KS */
    double *xilds = ylds + (size_t)NT * KSP * 64;         /* DENSE: [KSP][64], the normals, zero beyond KS */
    double *bias = xilds + (DENSE ? (size_t)KSP * 64 : 0); /* bias | n prec | row mean | prior mean | prior prec | sigma */
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
        if (DENSE)
            for (int i = tid; i < (KSP - KS) * 64; i += 64 * W) xilds[KS * 64 + i] = 0.0;
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
                    if (DENSE) { /* the normals into their tile; the proposal follows the product below */
                        double *xw = xilds + (2 * q0) * 64 + lane;
                        xw[(2 * qi) * 64] = xi0;
                        xw[(2 * qi + 1) * 64] = xi1;
                        continue;
                    }
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
            if (!DENSE) {
                int mv = moved_l ? 1 : 0;
                mv |= __shfl_xor(mv, 16, 64);
                mv |= __shfl_xor(mv, 32, 64);
                if (mv && g == 0) movedf[nt * 16 + c] = 1; /* every writer stores the same 1 */
            }
        }
        __syncthreads();
        if constexpr (DENSE) {
            /* ---- L xi on the matrix cores, into the proposal tile in the B layout ---- */
            {
                const WideEvalCtx cl{xilds, xilds + (2 * q0) * 64 + lane, nullptr, nullptr, ticket, KS, KSP, nq, lane, wave, g};
                const int MT_L = a.MT_L, n_chunks = (MT_L + TC - 1) / TC, rounds = (n_chunks + W - 1) / W;
#pragma unroll 1
                for (int r = 0; r < rounds; ++r) {
                    const int t0 = (r * W + wave) * TC;
                    if (t0 >= MT_L) break;
                    d4 acc[1][TC];
#pragma unroll
                    for (int tc = 0; tc < TC; ++tc) acc[0][tc] = d4{0.0, 0.0, 0.0, 0.0};
                    /* the rows of output tiles t0 .. t0 + TC - 1 end with column 16 (t0 + TC) - 1 = k-step 4 (t0 + TC) - 1 */
                    const int ring = PD * KB, k_need = 4 * (t0 + TC), k_run = ((k_need + ring - 1) / ring) * ring;
                    wide_products<1, TC, KB, PD>(cl, a.chol_frag, MT_L, t0, acc, k_run < KSP ? k_run : KSP);
#pragma unroll
                    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int sk = 4 * (t0 + tc) + rr;
                            if (t0 + tc < MT_L && sk < KS) ylds[sk * 64 + lane] = acc[0][tc][rr];
                        }
                }
            }
            __syncthreads();
            /* ---- y = x + sigma (L xi) (covariance.py:91-92, mrw.py:32-38: sigma is one), every wave its slice, in place ---- */
            {
                bool moved_l = false;
#pragma unroll
                for (int qi = 0; qi < QW; ++qi) {
                    if (qi < nq) {
                        const double x0 = xs[0][2 * qi], x1 = xs[0][2 * qi + 1];
                        const double y0 = x0 + sig_w[8 * qi] * y_w[(2 * qi) * 64];
                        const double y1 = x1 + sig_w[8 * qi + 4] * y_w[(2 * qi + 1) * 64];
                        y_w[(2 * qi) * 64] = y0;
                        y_w[(2 * qi + 1) * 64] = y1;
                        moved_l = moved_l || !(y0 == x0) || !(y1 == x1);
                    }
                }
                int mv = moved_l ? 1 : 0;
                mv |= __shfl_xor(mv, 16, 64);
                mv |= __shfl_xor(mv, 32, 64);
                if (mv && g == 0) movedf[c] = 1;
            }
            __syncthreads();
        }

        /* the accept uniforms BEFORE the evaluation: the deciding wave ends both hand-over chains, and nothing of the draw
         * depends on them (the empty asm keeps the compiler from sinking it back behind the last ticket) */
        double u_dec[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            u_dec[nt] = wave == W - 1 ? ymc_draw_uniform(a.seed, (uint32_t)(a.chain0 + kk[nt]), t) : 0.0;
            asm volatile("" : "+v"(u_dec[nt]));
        }
        /* ---- the evaluation: prior and residual sums handed from wave to wave behind tickets, F = A y on the matrix cores ---- */
        wide_evaluate<NT, W, QW, TC, KB, PD>(cx, a.blob, MT, bias, nprec, dbar, pm_w, pp_w, tkP, tkS);

        /* ---- the decision (metropolisHastings.py:55-73, mrw.py:51-57): the last wave, whose hand-over closed the sums ---- */
        const bool rec = clk.due();
        if (wave == W - 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const double logl = -0.5 * (combine4(handS[nt * 64 + lane]) + a.lik_const);
                const double lpy = a.likelihood_only ? logl : logl + (-0.5 * combine4(handP[nt * 64 + lane]));
                const double u = u_dec[nt];
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
            /* the record row in the TILED layout of the matrix-core kernels, [tile][k-step][lane]: a wave's store of one k-step
             * is 512 contiguous bytes (rows of 8 d bytes a chain would make it sixteen 32-byte pieces: 10 % of a transition at
             * m = 64, profiles/r04_wide_shapes.txt); the host untiles on fetch.  Padding lanes are written like the others. */
            if (rec && a.traj && tile0 + nt < nct) {
                double *rowt = a.traj + (((uint64_t)clk.row * nct + tile0 + nt) * (uint64_t)KS + 2 * q0) * 64 + lane;
#pragma unroll
                for (int qi = 0; qi < QW; ++qi)
                    if (qi < nq) {
                        rowt[(2 * qi) * 64] = xs[nt][2 * qi];
                        rowt[(2 * qi + 1) * 64] = xs[nt][2 * qi + 1];
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
 * Two-level delayed acceptance on the same plan.  One tile of 16 chains per workgroup: the sub-chain's state z is a slice
 * of registers per wave, the fine state x waits in global memory while the sub-chain runs (see x_park below); a sub-step
 * proposes from z into the LDS tile, evaluates the COARSE posterior and moves z; the fine step writes z into the tile,
 * evaluates the FINE posterior, fetches x, and decides with
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

    const uint64_t k = tile * 16 + c;
    const bool live = tile < nct && k < a.K;
    const uint64_t kk = live ? k : 0;
    const uint32_t chain = (uint32_t)(a.chain0 + kk);
    /* The fine state x is PARKED in global memory (a.x_park, this tile's [KS][64] in the tiled layout, this wave's slice):
     * it is needed only after the fine evaluation -- for `proposal == state`, the move and the record -- and as a second
     * slice of registers it was live across every evaluation's matrix loop (170 spilled registers at 18 pairs per wave).
     * zs is the sub-chain's state; between fine steps it IS the fine state. */
    double zs[2 * QW];
    double *park_w = a.x_park + ((uint64_t)tile * KS + 2 * q0) * 64 + lane;
    {
        const double *xr = a.x + kk * d + j0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                zs[2 * qi + e] = (live && qi < nq && j0 + 8 * qi + 4 * e < d) ? xr[8 * qi + 4 * e] : 0.0;
                if (qi < nq) park_w[(2 * qi + e) * 64] = zs[2 * qi + e];
            }
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
            double u = decider ? ymc_draw_uniform(a.seed, chain, t) : 0.0; /* before the evaluation: see linear_wide_fused_kernel */
            asm volatile("" : "+v"(u));
            wide_evaluate<NT, W, QW, TC, KB, PD>(cx, a.blob_c, MT_c, bias_c, nprec_c, dbar_c, pmean_c + j0, pprec_c + j0, tkP, tkS);
            if (decider) {
                const double logl = -0.5 * (combine4(handS[lane]) + a.lik_const_c);
                const double lpcy = logl + (-0.5 * combine4(handP[lane]));
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
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
            if (qi < nq) {
                y_w[(2 * qi) * 64] = zs[2 * qi];
                y_w[(2 * qi + 1) * 64] = zs[2 * qi + 1];
            }
        }
        __syncthreads();
        double u = decider ? ymc_draw_uniform(a.seed, chain, tbase + (uint64_t)S) : 0.0;
        asm volatile("" : "+v"(u));
        wide_evaluate<NT, W, QW, TC, KB, PD>(cx, a.blob_f, MT_f, bias_f, nprec_f, dbar_f, pmean_f + j0, pprec_f + j0, tkP, tkS);
        /* the fine state back from its parking place: did the sub-chain move at all (metropolisHastings.py:60-61)? */
        double xs[2 * QW];
        {
            bool moved_l = false;
#pragma unroll
            for (int i = 0; i < 2 * QW; ++i) xs[i] = i < 2 * nq ? park_w[i * 64] : 0.0;
#pragma unroll
            for (int i = 0; i < 2 * QW; ++i)
                if (i < 2 * nq) moved_l = moved_l || !(zs[i] == xs[i]);
            any_moved(moved_l);
        }
        __syncthreads();
        const bool rec = clk.due();
        if (decider) {
            const double logl = -0.5 * (combine4(handS[lane]) + a.lik_const_f);
            const double lpfz = logl + (-0.5 * combine4(handP[lane]));
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
            for (int i = 0; i < 2 * QW; ++i) {
                zs[i] = acc1 ? zs[i] : xs[i];
                if (acc1 && i < 2 * nq) park_w[i * 64] = zs[i]; /* a rejected chain's parked state stays */
            }
            if (rec && a.traj && tile < nct) { /* tiled, as linear_wide_fused_kernel's */
                double *rowt = a.traj + (((uint64_t)clk.row * nct + tile) * (uint64_t)KS + 2 * q0) * 64 + lane;
#pragma unroll
                for (int qi = 0; qi < QW; ++qi)
                    if (qi < nq) {
                        rowt[(2 * qi) * 64] = zs[2 * qi];
                        rowt[(2 * qi + 1) * 64] = zs[2 * qi + 1];
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
                if (qi < nq && j0 + 8 * qi + 4 * e < d) xr[8 * qi + 4 * e] = zs[2 * qi + e];
        if (decider && g == 0) {
            a.lp[kk] = lpf;
            a.lpc[kk] = lpc;
            a.acc_count[kk] = accepted;
            a.status[kk] = st;
        }
    }
}

/*
 * ---- a tile of 16 chains over a GROUP of workgroups: what the two kernels below share ----
 * Every member of a group holds the tile's state and builds the SAME proposal (the draws are addressed, not consumed: every
 * member computes the same bits); each multiplies its own share of the output tiles and sends the per-output TERMS -- not
 * sums: the canonical order is one sequential chain per lane -- to the group's LEADER (member 0): write-through (sc1)
 * stores, every storing wave drained, ONE agent-scope add per member on the tile's arrival counter.  The leader's last wave
 * polls that ONE word, one agent-scope acquire, a workgroup barrier, plain loads into LDS (cdna_hip_programming.md
 * Guideline 16, the R1 form), adds in the canonical order, decides, and publishes the sixteen accept flags tagged with the
 * transition in ONE aligned 8-byte word, which the other members' last waves poll (the data is the flag: R2).  A member
 * writes transition n + 1's terms only after it has read decision n, which the leader publishes after it has read all of
 * n's terms (the two slots by parity are a margin, not a need).  A tile's polled words have 256 bytes to themselves: all
 * tiles' words on one line cost 6 us per transition at 256 workgroups (every poll and every arrival queue at one channel).
 * Every member of every group must be resident (grid <= CUs, one workgroup per CU by its LDS); a poll that does not end
 * within the limit gives up, flags the chains (YMC_CHAIN_HANDOFF) and fails the run (check_internal_faults): every wave
 * reaches the end of the launch whatever happens.
 */
/* a lane's four terms of output tile t into the slot, lane-major [tile][lane][4]: two 16-byte write-through stores (8-byte
 * write-through stores cost 2.7 x per byte); the data registers must outlive the issue: the s_nop.  The caller drains. */
__device__ __forceinline__ void wide_send_terms(double *slot, int t, int lane, const d4 &terms)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    double *dst = slot + (size_t)t * 256 + 4 * lane;
    const d2 lo = {terms[0], terms[1]}, hi = {terms[2], terms[3]};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\ts_nop 1"
                 :: "v"(dst), "v"(lo), "v"(hi) : "memory");
}

/* the leader's last wave: all G members' arrivals for this transition (bounded), then ONE acquire */
__device__ __forceinline__ void wide_await_arrivals(uint32_t *arrived, uint32_t want, const WideSplitArgs &sa, volatile uint32_t *gave_up, int lane)
{
    if (lane == 0 && !*gave_up) {
        uint32_t spins = 0;
        while ((int32_t)(__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > sa.poll_limit) { /* a member is not running: do not hang the device */
                *gave_up = 1;
                __hip_atomic_fetch_add(sa.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
#if defined(YMC_WS_SC1LOAD)  /* experiment: every load of the slot write-through-aware (sc1), no acquire (Guideline 16, Rule) */
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  /* ONE acquire after the match: this CU's stale lines go */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    /* ... before the barrier lets the others load */
#endif
}

/* the leader, every wave: the slot into LDS a chunk of TCH output tiles at a time (16-byte units (tile, lane, half) -> LDS
 * [tile][4][64], eight units per thread requested together), the last wave adding each chunk in the canonical order --
 * outputs ascending.  Two barriers per chunk; the sum is the last wave's. */
template <int W>
__device__ __forceinline__ double wide_gather_sum(const double *slot, int MT, int TCH, double *terms, int tid, int lane, bool decider)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    double S = 0.0;
#pragma unroll 1
    for (int c0 = 0; c0 < MT; c0 += TCH) {
        const int nterm = (MT - c0 < TCH ? MT - c0 : TCH) * 256;
        const double *src = slot + (size_t)c0 * 256;
        const d2 *src2 = reinterpret_cast<const d2 *>(src);
        const int nunit = nterm / 2;
        __syncthreads();
#pragma unroll 1
        for (int i = tid; i < nunit; i += 8 * 64 * W) {
            d2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = i + u * 64 * W;
#if defined(YMC_WS_SC1LOAD)
                const double *s8 = src + 2 * (size_t)(idx < nunit ? idx : i);
                v[u][0] = __hip_atomic_load(s8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[u][1] = __hip_atomic_load(s8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                v[u] = src2[idx < nunit ? idx : i];
#endif
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = i + u * 64 * W;
                if (idx < nunit) {
                    double *dst = terms + (idx >> 7) * 256 + (idx & 1) * 128 + ((idx & 127) >> 1);
                    dst[0] = v[u][0];
                    dst[64] = v[u][1];
                }
            }
        }
        __syncthreads();
        if (decider) {
            const double *tl = terms + lane;
#pragma unroll 8
            for (int i = 0; i < nterm / 64; ++i) S = S + tl[i * 64];
        }
    }
    return S;
}

/* the leader's last wave: the chains' accept flags of this transition, ONE aligned 8-byte write-through store */
__device__ __forceinline__ void wide_publish_decision(unsigned long long *decision, uint32_t epoch, bool acc1, int lane)
{
    const unsigned long long flags = __ballot(acc1) & 0xffffull; /* lanes 0 .. 15: g = 0, chain c */
    if (lane == 0) __hip_atomic_store(decision, ((unsigned long long)epoch << 32) | flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* the other members' last wave: the leader's word for this transition (bounded); the sixteen flags, in every lane */
__device__ __forceinline__ uint32_t wide_await_decision(unsigned long long *decision, uint32_t epoch, const WideSplitArgs &sa,
                                                        volatile uint32_t *gave_up, int lane)
{
    unsigned long long v = 0;
    if (lane == 0 && !*gave_up) {
        uint32_t spins = 0;
        while (((v = __hip_atomic_load(decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > sa.poll_limit) {
                *gave_up = 1;
                __hip_atomic_fetch_add(sa.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                v = 0;
                break;
            }
        }
    }
    return (uint32_t)__shfl((int)(uint32_t)v, 0, 64);
}

/*
 * Small ensembles: ONE tile of 16 chains over G workgroups (G CUs).  With fewer tiles than CUs the fused kernel leaves most
 * of the chip idle while each busy CU's matrix pipe runs a tile's whole product (4 096 MFMAs at d = m = 512: 31 us).  This
 * build is the fused kernel's evaluation cut along the OUTPUT tiles: the proposal tile in LDS, a wave per output tile of the
 * member's share, the matrix's fragments streamed from L2 (any d the proposal tile fits, any number of output tiles).
 */
template <int W, int QW, bool CG>
__global__ __launch_bounds__(64 * W) void linear_wide_split_kernel(const WideSplitArgs sa)
{
    const WideFusedArgs &a = sa.f;
    ymc_stage_tables();
    extern __shared__ double lds[];
    constexpr int NT = 1, TC = 1, KB = 4;
    constexpr int PD = W >= 16 ? 2 : 4;
    const int KS = a.KS, MT = a.MT, m_pad = 16 * MT, d_pad = 4 * KS, d = a.d, G = sa.G;
    const int KSP = ((KS + PD * KB - 1) / (PD * KB)) * (PD * KB);
    double *ylds = lds;                                   /* [KSP][64], zero beyond KS */
    const int TCH = sa.chunk;                             /* output tiles per staging chunk of the leader's sum */
    double *terms = ylds + (size_t)KSP * 64;              /* [<= TCH][4][64]: the leader's copy of the slot */
    double *dconst = terms + (size_t)(MT < TCH ? MT : TCH) * 256; /* !CG: prior mean | prior prec | sigma */
    const double *consts = a.blob + (size_t)MT * KS * 64; /* bias | n prec | row mean (| the three above) in global memory */
    const double *pmean = CG ? consts + 3 * (size_t)m_pad : dconst;
    const double *pprec = pmean + d_pad, *sigma = pprec + d_pad;
    double *handP = dconst + (CG ? 0 : 3 * d_pad);        /* [64] */
    int *movedf = reinterpret_cast<int *>(handP + 64);    /* [16] */
    int *accf = movedf + 16;                              /* [16] */
    volatile uint32_t *ticket = reinterpret_cast<volatile uint32_t *>(accf + 16); /* [2] */
    uint32_t *arrive = const_cast<uint32_t *>(ticket) + 2; /* storing waves of this member through with their stores, so far */
    volatile uint32_t *gave_up = ticket + 3;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4;
    {
        if (!CG)
            for (int i = tid; i < 3 * d_pad; i += 64 * W) dconst[i] = consts[3 * (size_t)m_pad + i];
        for (int i = tid; i < 16; i += 64 * W) movedf[i] = 0;
        if (tid < 4) ticket[tid] = 0;
        for (int i = tid; i < (KSP - KS) * 64; i += 64 * W) ylds[KS * 64 + i] = 0.0;
    }
    const uint64_t nct = (a.K + 15) / 16;
    const uint64_t tile = blockIdx.x / (unsigned)G;
    const int member = (int)(blockIdx.x % (unsigned)G);
    const bool leader = member == 0;
    const int first_t = (member * MT) / G, n_mine = ((member + 1) * MT) / G - first_t; /* this member's output tiles */
    const int n_store = n_mine < W ? n_mine : W;                                       /* its waves that have some */
    const int Q = KS / 2, q0 = (wave * Q) / W, nq = ((wave + 1) * Q) / W - q0;
    const uint32_t pair_of_g = (g == 0) ? 0u : (g == 1) ? 2u : (g == 2) ? 1u : 3u;
    const int j0 = 8 * q0 + g;
    const double *sig_w = sigma + j0, *pm_w = pmean + j0, *pp_w = pprec + j0;
    double *y_w = ylds + (2 * q0) * 64 + lane;

    const uint64_t k = tile * 16 + c;
    const bool live = tile < nct && k < a.K;
    const uint64_t kk = live ? k : 0;
    const uint32_t chain = (uint32_t)(a.chain0 + kk);
    double xs[2 * QW];
    {
        const double *xr = a.x + kk * d + j0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                xs[2 * qi + e] = (live && qi < nq && j0 + 8 * qi + 4 * e < d) ? xr[8 * qi + 4 * e] : 0.0;
    }
    const bool decider = wave == W - 1;
    double lp = (decider && leader && live) ? a.lp[kk] : 0.0;
    uint64_t accepted = (decider && leader && live) ? a.acc_count[kk] : 0;
    uint32_t st = (decider && leader && live) ? a.status[kk] : 0;
    RecordClock clk(a.rec_stride, a.rec_since);
    __syncthreads();
    const WideEvalCtx cx{ylds, y_w, nullptr, handP, ticket, KS, KSP, nq, lane, wave, g};
    uint32_t tkP = 0;
    uint32_t *arrived = sa.arrived + tile * 64; /* a tile's polled words on lines of their own (256 bytes a tile) */
    unsigned long long *decision = reinterpret_cast<unsigned long long *>(sa.arrived + tile * 64 + 32);

#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t t = a.t0 + n;
        /* ---- the proposal of this wave's slice, into LDS: every member of the group the same ---- */
        {
            bool moved_l = false;
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
                if (qi < nq) {
                    double xi0, xi1;
#if defined(YMC_WS_NODRAW)   /* timing experiments only (tools/exp_wide_split.sh) */
                    xi0 = 1e-3 * (double)(qi + 1), xi1 = -1e-3 * (double)(chain & 7u);
#else
                    ymc_draw_normal_pair(a.seed, chain, t, 4u * (uint32_t)(q0 + qi) + pair_of_g, &xi0, &xi1);
#endif
                    swap_odd_even_rows(xi0, xi1);
                    const double x0 = xs[2 * qi], x1 = xs[2 * qi + 1];
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
                    y_w[(2 * qi) * 64] = y0;
                    y_w[(2 * qi + 1) * 64] = y1;
                    moved_l = moved_l || !(y0 == x0) || !(y1 == x1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            int mv = moved_l ? 1 : 0;
            mv |= __shfl_xor(mv, 16, 64);
            mv |= __shfl_xor(mv, 32, 64);
            if (leader && mv && g == 0) movedf[c] = 1;
        }
        __syncthreads();

        /* the prior's sums first (every member its own: it has the whole proposal): the hand-over runs down the waves
         * while the first ones multiply */
#if !defined(YMC_WS_NOPRIOR)
        if (leader && !a.likelihood_only) wide_prior_chain<NT, W, QW>(cx, pm_w, pp_w, tkP);
#endif

        /* ---- this member's output tiles: products, terms, out to the group ---- */
        const uint32_t epoch = n + 1;
        double *slot = sa.xch + ((tile * 2 + (n & 1u)) * (uint64_t)MT) * 256;
#pragma unroll 1
        for (int ti = wave; ti < n_mine; ti += W) {
            const int t_out = first_t + ti;
            d4 acc[NT][TC];
            acc[0][0] = d4{0.0, 0.0, 0.0, 0.0};
#if !defined(YMC_WS_NOPROD)
            wide_products<NT, TC, KB, PD>(cx, a.blob, MT, t_out, acc);
#endif
            wide_terms<NT, TC>(acc, MT, t_out, g, consts, consts + m_pad, consts + 2 * m_pad);
            wide_send_terms(slot, t_out, lane, acc[0][0]);
        }
        if (wave < n_store) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* EVERY storing wave drains its stores before it counts */
            if (lane == 0) {
                const uint32_t before = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (before + 1 == epoch * (uint32_t)n_store) /* the member's last: ONE add for all its stores */
                    __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        /* ---- the leader: the group's terms and the decision; the others: its word ---- */
        const bool rec = clk.due();
        if (leader) {
            const double u = decider ? ymc_draw_uniform(a.seed, chain, t) : 0.0; /* before the wait, not after it */
            double S = 0.0;
#if !defined(YMC_WS_NOXCH)
            if (decider) wide_await_arrivals(arrived, epoch * (uint32_t)G, sa, gave_up, lane);
            S = wide_gather_sum<W>(slot, MT, TCH, terms, tid, lane, decider);
#endif
            if (decider) {
                const double logl = -0.5 * (combine4(S) + a.lik_const);
                const double lpy = a.likelihood_only ? logl : logl + (-0.5 * combine4(handP[lane]));
                const bool moved = movedf[c] != 0;
                if (*gave_up) st |= YMC_CHAIN_HANDOFF;
                const bool acc1 = ymc_accept(lpy - lp, moved, ymc_finite(lpy), u, st);
                wide_publish_decision(decision, epoch, acc1, lane);
                lp = acc1 ? lpy : lp;
                accepted += acc1 ? 1u : 0u;
                if (g == 0) {
                    accf[c] = acc1 ? 1 : 0;
                    movedf[c] = 0;
                    if (live) {
                        if (a.acc) a.acc[(uint64_t)n * a.K + kk] = acc1 ? 1 : 0;
                        if (rec && a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + kk] = lp;
                    }
                }
            }
        } else if (decider) {
#if defined(YMC_WS_NOXCH)
            const uint32_t fl = 0x5a5au;
#else
            const uint32_t fl = wide_await_decision(decision, epoch, sa, gave_up, lane);
#endif
            if (g == 0) accf[c] = (int)((fl >> c) & 1u);
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
                        xs[2 * qi + e] = acc1 ? y : xs[2 * qi + e];
                    }
                }
            }
            if (rec && a.traj && leader && tile < nct) { /* tiled, as linear_wide_fused_kernel's */
                double *rowt = a.traj + (((uint64_t)clk.row * nct + tile) * (uint64_t)KS + 2 * q0) * 64 + lane;
#pragma unroll
                for (int qi = 0; qi < QW; ++qi)
                    if (qi < nq) {
                        rowt[(2 * qi) * 64] = xs[2 * qi];
                        rowt[(2 * qi + 1) * 64] = xs[2 * qi + 1];
                    }
            }
        }
        if (rec) clk.wrote();
    }

    if (live && leader) {
        double *xr = a.x + kk * d + j0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (qi < nq && j0 + 8 * qi + 4 * e < d) xr[8 * qi + 4 * e] = xs[2 * qi + e];
        if (decider && g == 0) {
            a.lp[kk] = lp;
            a.acc_count[kk] = accepted;
            a.status[kk] = st;
        }
    }
}

/*
 * The same group of workgroups as a SYSTOLIC pipeline (up to 640 parameters, one or two output tiles per member).  What a
 * member multiplies never changes during a launch -- its own rows of the matrix -- and wave w's slice of the proposal
 * (its q-range = the k-steps 2 q0 .. 2 (q0 + nq) - 1) is exactly the MFMA B operand of those k-steps, in the registers
 * of the lanes that made it.  So a wave keeps, for each of the member's output tiles, the matrix fragments of ITS k-steps in
 * registers for the whole launch; an output tile's accumulator starts in wave 0, takes each wave's k-steps in turn (k
 * ascending: the bits of every other kernel) and travels on through LDS behind tickets; the last wave turns it into terms.
 * No proposal tile in LDS, no matrix traffic at all after the first transition, the tiles of a member one stage apart in the
 * pipeline, all four matrix pipes of the CU at work.
 * While a wave waits for the group it draws the NEXT transition's normals: they are addressed by (chain, transition,
 * component), not by the outcome (the last wave, which has the group to attend to, draws its own while the next
 * transition's accumulators come down the pipeline).  The chains' state lives in LDS, every wave its slice: the registers are
 * for the matrix and the two sets of normals.
 */
template <int QW, int NM>
__global__ __launch_bounds__(512) void linear_wide_systolic_kernel(const WideSplitArgs sa)
{
    constexpr int W = 8;
    const WideFusedArgs &a = sa.f;
    ymc_stage_tables();
    extern __shared__ double lds[];
    const int KS = a.KS, MT = a.MT, m_pad = 16 * MT, d_pad = 4 * KS, d = a.d, G = sa.G;
    const int TCH = sa.chunk;                             /* output tiles per staging chunk of the leader's sum */
    const int stage = (W - 1) * NM * 256 > (MT < TCH ? MT : TCH) * 256 ? (W - 1) * NM * 256 : (MT < TCH ? MT : TCH) * 256;
    double *hand = lds;                                   /* [W - 1][NM][4][64]: an accumulator between wave w and w + 1 */
    double *terms = lds;                                  /* [<= TCH][4][64]: the leader's copy of the slot (the pipeline is empty then) */
    double *xlds = lds + stage;                           /* [W][2 QW][64]: the chains' state, every wave its slice (registers
                                                           * are for the matrix and the two sets of normals) */
    double *dconst = xlds + W * 2 * QW * 64;              /* prior mean | prior prec | sigma */
    const double *consts = a.blob + (size_t)MT * KS * 64; /* bias | n prec | row mean | the three above, in global memory */
    const double *pmean = dconst, *pprec = pmean + d_pad, *sigma = pprec + d_pad;
    double *mconst = dconst + 3 * d_pad;                  /* [3][NM][16]: bias | n prec | row mean of the member's outputs */
    double *handP = mconst + 3 * NM * 16;                 /* [64] */
    int *movedf = reinterpret_cast<int *>(handP + 64);    /* [16] */
    int *accf = movedf + 16;                              /* [16] */
    volatile uint32_t *ticket = reinterpret_cast<volatile uint32_t *>(accf + 16); /* [2]: [0] the prior's hand-over */
    volatile uint32_t *gave_up = ticket + 1;
    volatile uint32_t *passed = ticket + 2;               /* [W]: accumulators wave w has passed on so far */

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4;
    {
        for (int i = tid; i < 3 * d_pad; i += 64 * W) dconst[i] = consts[3 * (size_t)m_pad + i];
        for (int i = tid; i < 16; i += 64 * W) movedf[i] = 0;
        if (tid < 2 + W) ticket[tid] = 0;
    }
    const uint64_t nct = (a.K + 15) / 16;
    const uint64_t tile = blockIdx.x / (unsigned)G;
    const int member = (int)(blockIdx.x % (unsigned)G);
    const bool leader = member == 0;
    const int first_t = (member * MT) / G, n_mine = ((member + 1) * MT) / G - first_t; /* <= NM */
    for (int i = tid; i < 3 * NM * 16; i += 64 * W) {
        const int arr = i / (NM * 16), o = i % (NM * 16);
        mconst[i] = o < 16 * n_mine ? consts[(size_t)arr * m_pad + 16 * first_t + o] : 0.0;
    }
    const int Q = KS / 2, q0 = (wave * Q) / W, nq = ((wave + 1) * Q) / W - q0;
    const uint32_t pair_of_g = (g == 0) ? 0u : (g == 1) ? 2u : (g == 2) ? 1u : 3u;
    const int j0 = 8 * q0 + g;

    const uint64_t k = tile * 16 + c;
    const bool live = tile < nct && k < a.K;
    const uint64_t kk = live ? k : 0;
    const uint32_t chain = (uint32_t)(a.chain0 + kk);
    /* this wave's k-steps of the member's rows of the matrix: loaded ONCE */
    double Ar[NM][2 * QW];
#pragma unroll
    for (int ti = 0; ti < NM; ++ti)
#pragma unroll
        for (int j = 0; j < 2 * QW; ++j)
            Ar[ti][j] = (ti < n_mine && j < 2 * nq) ? a.blob[((size_t)(first_t + ti) * KS + 2 * q0 + j) * 64 + lane] : 0.0;
    double ys[2 * QW], xin[2 * QW];
    double *xl = xlds + (size_t)wave * 2 * QW * 64 + lane; /* + (2 qi + e) 64 */
    {
        const double *xr = a.x + kk * d + j0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                xl[(2 * qi + e) * 64] = (live && qi < nq && j0 + 8 * qi + 4 * e < d) ? xr[8 * qi + 4 * e] : 0.0;
    }
    const bool decider = wave == W - 1;
    double lp = (decider && leader && live) ? a.lp[kk] : 0.0;
    uint64_t accepted = (decider && leader && live) ? a.acc_count[kk] : 0;
    uint32_t st = (decider && leader && live) ? a.status[kk] : 0;
    RecordClock clk(a.rec_stride, a.rec_since);
    auto draw = [&](uint64_t t) { /* the normals of transition t for this wave's slice */
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
            if (qi < nq) {
                double xi0, xi1;
#if defined(YMC_WS_NODRAW)   /* timing experiments only (tools/exp_wide_split.sh) */
                xi0 = 1e-3 * (double)(qi + 1), xi1 = -1e-3 * (double)((chain + t) & 7u);
#else
                ymc_draw_normal_pair(a.seed, chain, t, 4u * (uint32_t)(q0 + qi) + pair_of_g, &xi0, &xi1);
#endif
                swap_odd_even_rows(xi0, xi1);
                xin[2 * qi] = xi0, xin[2 * qi + 1] = xi1;
            } else {
                xin[2 * qi] = 0.0, xin[2 * qi + 1] = 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto wait_lds = [&](volatile uint32_t *word, uint32_t want) { /* within the workgroup: never long */
        if (lane == 0)
            while ((int32_t)(*word - want) < 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_wave_barrier();
    };
    __syncthreads();
    draw(a.t0);
    uint32_t tkP = 0;
    uint32_t *arrived = sa.arrived + tile * 64; /* a tile's polled words on lines of their own (256 bytes a tile) */
    unsigned long long *decision = reinterpret_cast<unsigned long long *>(sa.arrived + tile * 64 + 32);

#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t t = a.t0 + n;
        const uint32_t epoch = n + 1;
        /* ---- the proposal of this wave's slice: registers only ---- */
        if (decider && n > 0) draw(t);
        {
            bool moved_l = false;
            const double *sig_w = sigma + j0, *pm_w = pmean + j0;
#pragma unroll
            for (int qi = 0; qi < QW; ++qi) {
                if (qi < nq) {
                    const double x0 = xl[(2 * qi) * 64], x1 = xl[(2 * qi + 1) * 64];
                    double y0, y1;
                    if (a.pcn) {
                        const double p0 = pm_w[8 * qi] + sig_w[8 * qi] * xin[2 * qi];
                        const double p1 = pm_w[8 * qi + 4] + sig_w[8 * qi + 4] * xin[2 * qi + 1];
                        y0 = a.pcn_c1 * x0 + a.pcn_c2 * p0;
                        y1 = a.pcn_c1 * x1 + a.pcn_c2 * p1;
                    } else {
                        y0 = x0 + sig_w[8 * qi] * xin[2 * qi];
                        y1 = x1 + sig_w[8 * qi + 4] * xin[2 * qi + 1];
                    }
                    ys[2 * qi] = y0, ys[2 * qi + 1] = y1;
                    moved_l = moved_l || !(y0 == x0) || !(y1 == x1);
                } else {
                    ys[2 * qi] = 0.0, ys[2 * qi + 1] = 0.0;
                }
            }
            if (leader) {
                int mv = moved_l ? 1 : 0;
                mv |= __shfl_xor(mv, 16, 64);
                mv |= __shfl_xor(mv, 32, 64);
                if (mv && g == 0) movedf[c] = 1;
            }
        }
        /* ---- the leader's prior sums: components ascending, from wave to wave (wide_prior_chain, from registers) ---- */
        if (leader && !a.likelihood_only) {
            const double *pm_w = pmean + j0, *pp_w = pprec + j0;
            double pt[2 * QW];
#pragma unroll
            for (int qi = 0; qi < QW; ++qi)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (qi < nq) {
                        const double dx = ys[2 * qi + e] - pm_w[8 * qi + 4 * e];
                        const double px = pp_w[8 * qi + 4 * e] * dx;
                        pt[2 * qi + e] = dx * px;
                    } else {
                        pt[2 * qi + e] = 0.0;
                    }
                }
            const uint32_t tk = tkP + (uint32_t)wave;
            double P = 0.0;
            if (wave > 0) {
                wide_wait_ticket(ticket, 0, tk, lane);
                P = handP[lane];
            }
#pragma unroll
            for (int qi = 0; qi < QW; ++qi)
                if (qi < nq) {
                    P = P + pt[2 * qi];
                    P = P + pt[2 * qi + 1];
                }
            handP[lane] = P;
            if (lane == 0) ticket[0] = tk + 1;
            tkP += (uint32_t)W;
        }

        /* ---- the pipeline: tile ti's accumulator arrives from wave w - 1, takes this wave's k-steps, moves on ---- */
        double *slot = sa.xch + ((tile * 2 + (n & 1u)) * (uint64_t)MT) * 256;
        const uint32_t base = n * (uint32_t)n_mine;
#pragma unroll
        for (int ti = 0; ti < NM; ++ti) {
            if (ti < n_mine) {
                d4 acc = d4{0.0, 0.0, 0.0, 0.0};
                if (wave > 0) {
                    wait_lds(passed + wave - 1, base + (uint32_t)ti + 1u);
                    const double *h = hand + ((size_t)(wave - 1) * NM + ti) * 256 + lane;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) acc[rr] = h[64 * rr];
                }
#if !defined(YMC_WS_NOPROD)
#pragma unroll
                for (int j = 0; j < 2 * QW; ++j)
                    if (j < 2 * nq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[ti][j], ys[j], acc, 0, 0, 0);
#endif
                if (wave < W - 1) {
                    double *h = hand + ((size_t)wave * NM + ti) * 256 + lane;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) h[64 * rr] = acc[rr];
                    if (lane == 0) passed[wave] = base + (uint32_t)ti + 1u; /* a wave's LDS operations are performed in order */
                } else {
                    d4 tacc[1][1];
                    tacc[0][0] = acc;
                    const double *mc = mconst - 16 * first_t; /* wide_terms indexes by the output's number */
                    wide_terms<1, 1>(tacc, MT, first_t + ti, g, mc, mc + NM * 16, mc + 2 * NM * 16);
                    wide_send_terms(slot, first_t + ti, lane, tacc[0][0]);
                }
            }
        }
        if (decider) { /* the only storing wave of the member: drained, then ONE arrival for all its stores */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        /* ---- meanwhile: the next transition's normals (the last wave has the group to attend to: it draws its own while
         * the next transition's accumulators come down the pipeline) ---- */
        if (!decider && n + 1 < a.nsteps) draw(t + 1);

        const bool rec = clk.due();
        if (leader) {
            /* ---- the leader: all G arrivals, the slot into LDS (a chunk at a time), the sum, the decision ---- */
            const double u = decider ? ymc_draw_uniform(a.seed, chain, t) : 0.0; /* before the wait, not after it */
            double S = 0.0;
#if defined(YMC_WS_NOXCH)
            __syncthreads();
#else
            if (decider) wide_await_arrivals(arrived, epoch * (uint32_t)G, sa, gave_up, lane);
            S = wide_gather_sum<W>(slot, MT, TCH, terms, tid, lane, decider);
#endif
            if (decider) {
                const double logl = -0.5 * (combine4(S) + a.lik_const);
                const double lpy = a.likelihood_only ? logl : logl + (-0.5 * combine4(handP[lane]));
                const bool moved = movedf[c] != 0;
                if (*gave_up) st |= YMC_CHAIN_HANDOFF;
                const bool acc1 = ymc_accept(lpy - lp, moved, ymc_finite(lpy), u, st);
                wide_publish_decision(decision, epoch, acc1, lane);
                lp = acc1 ? lpy : lp;
                accepted += acc1 ? 1u : 0u;
                if (g == 0) {
                    accf[c] = acc1 ? 1 : 0;
                    movedf[c] = 0;
                    if (live) {
                        if (a.acc) a.acc[(uint64_t)n * a.K + kk] = acc1 ? 1 : 0;
                        if (rec && a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + kk] = lp;
                    }
                }
            }
        } else if (decider) {
            /* ---- the others: the leader's word for this transition ---- */
#if defined(YMC_WS_NOXCH)
            const uint32_t fl = 0x5a5au;
#else
            const uint32_t fl = wide_await_decision(decision, epoch, sa, gave_up, lane);
#endif
            if (g == 0) accf[c] = (int)((fl >> c) & 1u);
        }
        __syncthreads();
        /* ---- the move, every wave its slice; the leader writes the record row ---- */
        {
            const bool acc1 = accf[c] != 0;
            if (acc1) {
#pragma unroll
                for (int i = 0; i < 2 * QW; ++i) xl[i * 64] = ys[i];
            }
            if (rec && a.traj && leader && tile < nct) { /* tiled, as linear_wide_fused_kernel's */
                double *rowt = a.traj + (((uint64_t)clk.row * nct + tile) * (uint64_t)KS + 2 * q0) * 64 + lane;
#pragma unroll
                for (int qi = 0; qi < QW; ++qi)
                    if (qi < nq) {
                        rowt[(2 * qi) * 64] = xl[(2 * qi) * 64];
                        rowt[(2 * qi + 1) * 64] = xl[(2 * qi + 1) * 64];
                    }
            }
        }
        if (rec) clk.wrote();
        /* no barrier here: the flags are rewritten only behind the next transition's pipeline and exchange, and the hand-over
         * slots are written in the order they are read */
    }

    if (live && leader) {
        double *xr = a.x + kk * d + j0;
#pragma unroll
        for (int qi = 0; qi < QW; ++qi)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (qi < nq && j0 + 8 * qi + 4 * e < d) xr[8 * qi + 4 * e] = xl[(2 * qi + e) * 64];
        if (decider && g == 0) {
            a.lp[kk] = lp;
            a.acc_count[kk] = accepted;
            a.status[kk] = st;
        }
    }
}

}  // namespace

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
    /* a matrix that no XCD's L2 holds (8 MB and more) is streamed from the memory side by every workgroup: sixteen waves a
     * workgroup (four per SIMD, 128 registers a wave, one tile) keep more requests in flight -- 0.50 -> 0.64 of the fp64 peak
     * at d = 512, m = 2 048; below that size the builds for 256 registers are faster (profiles/r04_wide_shapes.txt) */
    const char *ew = std::getenv("YMC_WIDE_W");
    const bool beyond_l2 = (size_t)KS * (size_t)MT * 512 >= ((size_t)8 << 20);
    if (Q <= 144 && (ew ? std::atoi(ew) == 16 : (beyond_l2 && nct >= 2ull * (uint64_t)cus))) {
        f.W = 16, f.NT = 1, f.QW = 9;
    } else if (nct >= 2ull * (uint64_t)cus) { /* eight waves = two per SIMD, 256 registers each */
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

template <int NT, int W, int QW, int TC, bool CG, bool DENSE = false>
hipError_t launch_fused(const WideFusedArgs &a, size_t lds, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_wide_fused_kernel<NT, W, QW, TC, CG, DENSE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t nct = (a.K + 15) / 16;
    hipLaunchKernelGGL((linear_wide_fused_kernel<NT, W, QW, TC, CG, DENSE>), dim3((unsigned)((nct + NT - 1) / NT)), dim3(64 * W), lds, s, a);
    return hipGetLastError();
}

/* the dense proposal's build: one tile per workgroup of eight waves, two LDS tiles (the normals, the proposal) */
size_t dense_fit(int KS, int MT, int MT_L, FusedPick &f)
{
    const int Q = KS / 2;
    f.NT = 1, f.W = 8, f.QW = 18;
    if ((Q + f.W - 1) / f.W > f.QW) return 0;
    const int most = MT > MT_L ? MT : MT_L;
    f.TC = (most + f.W - 1) / f.W >= 2 ? 2 : 1;
    for (int cg = 0; cg < 2; ++cg) {
        f.CG = cg != 0;
        const int kb = 4 / f.TC, ring = kb * 4;
        const size_t ksp = (size_t)((KS + ring - 1) / ring) * ring;
        const size_t lds = fused_lds(f, KS, MT) + ksp * 64 * sizeof(double);
        if (lds <= kTileLdsLimit) return lds;
    }
    return 0;
}
}  // namespace

size_t ymc_wide_fused_dense_lds_bytes(int KS, int MT, int MT_L)
{
    if (const char *env = std::getenv("YMC_WIDE_FUSED"))
        if (env[0] == '0') return 0;
    FusedPick f;
    return dense_fit(KS, MT, MT_L, f);
}

hipError_t ymc_launch_wide_fused_dense(const WideFusedArgs &a, hipStream_t s)
{
    FusedPick f;
    const size_t lds = dense_fit(a.KS, a.MT, a.MT_L, f);
    if (!lds || !a.chol_frag || a.pcn) return hipErrorInvalidValue;
    if (f.TC == 2) return f.CG ? launch_fused<1, 8, 18, 2, true, true>(a, lds, s) : launch_fused<1, 8, 18, 2, false, true>(a, lds, s);
    return f.CG ? launch_fused<1, 8, 18, 1, true, true>(a, lds, s) : launch_fused<1, 8, 18, 1, false, true>(a, lds, s);
}

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

/* ---- small ensembles: a tile over G workgroups ---- */
namespace {
struct SplitPick {
    int W, QW, G, NM; /* NM > 0: the systolic build with room for NM output tiles per member */
    int chunk;        /* ... and the output tiles its leader stages per round of its sum */
    bool CG;
};

size_t split_lds(const SplitPick &f, int KS, int MT)
{
    const size_t d_pad = 4 * (size_t)KS;
    if (f.NM) {
        const size_t chunk = (size_t)(MT < f.chunk ? MT : f.chunk) * 256, hand = (size_t)7 * f.NM * 256;
        return ((hand > chunk ? hand : chunk) + (size_t)8 * 2 * f.QW * 64 + 3 * d_pad + 3 * (size_t)f.NM * 16 + 64) * sizeof(double) +
               (2 * 16 + 2 + 8) * sizeof(int);
    }
    const int ring = 4 * (f.W >= 16 ? 2 : 4);
    const size_t ksp = (size_t)((KS + ring - 1) / ring) * ring;
    return (ksp * 64 + (size_t)(MT < f.chunk ? MT : f.chunk) * 256 + (f.CG ? 0 : 3 * d_pad) + 64) * sizeof(double) + (2 * 16 + 4) * sizeof(int);
}

size_t split_fit(int KS, int MT, uint64_t K, int cus, SplitPick &f)
{
    const char *env = std::getenv("YMC_WIDE_SPLIT");
    if (env && env[0] == '0') return 0;
    if (const char *ef = std::getenv("YMC_WIDE_FUSED")) /* the launch-per-stage path asked for */
        if (ef[0] == '0') return 0;
    const uint64_t nct = (K + 15) / 16;
    if (nct == 0 || nct * 2 > (uint64_t)cus) return 0; /* every member resident: one workgroup per CU */
    int G = (int)((uint64_t)cus / nct);
    if (G > MT) G = MT;
    if (G > 32) G = 32;
    if (const char *eg = std::getenv("YMC_WIDE_SPLIT_G")) { /* experiments and tests: the group's size */
        const int want = std::atoi(eg);
        if (want >= 1 && want <= G) G = want;
    }
    if (G < 2 && !(env && env[0] == 'f')) return 0;     /* 'f'orce: a group of one runs the protocol alone (tests) */
    f.G = G;
    const int Q = KS / 2;
    /* the systolic build where a wave's k-steps of the member's rows fit its registers; YMC_WIDE_SPLIT_KERNEL=stream: the other */
    const char *ek = std::getenv("YMC_WIDE_SPLIT_KERNEL");
    const int per_member = (MT + G - 1) / G, qw8 = (Q + 7) / 8;
    /* (with four output tiles per member the fragments no longer fit the registers beside the draws: the other build is faster) */
    if (!(ek && ek[0] == 's' && ek[1] == 't') && qw8 <= 10 && per_member <= 2) {
        f.W = 8, f.QW = qw8 <= 8 ? 8 : 10, f.NM = per_member, f.CG = false;
        for (f.chunk = 32; f.chunk >= 8; f.chunk /= 2) {
            const size_t lds = split_lds(f, KS, MT);
            if (lds <= kTileLdsLimit) return lds;
        }
    }
    f.NM = 0;
    f.W = 16, f.QW = 9;
    if (const char *ew = std::getenv("YMC_WIDE_SPLIT_W"))
        if (std::atoi(ew) == 8) f.W = 8, f.QW = 18;
    if ((Q + f.W - 1) / f.W > f.QW) return 0;
    for (f.chunk = 32; f.chunk >= 8; f.chunk /= 2)
        for (int cg = 0; cg < 2; ++cg) {
            f.CG = cg != 0;
            const size_t lds = split_lds(f, KS, MT);
            if (lds <= kTileLdsLimit) return lds;
        }
    return 0;
}

template <int W, int QW, bool CG>
hipError_t launch_split(const WideSplitArgs &a, size_t lds, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_wide_split_kernel<W, QW, CG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t nct = (a.f.K + 15) / 16;
    hipLaunchKernelGGL((linear_wide_split_kernel<W, QW, CG>), dim3((unsigned)(nct * (uint64_t)a.G)), dim3(64 * W), lds, s, a);
    return hipGetLastError();
}

template <int QW, int NM>
hipError_t launch_systolic(const WideSplitArgs &a, size_t lds, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_wide_systolic_kernel<QW, NM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t nct = (a.f.K + 15) / 16;
    hipLaunchKernelGGL((linear_wide_systolic_kernel<QW, NM>), dim3((unsigned)(nct * (uint64_t)a.G)), dim3(512), lds, s, a);
    return hipGetLastError();
}

}  // namespace

int ymc_wide_split_group(int KS, int MT, uint64_t K, int cus)
{
    SplitPick f;
    return split_fit(KS, MT, K, cus, f) ? f.G : 0;
}

size_t ymc_wide_split_exchange_bytes(int MT, uint64_t K) { return ((K + 15) / 16) * 2 * (size_t)MT * 256 * sizeof(double); }

/* the polled words of a launch, one block, 256 bytes a tile (its arrival counter and, 128 bytes on, its decision word: a
 * group's pollers and arrivals meet nobody else's on a line) */
size_t ymc_wide_split_counter_bytes(uint64_t K) { return ((K + 15) / 16) * 256; }

hipError_t ymc_launch_wide_split(WideSplitArgs a, int cus, hipStream_t s)
{
    SplitPick f;
    const size_t lds = split_fit(a.f.KS, a.f.MT, a.f.K, cus, f);
    if (!lds) return hipErrorInvalidValue;
    a.G = f.G;
    a.chunk = f.chunk;
    /* every polled word zeroed before EVERY launch (arrivals and tags count within the launch) */
    hipError_t e = hipMemsetAsync(a.arrived, 0, ymc_wide_split_counter_bytes(a.f.K), s);
    if (e != hipSuccess) return e;
#define YMC_SYSTOLIC(QW_, NM_) \
    if (f.NM == NM_ && f.QW == QW_) return launch_systolic<QW_, NM_>(a, lds, s);
    YMC_SYSTOLIC(8, 1) YMC_SYSTOLIC(8, 2) YMC_SYSTOLIC(10, 1) YMC_SYSTOLIC(10, 2)
#undef YMC_SYSTOLIC
    if (f.W == 16) return f.CG ? launch_split<16, 9, true>(a, lds, s) : launch_split<16, 9, false>(a, lds, s);
    return f.CG ? launch_split<8, 18, true>(a, lds, s) : launch_split<8, 18, false>(a, lds, s);
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
    /* sixteen waves (four per SIMD, 128 registers) beat eight at every shape measured but the smallest forward maps
     * (profiles/r04_wide_shapes.txt: d = 640, outputs 64 / 256: 11.3 -> 8.5 ms per 4 fine steps; d = 320, outputs 32 / 64: 4.0 -> 4.7):
     * an evaluation is a chain of hand-overs and barriers, and more waves hide more of it */
    bool eight = MT_f <= 4 && nct >= (uint64_t)cus;
    if (const char *env = std::getenv("YMC_WIDE2_W")) eight = std::atoi(env) != 16; /* experiments and tests: the workgroup's size */
    if ((Q + 15) / 16 > 9) eight = true;
    if (eight) f.W = 8, f.QW = (Q + 7) / 8 <= 9 ? 9 : 18;
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
