/*
 * ymc_tile.hip -- the linear-forward-model posterior at large dimension
 * (BASELINE configs[2]: d = data_dim = 128, 262 144 chains): random-walk
 * Metropolis with the forward map G(theta) = A theta + b
 * (exampleSetup.py:42-47) evaluated on the fp64 matrix cores.
 *
 * Mapping (one wave = 16 chains, one workgroup = 8 waves = 128 chains, two waves per SIMD):
 *   lane l = 16 g + c owns chain c of the wave's tile and the parameter
 *   components j = 4 s + g, s = 0..KS-1 -- exactly the B-operand layout of
 *   v_mfma_f64_16x16x4_f64 (B[k = l>>4][col = l&15]).  The state x and the
 *   proposal y therefore never leave registers; the forward map is
 *       D[output][chain] = sum_k A[output][k] * y[chain][k]
 *   with A's 16x4 fragments read from LDS (the whole matrix, 128 KB at
 *   d = m = 128, is staged once per workgroup) and one accumulator tile
 *   (16 outputs x 16 chains) per MFMA chain.  On gfx950 the f64 MFMA is
 *   bit-for-bit a k-ordered fma chain (tools/probe_mfma_f64.hip), so F equals
 *   the CPU checker's  acc = fma(A[m][j], theta[j], acc)  loop exactly.
 *   The accumulator layout puts output m = 16 tile + 4 r + g of chain c in
 *   lane (g, c): each lane sums the likelihood terms of the outputs
 *   m = g (mod 4) in increasing m -- the ensemble's canonical "four
 *   interleaved partial sums" -- and two cross-lane adds finish
 *   (S0 + S1) + (S2 + S3).  The prior is summed the same way over j = g (mod 4).
 *   Normals: lane (g, c) runs Box-Muller on pair 4 q + {0,2,1,3}[g], keeps
 *   the output that belongs to its own component set and swaps the other with
 *   lane l ^ 16 (one 64-bit exchange per pair).
 * Record layout (coalesced 512-B stores): traj[row][chain tile][s][lane],
 * i.e. logical (row, chain = 16 tile + c, component = 4 s + g); untiled by
 * untile_kernel when the host fetches it.
 */
/* x and y take 128 of the 256 VGPRs a wave may use at two waves per SIMD: keep the
 * polynomial coefficients of the draw transform in SGPRs */
#define YMC_COEFF_IN_SGPR 1
#include "ymc_kernels.h"
#include "ymc_tile.h"
#include "ymc_tile_dev.h"

namespace {

/* A wave raises its priority while it draws and proposes (vector instructions) and drops it for its forward map:
 * next to a SIMD partner that is issuing MFMAs the drawing wave otherwise gets the issue slots the older wave leaves
 * it.  +1.5...2 % on C3 and on its two-level workload; s_sleep or a scalar-memory wait at the phase boundaries and
 * the opposite priorities change nothing (profiles/r03_tile_experiments.txt). */
#define YMC_PHASE_DRAWS() __builtin_amdgcn_s_setprio(1)
#define YMC_PHASE_MAP() __builtin_amdgcn_s_setprio(0)

/* PROP: the proposal of the kernel -- diagonal random walk, pCN, or random walk with a dense covariance.  A
 * compile-time choice: the dense branch as a run-time test cost the diagonal kernel 20 VGPRs (8 spilled at
 * KS = 32) and 20 % of its speed. */
enum { kPropDiag = 0, kPropPcn = 1, kPropDense = 2, kPropDenseStaged = 3 };

/* kPropDenseStaged: the dense proposal's factor L (read by every transition) through the workgroup's shared LDS
 * double buffer instead of every wave's own L2 reads, as the fine matrix of linear_tile_mlda2_kernel; waves past
 * the last tile then stay in the loop for the barriers and touch no memory. */
/* ASTREAM: the forward matrix does not fit the LDS beside its constants (m d > ~19 000 entries, or d > 128): it stays
 * in global memory (L2-resident up to a few MB) and streams through the workgroup's shared double buffer like the
 * upper matrices of the multi-level kernels -- any number of outputs whose constants fit the LDS (m <~ 5 000), and,
 * with WAVES = 4 (one wave per SIMD: x and y are 256 registers at KS = 64), parameter dimensions up to 256. */
template <int KS, int PROP, int WAVES = kTileWaves, bool ASTREAM = false>
__global__ __launch_bounds__(64 * WAVES) void linear_tile_kernel(const TileArgs a)
{
    constexpr bool PCN = PROP == kPropPcn, DENSE = PROP == kPropDense || PROP == kPropDenseStaged;
    constexpr bool STAGED = PROP == kPropDenseStaged || ASTREAM;
    constexpr int kThreads = 64 * WAVES;
    static_assert(!(ASTREAM && PROP == kPropDense), "a streamed forward matrix takes the proposal's factor through the same buffer");
    ymc_stage_tables();
    extern __shared__ double lds[];
    const int MT = a.MT, m_pad = 16 * MT, d_pad = 4 * KS;
    const size_t frag_doubles = (size_t)MT * KS * 64, const_doubles = (size_t)m_pad * 3 + 3 * (size_t)d_pad;
    const double *Alds = ASTREAM ? a.blob : lds;            /* ASTREAM: the fragments stay in global memory */
    double *bias = lds + (ASTREAM ? 0 : frag_doubles);
    double *nprec = bias + m_pad;
    double *dbar = nprec + m_pad;
    double *pmean = dbar + m_pad;
    double *pprec = pmean + d_pad;
    double *sigma = pprec + d_pad;
    const size_t blob_doubles = (ASTREAM ? 0 : frag_doubles) + const_doubles;

    const int tid = threadIdx.x;
    if constexpr (ASTREAM) {
        for (size_t i = tid; i < const_doubles; i += kThreads) bias[i] = a.blob[frag_doubles + i];
    } else {
        ymc_stage_block<kThreads>(lds, a.blob, blob_doubles / 2, tid);
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const uint64_t nct = (a.K + 15) / 16;
    const uint64_t tile = (uint64_t)blockIdx.x * WAVES + wave;
    const bool wave_live = tile < nct;   /* STAGED: a wave past the last tile stays for the barriers and touches no memory */
    if (!STAGED && !wave_live) return;
    const uint64_t k = tile * 16 + c;
    const bool live = wave_live && k < a.K;
    const uint32_t chain = (uint32_t)(a.chain0 + k);
    const int d = a.d;
    const uint32_t pair_of_g = (g == 0) ? 0u : (g == 1) ? 2u : (g == 2) ? 1u : 3u;
    double *stage = lds + blob_doubles;                          /* STAGED: the stream's double buffer (ymc_tile_dev.h) */

    double x[KS], y[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int j = 4 * s + g;
        x[s] = (live && j < d) ? a.x[k * d + j] : 0.0;
    }
    double lp = live ? a.lp[k] : 0.0;
    uint64_t accepted = live ? a.acc_count[k] : 0;
    uint32_t st = live ? a.status[k] : 0;
    UniformCache uc;
    uc.valid = false;
    uc.pair = 0;
    RecordClock clk(a.rec_stride, a.rec_since);

#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t t = a.t0 + n;
        YMC_PHASE_DRAWS();
        /* proposal y = x + sigma * xi (mrw.py:32-38) in the MFMA operand layout */
        bool moved_l = false;
        /* one pair of this lane: even rows (g = 0, 2) drew the pair of components (8 q + g, 8 q + g + 1), odd rows that of
         * (8 q + 3 + g, 8 q + 4 + g): an even row keeps its first normal and takes its partner's first, an odd row takes
         * its partner's second and keeps its own second */
        auto place = [&](int q, double xi0, double xi1) { /* components 8 q + g (s = 2 q) and 8 q + 4 + g (s = 2 q + 1) */
            swap_odd_even_rows(xi0, xi1);
            if constexpr (DENSE) { /* dense proposal covariance: park xi in y, L xi follows below */
                y[2 * q] = xi0;
                y[2 * q + 1] = xi1;
            } else if constexpr (PCN) { /* pcn.py:27-36: xi ~ prior, y = sqrt(1 - 2h) x + sqrt(2h) xi */
                const double p0 = pmean[8 * q + g] + sigma[8 * q + g] * xi0;
                const double p1 = pmean[8 * q + 4 + g] + sigma[8 * q + 4 + g] * xi1;
                y[2 * q] = a.pcn_c1 * x[2 * q] + a.pcn_c2 * p0;
                y[2 * q + 1] = a.pcn_c1 * x[2 * q + 1] + a.pcn_c2 * p1;
            } else {
                y[2 * q] = x[2 * q] + sigma[8 * q + g] * xi0;
                y[2 * q + 1] = x[2 * q + 1] + sigma[8 * q + 4 + g] * xi1;
            }
            if constexpr (!DENSE) moved_l = moved_l || !(y[2 * q] == x[2 * q]) || !(y[2 * q + 1] == x[2 * q + 1]);
        };
#if defined(YMC_DRAW_SPEC_V2X)
        /* experiment: this lane's pairs 3 i .. 3 i + 2 from two blocks, what is left over from blocks of their own */
        constexpr int kTriples = (KS / 2) / 3;
#pragma unroll
        for (int i = 0; i < kTriples; ++i) {
            double za, zb, zc, zd, ze, zf;
            ymc_v2x_normal_triple(a.seed, chain, t, 64u * (uint32_t)i + pair_of_g, &za, &zb, &zc, &zd, &ze, &zf);
            place(3 * i, za, zb);
            place(3 * i + 1, zc, zd);
            place(3 * i + 2, ze, zf);
        }
#pragma unroll
        for (int q = 3 * kTriples; q < KS / 2; ++q) {
            double xi0, xi1;
            ymc_draw_normal_pair(a.seed, chain, t, 4u * (uint32_t)q + pair_of_g, &xi0, &xi1);
            place(q, xi0, xi1);
        }
#else
#pragma unroll
        for (int q = 0; q < KS / 2; ++q) {
            double xi0, xi1;
            ymc_draw_normal_pair(a.seed, chain, t, 4u * (uint32_t)q + pair_of_g, &xi0, &xi1);
            place(q, xi0, xi1);
        }
#endif
        if constexpr (DENSE) {
            /* y = x + L xi (covariance.py:91-92, gaussian.py:66): row i of L xi is one k-ordered fma chain on the
             * matrix cores -- output 16 mt + 4 r + g of the accumulator IS component 4 s + g, s = 4 mt + r, of this
             * lane.  The rows pass through a scratch in the record's layout, because y holds xi until the last
             * product (L's fragments come from global memory: 128 KB, L2-resident, as the fine matrix of the
             * two-level kernel). */
            double *wt = a.w_tiled + ((wave_live ? tile : 0) * KS) * 64 + lane;
            if constexpr (STAGED) {
                ymc_stream_tile_pairs<KS, 4, kThreads>(a.chol_frag, a.MT_L / 2, stage, tid, lane, y,
                                                           [&](int pair, const d4 &acc0, const d4 &acc1) {
                    if (wave_live) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (8 * pair + r < KS) wt[(8 * pair + r) * 64] = acc0[r];
                            if (8 * pair + 4 + r < KS) wt[(8 * pair + 4 + r) * 64] = acc1[r];
                        }
                    }
                });
            } else {
#pragma unroll 1
                for (int mt = 0; mt < a.MT_L; mt += 2) {
                    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                    const double *L0 = a.chol_frag + (size_t)mt * KS * 64 + lane;
                    const double *L1 = L0 + (size_t)KS * 64;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(L0[s * 64], y[s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(L1[s * 64], y[s], acc1, 0, 0, 0);
                    }
                    if (wave_live) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (4 * mt + r < KS) wt[(4 * mt + r) * 64] = acc0[r];
                            if (4 * (mt + 1) + r < KS) wt[(4 * (mt + 1) + r) * 64] = acc1[r];
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                y[s] = x[s] + wt[s * 64];
                moved_l = moved_l || !(y[s] == x[s]);
            }
        }
        int mv = moved_l ? 1 : 0;
        mv |= __shfl_xor(mv, 16, 64);
        mv |= __shfl_xor(mv, 32, 64);
        const bool moved = mv != 0;

        /* forward map on the matrix cores + Gaussian-noise likelihood (likelihood.py:74-84) through
         * the row-mean identity: |F - Dbar|^2_{n P} + const (ymc_linear_data_summary) */
        YMC_PHASE_MAP();
        double Sres;
        if constexpr (ASTREAM) {
            Sres = 0.0;
            ymc_stream_tile_pairs<KS, 4, kThreads>(Alds, MT / 2, stage, tid, lane, y, [&](int pair, const d4 &acc0, const d4 &acc1) {
                ymc_tile_add_residuals(2 * pair, acc0, acc1, bias, nprec, dbar, g, Sres);
            });
        } else {
            Sres = ymc_tile_residual_sum<KS>(Alds, MT, bias, nprec, dbar, lane, g, y);
        }
        const double logl = -0.5 * (combine4(Sres) + a.lik_const);

        /* Gaussian prior (gaussian.py:19-24); the target of pCN is the likelihood alone (pcn.py:51-57) */
        double lpy = logl;
        if constexpr (!PCN) lpy = logl + ymc_tile_logprior<KS>(pmean, pprec, g, y);

        /* accept / reject (metropolisHastings.py:55-73, mrw.py:51-57) */
        const double u = ymc_uniform_cached(uc, a.seed, chain, t);
        const bool acc = ymc_accept(lpy - lp, moved, ymc_finite(lpy), u, st);
        if (acc) { /* under the accepting lanes' EXEC mask: KS 64-bit moves instead of 2 KS selects */
#pragma unroll
            for (int s = 0; s < KS; ++s) x[s] = y[s];
        }
        lp = acc ? lpy : lp;
        accepted += acc ? 1u : 0u;

        const bool rec = clk.due();
        if (rec && a.traj && wave_live) {
            double *row = a.traj + (((uint64_t)clk.row * nct + tile) * KS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) row[s * 64] = x[s];
        }
        if (live && g == 0) {
            if (a.acc) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
            if (rec && a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + k] = lp;
        }
        if (rec) clk.wrote();
    }

    if (live) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int j = 4 * s + g;
            if (j < d) a.x[k * d + j] = x[s];
        }
        if (g == 0) {
            a.lp[k] = lp;
            a.acc_count[k] = accepted;
            a.status[k] = st;
        }
    }
}

/*
 * Two-level delayed acceptance over two linear models at large dimension (MLDABuilder.bayesModel = a
 * BayesianRegressionModelHierarchy, mlda.py:315-335; example_inference_linearModel_twoLevel.py at C3's size).
 * Same mapping as linear_tile_kernel.  The COARSE model is evaluated S times per fine step, so its matrix is
 * the one staged in LDS; the FINE model's fragments come from global memory (128 KB, resident in every XCD's
 * L2) -- two matrices do not fit the 160 KB of LDS.  Registers hold the fine state x, the sub-chain state z and
 * the proposal y: nothing of a chain's state goes through memory between fine steps (up to round 2 the fine
 * state was parked in a tiled scratch and read back once per fine step: 2 KB of HBM traffic per chain and fine
 * step beside the 1 KB of the record, and at KS = 32 the 256 registers of two waves per SIMD did not hold z, y
 * and the working set of the draws -- 32 VGPRs spilled into the loops).  WAVES = waves per workgroup: 4 at
 * KS = 32 (one wave per SIMD, up to 512 registers -- the LDS admits one workgroup per CU anyway), 8 below.
 * Draw addressing as mlda2_kernel: fine step n owns transitions t = n (S + 1) + s.
 *
 * STAGED: the waves of a workgroup take their fine steps in lockstep (same S for all), so the fine
 * matrix is brought from L2 ONCE per workgroup and fine step instead of once per wave: it streams through a
 * double buffer in the last 16 KB of LDS (units of a quarter of the k-steps of a pair of output tiles;
 * every thread fetches its share of the next unit into registers while the MFMAs of the current one run; one
 * barrier per unit).  A wave reading its own fragments from L2 keeps few loads in flight beside x and y --
 * measured 5.7 x the cost of an LDS-fed GEMV.  Waves past the last tile stay in the loop for the barriers and
 * touch no memory.  The plain form remains for shapes whose constants leave no room for the buffer.
 */
/* where the fine state x lives between fine steps; kXLds and kXMemC take the COARSE matrix from L2 too (through the
 * shared double buffer, as the fine one) */
enum { kXRegs = 0, kXLds = 1, kXMem = 2, kXMemC = 3 };

template <int KS, bool STAGED, int WAVES, int XM>
__global__ __launch_bounds__(64 * WAVES) void linear_tile_mlda2_kernel(const TileArgs a)
{
    constexpr bool XLDS = XM == kXLds, XMEM = XM == kXMem || XM == kXMemC, CSTREAM = XM == kXLds || XM == kXMemC;
    static_assert(!CSTREAM || STAGED, "a streamed coarse matrix shares the fine one's double buffer");
    constexpr int kThreads = 64 * WAVES;
    ymc_stage_tables();
    extern __shared__ double lds[];
    const int MTc = a.MT, MTf = a.MT_f, m_pad_c = 16 * MTc, m_pad_f = 16 * MTf, d_pad = 4 * KS;
    /* coarse level: [A fragments, unless CSTREAM] | bias | n * noise_prec | row mean | prior mean | prior prec | sigma */
    const size_t frag_c = (size_t)MTc * KS * 64, consts_c = (size_t)m_pad_c * 3 + 3 * (size_t)d_pad;
    const double *Ac = CSTREAM ? a.blob : lds;
    double *bias_c = lds + (CSTREAM ? 0 : frag_c);
    double *nprec_c = bias_c + m_pad_c;
    double *dbar_c = nprec_c + m_pad_c;
    double *pmean_c = dbar_c + m_pad_c;
    double *pprec_c = pmean_c + d_pad;
    double *sigma = pprec_c + d_pad;
    /* fine level: the constants only */
    double *bias_f = bias_c + consts_c;
    double *nprec_f = bias_f + m_pad_f;
    double *dbar_f = nprec_f + m_pad_f;
    double *pmean_f = dbar_f + m_pad_f;
    double *pprec_f = pmean_f + d_pad;
    const size_t consts_f = (size_t)m_pad_f * 3 + 2 * (size_t)d_pad;
    const double *Af = a.blob_f;
    double *stage = bias_f + consts_f;                           /* STAGED: the stream's double buffer (ymc_tile_dev.h) */

    const int tid = threadIdx.x;
    if constexpr (CSTREAM) {
        for (size_t i = tid; i < consts_c; i += kThreads) bias_c[i] = a.blob[frag_c + i];
    } else {
        ymc_stage_block<kThreads>(lds, a.blob, (frag_c + consts_c) / 2, tid);
    }
    for (size_t i = tid; i < consts_f; i += kThreads) bias_f[i] = a.blob_f[(size_t)MTf * KS * 64 + i];
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const uint64_t nct = (a.K + 15) / 16;
    const uint64_t tile = (uint64_t)blockIdx.x * WAVES + wave;
    const bool wave_live = tile < nct; /* a wave past the last tile: STAGED keeps it for the barriers, it touches no memory */
    if (!STAGED && !wave_live) return;
    const uint64_t k = tile * 16 + c;
    const bool live = wave_live && k < a.K;
    const uint32_t chain = (uint32_t)(a.chain0 + k);
    const int d = a.d, S = a.S;
    const uint32_t pair_of_g = (g == 0) ? 0u : (g == 1) ? 2u : (g == 2) ? 1u : 3u;

    /* the fine state x: registers; (XLDS) this lane's column of the wave's block in LDS; or (XMEM) memory -- the
     * record row written last (it IS the fine state) when every fine step is recorded, else a scratch in the
     * record's layout --, read into y's registers once per fine step (y is dead between sub-chain and decision) */
    constexpr int KX = XM == kXRegs ? KS : 1;
    double x[KX], z[KS], y[KS];
    double *xs = stage + 2 * ymc_stream_unit<KS, 4>() + (size_t)wave * KS * 64 + lane;
    double *xt = a.x_tiled + ((wave_live ? tile : 0) * KS) * 64 + lane;
    const double *xsrc = xt;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int j = 4 * s + g;
        z[s] = (live && j < d) ? a.x[k * d + j] : 0.0;
        if constexpr (XLDS) xs[s * 64] = z[s];
        else if constexpr (XMEM) {
            if (wave_live) xt[s * 64] = z[s];
        } else x[s] = z[s];
    }
    double lpf = live ? a.lp[k] : 0.0;
    double lpc = live ? a.lpc[k] : 0.0;
    uint64_t accepted = live ? a.acc_count[k] : 0;
    uint32_t st = live ? a.status[k] : 0;
    UniformCache uc;
    uc.valid = false;
    uc.pair = 0;
    RecordClock clk(a.rec_stride, a.rec_since);

    /* Gaussian-noise likelihood of the forward value A v + b through the row-mean identity, the forward map
     * on the matrix cores (as linear_tile_kernel); A's fragments come from LDS or from global memory */
    auto loglik = [&](const double *A, int MT, const double *bias, const double *nprec, const double *dbar,
                      double lik_const, const double (&v)[KS]) {
        return -0.5 * (combine4(ymc_tile_residual_sum<KS>(A, MT, bias, nprec, dbar, lane, g, v)) + lik_const);
    };
    auto logprior = [&](const double *pmean, const double *pprec, const double (&v)[KS]) {
        return ymc_tile_logprior<KS>(pmean, pprec, g, v);
    };
    auto any_differs = [&](const double (&p)[KS], const double (&q)[KS]) { return ymc_tile_any_differs<KS>(p, q); };
    /* STAGED: a level's log-likelihood over the workgroup's shared stream of its matrix
     * (ymc_stream_tile_pairs, ymc_tile_dev.h); every wave calls it at the same point of the loop */
    auto loglik_staged = [&](const double *A, int MT, const double *bias, const double *nprec, const double *dbar,
                             double lik_const, const double (&v)[KS]) {
        double Ssum = 0.0;
        ymc_stream_tile_pairs<KS, 4, kThreads>(A, MT / 2, stage, tid, lane, v, [&](int pair, const d4 &acc0, const d4 &acc1) {
            ymc_tile_add_residuals(2 * pair, acc0, acc1, bias, nprec, dbar, g, Ssum);
        });
        return -0.5 * (combine4(Ssum) + lik_const);
    };

    YMC_PROF_DECL;
#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t tbase = (a.t0 + n) * (uint64_t)(S + 1);
        double lpcz = lpc;
        YMC_PROF_START();
        /* surrogate.run(S + 1, state): S random-walk transitions on the coarse posterior, from z == x */
#pragma unroll 1
        for (int sub = 0; sub < S; ++sub) {
            const uint64_t t = tbase + (uint64_t)sub;
            YMC_PHASE_DRAWS();
#pragma unroll
            for (int q = 0; q < KS / 2; ++q) {
                double xi0, xi1;
                ymc_draw_normal_pair(a.seed, chain, t, 4u * (uint32_t)q + pair_of_g, &xi0, &xi1);
                swap_odd_even_rows(xi0, xi1);
                y[2 * q] = z[2 * q] + sigma[8 * q + g] * xi0;
                y[2 * q + 1] = z[2 * q + 1] + sigma[8 * q + 4 + g] * xi1;
            }
            const bool moved = any_differs(y, z);
            YMC_PROF(0);
            YMC_PHASE_MAP();
            double llc;
            if constexpr (CSTREAM) llc = loglik_staged(Ac, MTc, bias_c, nprec_c, dbar_c, a.lik_const, y);
            else llc = loglik(Ac, MTc, bias_c, nprec_c, dbar_c, a.lik_const, y);
                YMC_PROF(1);
            const double lpcy = llc + logprior(pmean_c, pprec_c, y);
            const double u = ymc_uniform_cached(uc, a.seed, chain, t);
            const bool acc = ymc_accept(lpcy - lpcz, moved, ymc_finite(lpcy), u, st);
            if (acc) { /* 64-bit moves under the accepting lanes' EXEC mask instead of 2 KS selects */
#pragma unroll
                for (int s = 0; s < KS; ++s) z[s] = y[s];
            }
            lpcz = acc ? lpcy : lpcz;
            if (a.sub_acc && live && g == 0) a.sub_acc[((uint64_t)n * S + sub) * a.K + k] = acc ? 1 : 0;
            YMC_PROF(2);
        }
        /* fine accept / reject (mlda.py:146-154) of the proposal z against the fine state x */
        if constexpr (XMEM) {
#pragma unroll
            for (int s = 0; s < KS; ++s) y[s] = wave_live ? xsrc[s * 64] : 0.0;     /* in flight while the fine map runs */
        }
        bool moved = false;
        if constexpr (XMEM) {
        } else if constexpr (XLDS) {
            bool dl = false;
#pragma unroll
            for (int s = 0; s < KS; ++s) dl = dl || !(z[s] == xs[s * 64]);
            int mv = dl ? 1 : 0;
            mv |= __shfl_xor(mv, 16, 64);
            mv |= __shfl_xor(mv, 32, 64);
            moved = mv != 0;
        } else {
            moved = ymc_tile_any_differs<KS>(z, reinterpret_cast<const double (&)[KS]>(x));
        }
        YMC_PROF(3);
        double llf;
        if constexpr (STAGED) llf = loglik_staged(Af, MTf, bias_f, nprec_f, dbar_f, a.lik_const_f, z);
        else llf = loglik(Af, MTf, bias_f, nprec_f, dbar_f, a.lik_const_f, z);
        YMC_PROF(4);
        if constexpr (XMEM) moved = any_differs(z, y);
        const double lpfy = llf + logprior(pmean_f, pprec_f, z);
        const double log_ratio = ((lpfy + lpc) - lpcz) - lpf;
        const double u = ymc_uniform_cached(uc, a.seed, chain, tbase + (uint64_t)S);
        const bool acc = ymc_accept(log_ratio, moved, ymc_finite(lpfy) && ymc_finite(lpcz), u, st);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if constexpr (XLDS) {
                z[s] = acc ? z[s] : xs[s * 64];
                xs[s * 64] = z[s];
            } else if constexpr (XMEM) {
                z[s] = acc ? z[s] : y[s];
            } else {
                x[s] = acc ? z[s] : x[s];
                z[s] = x[s];
            }
        }
        lpf = acc ? lpfy : lpf;
        lpc = acc ? lpcz : lpc;
        accepted += acc ? 1u : 0u;
        YMC_PROF(5);

        const bool rec = clk.due();
        if (rec && a.traj && wave_live) {
            double *row = a.traj + (((uint64_t)clk.row * nct + tile) * KS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) row[s * 64] = z[s];       /* z == x here */
            if constexpr (XMEM) xsrc = row;
        } else if constexpr (XMEM) {
            if (wave_live) {
#pragma unroll
                for (int s = 0; s < KS; ++s) xt[s * 64] = z[s];
                xsrc = xt;
            }
        }
        if (live && g == 0) {
            if (a.acc) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
            if (rec && a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + k] = lpf;
        }
        if (rec) clk.wrote();
        YMC_PROF(6);
    }
    YMC_PROF_FLUSH();

    if (live) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int j = 4 * s + g;
            if (j < d) a.x[k * d + j] = z[s];
        }
        if (g == 0) {
            a.lp[k] = lpf;
            a.lpc[k] = lpc;
            a.acc_count[k] = accepted;
            a.status[k] = st;
        }
    }
}

/* record in tiled layout -> out[rows][K][d] */
__global__ __launch_bounds__(256) void untile_kernel(const double *__restrict__ tiled, uint64_t row0, uint64_t rows,
                                                     uint64_t K, int d, int KS, double *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * K * (uint64_t)d) return;
    const uint64_t j = i % d, k = (i / d) % K, r = i / ((uint64_t)d * K);
    const uint64_t nct = (K + 15) / 16;
    out[i] = tiled[((((row0 + r) * nct + k / 16) * KS + j / 4) * 64) + (j % 4) * 16 + (k % 16)];
}

/* FullDiagnostics' Welford accumulator (diagnostics.py:91-94, estimation.py:36-53)
 * for the tile kernel, which has no registers left for 2 x 128 running moments:
 * replayed from the record after each launch.  Thread i owns one (chain,
 * component) in the record's tiled order, so every row is read as coalesced
 * 512-B segments; the update sequence per component is the online one, hence
 * the same bits.  The pre-transition state of the launch's first transition is
 * the carried state saved before the launch (`x_before`). */
__global__ __launch_bounds__(256) void welford_replay_kernel(const double *__restrict__ tiled,
                                                             const double *__restrict__ x_before, uint64_t rows,
                                                             uint64_t K, int d, int KS, uint64_t wf_n0,
                                                             double *__restrict__ wf_mean, double *__restrict__ wf_m2)
{
    const uint64_t nct = (K + 15) / 16;
    const uint64_t per_row = nct * (uint64_t)KS * 64;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_row) return;
    const uint64_t lane = i & 63, s = (i >> 6) % KS, ct = (i >> 6) / KS;
    const uint64_t k = ct * 16 + (lane & 15);
    const int j = (int)(4 * s + (lane >> 4));
    if (k >= K || j >= d) return;
    double mean = wf_mean[k * d + j], m2 = wf_m2[k * d + j];
    double v = x_before[k * d + j];
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
        for (int q = 0; q < 8; ++q) nx[q] = tiled[(n + q) * per_row + i];
#pragma unroll
        for (int q = 0; q < 8; ++q) step(n + q, nx[q]);
    }
    for (; n < rows; ++n) step(n, tiled[n * per_row + i]);
    wf_mean[k * d + j] = mean;
    wf_m2[k * d + j] = m2;
}

/* DensityInterface.evaluate_log / ForwardModel.evaluate for a batch at large
 * dimension: one lane per parameter vector, plain loops in the canonical
 * order (used for the initial states and for host-side evaluate_log calls) */
__global__ __launch_bounds__(256) void linear_big_eval_kernel(const BigLinearArgs p, uint64_t n,
                                                              const double *__restrict__ xin,
                                                              double *__restrict__ lp_out,
                                                              double *__restrict__ fwd_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *x = xin + i * p.d;
    double S[4] = {0.0, 0.0, 0.0, 0.0};
    for (int m4 = 0; m4 < p.m; m4 += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m4 + q;
            if (m < p.m) {
                double acc = 0.0;
                for (int j = 0; j < p.d; ++j) acc = ymc_fma(p.A[(size_t)m * p.d + j], x[j], acc);
                const double F = acc + p.b[m];
                if (fwd_out) fwd_out[i * p.m + m] = F;
                const double res = F - p.dbar[m];
                const double px = p.nprec[m] * res;
                S[q] = S[q] + res * px;
            }
        }
    }
    if (!lp_out) return;
    const double total = ((S[0] + S[1]) + (S[2] + S[3])) + p.lik_const;
    double P[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j4 = 0; j4 < p.d; j4 += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j4 + q;
            if (j < p.d) {
                const double dx = x[j] - p.prior_mean[j];
                const double px = p.prior_prec[j] * dx;
                P[q] = P[q] + dx * px;
            }
        }
    }
    lp_out[i] = p.likelihood_only ? -0.5 * total : -0.5 * total + (-0.5 * ((P[0] + P[1]) + (P[2] + P[3])));
}

/* the wide path's elementwise kernels (WideArgs, ymc_tile.h) */
__global__ __launch_bounds__(256) void wide_propose_kernel(const WideArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.K * (uint64_t)a.d) return;
    const int j = (int)(i % (uint64_t)a.d);
    const double x = a.from[i], xi = a.xi[i];
    double y;
    if (a.pcn) { /* pcn.py:27-36 as linear_tile_kernel: xi ~ prior, y = sqrt(1 - 2h) x + sqrt(2h) xi */
        const double p0 = a.pmean[j] + a.sigma[j] * xi;
        y = a.pcn_c1 * x + a.pcn_c2 * p0;
    } else {
        y = x + a.sigma[j] * xi; /* mrw.py:32-38 */
    }
    a.y[i] = y;
    const uint64_t k = i / (uint64_t)a.d;
    if (!(y == x)) a.differs[k] = 1u; /* metropolisHastings.py:60-61, gathered per chain: every writer stores the same 1 */
    a.y_tiled[((k / 16) * (uint64_t)a.KS + (uint64_t)(j / 4)) * 64 + (uint64_t)(j % 4) * 16 + k % 16] = y;
}

/* rng_fill_kernel + wide_propose_kernel in one: thread (k, j) draws the normal pair j of chain k's transition t and writes
 * components 2 j, 2 j + 1 of the proposal (plain and tiled); thread (k, 0) also draws the accept uniform */
__global__ __launch_bounds__(256) void wide_draw_propose_kernel(const WideArgs a, uint64_t seed, uint64_t chain0, uint64_t t)
{
    ymc_stage_tables();
    const uint64_t pairs = ((uint64_t)a.d + 1) / 2;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.K * pairs) return;
    const uint64_t k = i / pairs;
    const int jp = (int)(i % pairs);
    const uint32_t chain = (uint32_t)(chain0 + k);
    double z[2];
    ymc_draw_normal_pair(seed, chain, t, (uint32_t)jp, &z[0], &z[1]);
    if (jp == 0) const_cast<double *>(a.u)[k] = ymc_draw_uniform(seed, chain, t);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = 2 * jp + e;
        if (j < a.d) {
            const uint64_t idx = k * (uint64_t)a.d + (uint64_t)j;
            const double x = a.from[idx];
            double y;
            if (a.pcn) {
                const double p0 = a.pmean[j] + a.sigma[j] * z[e];
                y = a.pcn_c1 * x + a.pcn_c2 * p0;
            } else {
                y = x + a.sigma[j] * z[e];
            }
            a.y[idx] = y;
            if (!(y == x)) a.differs[k] = 1u;
            a.y_tiled[((k / 16) * (uint64_t)a.KS + (uint64_t)(j / 4)) * 64 + (uint64_t)(j % 4) * 16 + k % 16] = y;
        }
    }
}

__global__ __launch_bounds__(256) void wide_draws_kernel(const WideArgs a, uint64_t seed, uint64_t chain0, uint64_t t,
                                                         double *__restrict__ xi_tiled)
{
    ymc_stage_tables();
    const uint64_t pairs = ((uint64_t)a.d + 1) / 2;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.K * pairs) return;
    const uint64_t k = i / pairs;
    const int jp = (int)(i % pairs);
    const uint32_t chain = (uint32_t)(chain0 + k);
    double z[2];
    ymc_draw_normal_pair(seed, chain, t, (uint32_t)jp, &z[0], &z[1]);
    if (jp == 0) const_cast<double *>(a.u)[k] = ymc_draw_uniform(seed, chain, t);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = 2 * jp + e;
        if (j < a.d) xi_tiled[((k / 16) * (uint64_t)a.KS + (uint64_t)(j / 4)) * 64 + (uint64_t)(j % 4) * 16 + k % 16] = z[e];
    }
}

/* The linear posterior of 16 points per wave at ANY parameter dimension: the matrix-core kernels' evaluation (lane (g, c)
 * owns components 4 s + g of point c; k ascending per accumulator; the residual sums and the prior in the canonical four
 * partial sums) with the point read from its tiled array instead of registers and the matrix's fragments from global
 * memory -- two 8-byte loads per MFMA and lane, L2-bound. */
template <int TILES>
__global__ __launch_bounds__(512) void linear_wide_eval_kernel(const double *__restrict__ blob, int KS, int MT, double lik_const,
                                                               int likelihood_only, const double *__restrict__ y_tiled,
                                                               uint64_t K, double *__restrict__ lp_out)
{
    /* TILES = 2 tiles of 16 points per wave: a fragment of the matrix feeds two MFMAs (the matrix is what a wave reads most
     * of); TILES = 1 for small ensembles, where waves are what is short (the second tile is then the first again) */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const uint64_t nct = (K + 15) / 16, tile0 = ((uint64_t)blockIdx.x * 8 + wave) * TILES;
    if (tile0 >= nct) return;
    const bool two = TILES == 2 && tile0 + 1 < nct;
    const int m_pad = 16 * MT, d_pad = 4 * KS;
    const double *bias = blob + (size_t)MT * KS * 64, *nprec = bias + m_pad, *dbar = nprec + m_pad;
    const double *pmean = dbar + m_pad, *pprec = pmean + d_pad;
    const double *ya = y_tiled + (tile0 * (uint64_t)KS) * 64 + lane;
    const double *yb = two ? ya + (size_t)KS * 64 : ya; /* a last odd tile is evaluated twice and written once */
    double Sa = 0.0, Sb = 0.0;
#pragma unroll 1
    for (int mt = 0; mt < MT; mt += 2) {
        d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0}, b0 = {0.0, 0.0, 0.0, 0.0}, b1 = {0.0, 0.0, 0.0, 0.0};
        const double *A0 = blob + (size_t)mt * KS * 64 + lane, *A1 = A0 + (size_t)KS * 64;
#pragma unroll 1
        for (int s0 = 0; s0 < KS; s0 += 4) { /* KS is a multiple of 4 here: four steps' loads in flight together */
            double va[4], vb[4], f0[4], f1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                va[q] = ya[(s0 + q) * 64], vb[q] = yb[(s0 + q) * 64];
                f0[q] = A0[(s0 + q) * 64], f1[q] = A1[(s0 + q) * 64];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0[q], va[q], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1[q], va[q], a1, 0, 0, 0);
                b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0[q], vb[q], b0, 0, 0, 0);
                b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1[q], vb[q], b1, 0, 0, 0);
            }
        }
        ymc_tile_add_residuals(mt, a0, a1, bias, nprec, dbar, g, Sa);
        ymc_tile_add_residuals(mt, b0, b1, bias, nprec, dbar, g, Sb);
    }
    const double logl_a = -0.5 * (combine4(Sa) + lik_const), logl_b = -0.5 * (combine4(Sb) + lik_const);
    double Pa = 0.0, Pb = 0.0;
    for (int s = 0; s < KS; ++s) {
        const int j = 4 * s + g;
        const double pm = pmean[j], pp = pprec[j];
        const double dxa = ya[s * 64] - pm, dxb = yb[s * 64] - pm;
        Pa = Pa + dxa * (pp * dxa);
        Pb = Pb + dxb * (pp * dxb);
    }
    const double lpa = likelihood_only ? logl_a : logl_a + (-0.5 * combine4(Pa));
    const double lpb = likelihood_only ? logl_b : logl_b + (-0.5 * combine4(Pb));
    const uint64_t ka = tile0 * 16 + c, kb = ka + 16;
    if (g == 0 && ka < K) lp_out[ka] = lpa;
    if (g == 0 && two && kb < K) lp_out[kb] = lpb;
}

/* out[k][i] = sum_j L[i][j] in[k][j]: a wave per tile of 16 points and pair of output tiles (ymc_launch_wide_rows) */
__global__ __launch_bounds__(512) void linear_wide_rows_kernel(const double *__restrict__ frag, int KS, int MT_L, int d,
                                                               const double *__restrict__ in_tiled, uint64_t K,
                                                               double *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const uint64_t nct = (K + 15) / 16, pairs = (uint64_t)MT_L / 2, w = (uint64_t)blockIdx.x * 8 + wave;
    if (w >= nct * pairs) return;
    const uint64_t tile = w / pairs;
    const int mt = 2 * (int)(w % pairs);
    const double *v = in_tiled + (tile * (uint64_t)KS) * 64 + lane;
    const uint64_t k = tile * 16 + c;
    d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
    const double *L0 = frag + (size_t)mt * KS * 64 + lane, *L1 = L0 + (size_t)KS * 64;
#pragma unroll 1
    for (int s0 = 0; s0 < KS; s0 += 4) {
        double vv[4], f0[4], f1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) vv[q] = v[(s0 + q) * 64], f0[q] = L0[(s0 + q) * 64], f1[q] = L1[(s0 + q) * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0[q], vv[q], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1[q], vv[q], a1, 0, 0, 0);
        }
    }
    if (k < K) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i0 = 16 * mt + 4 * r + g, i1 = i0 + 16;
            if (i0 < d) out[k * (uint64_t)d + i0] = a0[r];
            if (i1 < d) out[k * (uint64_t)d + i1] = a1[r];
        }
    }
}

/* Few points (a handful of tiles): a wave per tile AND pair of output tiles writes the pair's eight residual terms per lane,
 * a second kernel adds them in the canonical order (the order linear_wide_eval_kernel adds them in: same bits) */
__global__ __launch_bounds__(256) void linear_wide_terms_kernel(const double *__restrict__ blob, int KS, int MT,
                                                                const double *__restrict__ y_tiled, uint64_t nct,
                                                                double *__restrict__ terms)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const uint64_t pairs = (uint64_t)MT / 2, w = (uint64_t)blockIdx.x * 4 + wave;
    if (w >= nct * pairs) return;
    const uint64_t tile = w / pairs;
    const int mt = 2 * (int)(w % pairs);
    const int m_pad = 16 * MT;
    const double *bias = blob + (size_t)MT * KS * 64, *nprec = bias + m_pad, *dbar = nprec + m_pad;
    const double *yt = y_tiled + (tile * (uint64_t)KS) * 64 + lane;
    d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
    const double *A0 = blob + (size_t)mt * KS * 64 + lane, *A1 = A0 + (size_t)KS * 64;
#pragma unroll 1
    for (int s0 = 0; s0 < KS; s0 += 4) {
        double v[4], f0[4], f1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = yt[(s0 + q) * 64], f0[q] = A0[(s0 + q) * 64], f1[q] = A1[(s0 + q) * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0[q], v[q], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1[q], v[q], a1, 0, 0, 0);
        }
    }
    double *out = terms + (w * 8) * 64 + lane;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { /* ymc_tile_add_residuals' terms, one by one */
            const int m = 16 * (mt + h) + 4 * r + g;
            const double F = (h ? a1[r] : a0[r]) + bias[m];
            const double res = F - dbar[m];
            const double px = nprec[m] * res;
            out[(4 * h + r) * 64] = res * px;
        }
    }
}

__global__ __launch_bounds__(256) void linear_wide_sum_kernel(const double *__restrict__ blob, int KS, int MT, double lik_const,
                                                              int likelihood_only, const double *__restrict__ y_tiled,
                                                              const double *__restrict__ terms, uint64_t K,
                                                              double *__restrict__ lp_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const uint64_t nct = (K + 15) / 16, tile = (uint64_t)blockIdx.x * 4 + wave;
    if (tile >= nct) return;
    const int m_pad = 16 * MT, d_pad = 4 * KS;
    const double *pmean = blob + (size_t)MT * KS * 64 + 3 * (size_t)m_pad, *pprec = pmean + d_pad;
    const double *yt = y_tiled + (tile * (uint64_t)KS) * 64 + lane;
    const double *tp = terms + (tile * ((uint64_t)MT / 2) * 8) * 64 + lane;
    double Ssum = 0.0;
    for (int i = 0; i < 4 * MT; ++i) Ssum = Ssum + tp[(size_t)i * 64];
    const double logl = -0.5 * (combine4(Ssum) + lik_const);
    double P = 0.0;
    for (int s = 0; s < KS; ++s) {
        const int j = 4 * s + g;
        const double dx = yt[s * 64] - pmean[j];
        P = P + dx * (pprec[j] * dx);
    }
    const double lp = likelihood_only ? logl : logl + (-0.5 * combine4(P));
    const uint64_t k = tile * 16 + c;
    if (g == 0 && k < K) lp_out[k] = lp;
}

__global__ __launch_bounds__(256) void wide_decide_kernel(const WideArgs a)
{
    ymc_stage_tables();
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.K) return;
    const bool moved = a.differs[k] != 0; /* set by the proposal kernel (a scan of the two rows here, one lane per chain,
                                             touched a cache line per lane and component) */
    a.differs[k] = 0;
    uint32_t st = a.status[k];
    const double lpy = a.lpy[k], lp = a.cur_lp[k];
    const bool acc = ymc_accept(lpy - lp, moved, ymc_finite(lpy), a.u[k], st); /* mrw.py:51-57 */
    a.accflag[k] = acc ? 1 : 0;
    a.cur_lp[k] = acc ? lpy : lp;
    if (a.count) a.count[k] += acc ? 1u : 0u;
    a.status[k] = st;
    if (a.acc_row) a.acc_row[k] = acc ? 1 : 0;
    if (a.lp_row) a.lp_row[k] = acc ? lpy : lp;
}

__global__ __launch_bounds__(256) void wide_apply_kernel(const WideArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.K * (uint64_t)a.d) return;
    const uint64_t k = i / (uint64_t)a.d;
    const int j = (int)(i % (uint64_t)a.d);
    double x = a.dst[i];
    if (a.wf_mean) { /* the state BEFORE the move, as the lane kernels (estimation.py:36-53) */
        double mean = a.wf_mean[i], m2 = a.wf_m2[i];
        const double delta = x - mean;
        mean = mean + ymc_div_by_recip(delta, a.wf_dn, a.wf_rdn);
        const double delta2 = x - mean;
        m2 = m2 + delta * delta2;
        a.wf_mean[i] = mean;
        a.wf_m2[i] = m2;
    }
    x = a.accflag[k] ? a.y[i] : x;
    a.dst[i] = x;
    if (a.dst_tiled) a.dst_tiled[((k / 16) * (uint64_t)a.KS + (uint64_t)(j / 4)) * 64 + (uint64_t)(j % 4) * 16 + k % 16] = x;
    if (a.traj_row) a.traj_row[i] = x;
}

/* dst (and its matrix-core copy) <- src, [K][d] */
__global__ __launch_bounds__(256) void wide_copy_kernel(const WideArgs a, const double *__restrict__ src, double *__restrict__ dst,
                                                        double *__restrict__ dst_tiled)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.K * (uint64_t)a.d) return;
    const uint64_t k = i / (uint64_t)a.d;
    const int j = (int)(i % (uint64_t)a.d);
    const double v = src[i];
    dst[i] = v;
    if (dst_tiled) dst_tiled[((k / 16) * (uint64_t)a.KS + (uint64_t)(j / 4)) * 64 + (uint64_t)(j % 4) * 16 + k % 16] = v;
}

__global__ __launch_bounds__(256) void wide_differs_kernel(const WideArgs a, const double *__restrict__ p, const double *__restrict__ q)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.K * (uint64_t)a.d) return;
    if (!(p[i] == q[i])) a.differs[i / (uint64_t)a.d] = 1u;
}

/* the fine accept / reject of two-level delayed acceptance (mlda.py:146-154, as mlda2_body): the sub-chain's end z against
 * the fine state x with exp(P(z) + Q(x) - Q(z) - P(x)) */
__global__ __launch_bounds__(256) void wide_fine_decide_kernel(const WideArgs a)
{
    ymc_stage_tables();
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.K) return;
    const bool moved = a.differs[k] != 0; /* set by wide_differs_kernel (z against x) */
    a.differs[k] = 0;
    uint32_t st = a.status[k];
    const double lpfy = a.lpfz[k], lpf = a.lp[k], lpc = a.lpc[k], lpcz = a.lpcz[k];
    const double log_ratio = ((lpfy + lpc) - lpcz) - lpf;
    const bool acc = ymc_accept(log_ratio, moved, ymc_finite(lpfy) && ymc_finite(lpcz), a.u[k], st);
    a.accflag[k] = acc ? 1 : 0;
    a.lp[k] = acc ? lpfy : lpf;
    a.lpc[k] = acc ? lpcz : lpc;
    a.acc_count[k] += acc ? 1u : 0u;
    a.status[k] = st;
    if (a.acc_row) a.acc_row[k] = acc ? 1 : 0;
    if (a.lp_row) a.lp_row[k] = acc ? lpfy : lpf;
}

__global__ __launch_bounds__(256) void wide_fine_apply_kernel(const WideArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.K * (uint64_t)a.d) return;
    const uint64_t k = i / (uint64_t)a.d;
    double x = a.x[i];
    if (a.wf_mean) {
        double mean = a.wf_mean[i], m2 = a.wf_m2[i];
        const double delta = x - mean;
        mean = mean + ymc_div_by_recip(delta, a.wf_dn, a.wf_rdn);
        const double delta2 = x - mean;
        m2 = m2 + delta * delta2;
        a.wf_mean[i] = mean;
        a.wf_m2[i] = m2;
    }
    x = a.accflag[k] ? a.z[i] : x;
    a.x[i] = x;
    if (a.dst_tiled) {
        const int j = (int)(i % (uint64_t)a.d);
        a.dst_tiled[((k / 16) * (uint64_t)a.KS + (uint64_t)(j / 4)) * 64 + (uint64_t)(j % 4) * 16 + k % 16] = x;
    }
    if (a.traj_row) a.traj_row[i] = x;
}

}  // namespace

hipError_t ymc_launch_wide_propose(const WideArgs &a, hipStream_t s)
{
    const uint64_t n = a.K * (uint64_t)a.d;
    hipLaunchKernelGGL(wide_propose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t ymc_launch_wide_draw_propose(const WideArgs &a, uint64_t seed, uint64_t chain0, uint64_t t, hipStream_t s)
{
    const uint64_t n = a.K * (((uint64_t)a.d + 1) / 2);
    hipLaunchKernelGGL(wide_draw_propose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, seed, chain0, t);
    return hipGetLastError();
}

hipError_t ymc_launch_wide_draws(const WideArgs &a, uint64_t seed, uint64_t chain0, uint64_t t, double *xi_tiled, hipStream_t s)
{
    const uint64_t n = a.K * (((uint64_t)a.d + 1) / 2);
    hipLaunchKernelGGL(wide_draws_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, seed, chain0, t, xi_tiled);
    return hipGetLastError();
}

hipError_t ymc_launch_wide_copy(const WideArgs &a, const double *src, double *dst, double *dst_tiled, hipStream_t s)
{
    const uint64_t n = a.K * (uint64_t)a.d;
    hipLaunchKernelGGL(wide_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, src, dst, dst_tiled);
    return hipGetLastError();
}

hipError_t ymc_launch_wide_differs(const WideArgs &a, const double *p, const double *q, hipStream_t s)
{
    const uint64_t n = a.K * (uint64_t)a.d;
    hipLaunchKernelGGL(wide_differs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, p, q);
    return hipGetLastError();
}

hipError_t ymc_launch_wide_fine_decide(const WideArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(wide_fine_decide_kernel, dim3((unsigned)((a.K + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t ymc_launch_wide_fine_apply(const WideArgs &a, hipStream_t s)
{
    const uint64_t n = a.K * (uint64_t)a.d;
    hipLaunchKernelGGL(wide_fine_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

constexpr uint64_t kWideSplitTiles = 128; /* fewer tiles than this: a wave per tile and pair of output tiles */

size_t ymc_wide_eval_terms(int MT, uint64_t K)
{
    const uint64_t nct = (K + 15) / 16;
    return nct < kWideSplitTiles ? (size_t)(nct * ((uint64_t)MT / 2) * 8 * 64) : 0;
}

hipError_t ymc_launch_wide_eval(const double *blob, int KS, int MT, double lik_const, int likelihood_only, const double *y_tiled,
                                uint64_t K, double *lp_out, hipStream_t s, double *terms)
{
    const uint64_t nct = (K + 15) / 16;
    if (terms && nct < kWideSplitTiles) {
        const uint64_t waves = nct * ((uint64_t)MT / 2);
        hipLaunchKernelGGL(linear_wide_terms_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, blob, KS, MT, y_tiled, nct, terms);
        hipLaunchKernelGGL(linear_wide_sum_kernel, dim3((unsigned)((nct + 3) / 4)), dim3(256), 0, s, blob, KS, MT, lik_const,
                           likelihood_only, y_tiled, terms, K, lp_out);
        return hipGetLastError();
    }
    if (nct >= 2048) /* two tiles per wave once every SIMD has a wave anyway */
        hipLaunchKernelGGL(linear_wide_eval_kernel<2>, dim3((unsigned)((nct + 15) / 16)), dim3(512), 0, s, blob, KS, MT, lik_const,
                           likelihood_only, y_tiled, K, lp_out);
    else
        hipLaunchKernelGGL(linear_wide_eval_kernel<1>, dim3((unsigned)((nct + 7) / 8)), dim3(512), 0, s, blob, KS, MT, lik_const,
                           likelihood_only, y_tiled, K, lp_out);
    return hipGetLastError();
}

hipError_t ymc_launch_wide_rows(const double *frag, int KS, int MT_L, int d, const double *in_tiled, uint64_t K, double *out,
                                hipStream_t s)
{
    const uint64_t waves = ((K + 15) / 16) * ((uint64_t)MT_L / 2);
    hipLaunchKernelGGL(linear_wide_rows_kernel, dim3((unsigned)((waves + 7) / 8)), dim3(512), 0, s, frag, KS, MT_L, d, in_tiled, K, out);
    return hipGetLastError();
}

hipError_t ymc_launch_wide_decide(const WideArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(wide_decide_kernel, dim3((unsigned)((a.K + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t ymc_launch_wide_apply(const WideArgs &a, hipStream_t s)
{
    const uint64_t n = a.K * (uint64_t)a.d;
    hipLaunchKernelGGL(wide_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

size_t ymc_tile_lds_bytes(int KS, int MT)
{
    const size_t m_pad = 16 * (size_t)MT, d_pad = 4 * (size_t)KS;
    return ((size_t)MT * KS * 64 + m_pad * 3 + 3 * d_pad) * sizeof(double);  /* + 2 KB static (log table) */
}

size_t ymc_tile_stream_lds_bytes(int KS, int MT)
{
    const size_t m_pad = 16 * (size_t)MT, d_pad = 4 * (size_t)KS;
    const size_t unit = (size_t)2 * (KS / 4) * 64;     /* ymc_stream_unit<KS, 4>() */
    return (m_pad * 3 + 3 * d_pad + 2 * unit) * sizeof(double);
}

template <int KS, int PROP, int WAVES = kTileWaves, bool ASTREAM = false>
static hipError_t launch_tile_as(const TileArgs &a, hipStream_t s, size_t extra_lds = 0)
{
    const size_t lds = ASTREAM ? ymc_tile_stream_lds_bytes(KS, a.MT) : ymc_tile_lds_bytes(KS, a.MT) + extra_lds;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_tile_kernel<KS, PROP, WAVES, ASTREAM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t nct = (a.K + 15) / 16;
    const unsigned grid = (unsigned)((nct + WAVES - 1) / WAVES);
    hipLaunchKernelGGL((linear_tile_kernel<KS, PROP, WAVES, ASTREAM>), dim3(grid), dim3(64 * WAVES), lds, s, a);
    return hipGetLastError();
}

size_t ymc_tile_mlda2_lds_bytes(int KS, int MT_coarse, int MT_fine)
{
    return ymc_tile_lds_bytes(KS, MT_coarse) + (16 * (size_t)MT_fine * 3 + 2 * 4 * (size_t)KS) * sizeof(double);
}

/* The builds of the two-level kernel (identical results; YMC_TILE_MLDA2=mem|regs|lds|plain and YMC_TILE_STAGE=0
 * override the choice, the tests run all of them):
 *   mem    the coarse matrix resident in LDS, the fine one streamed, 8 waves (two per SIMD: z and y in 128 of their
 *          256 registers); the fine state is read back from the record row written last, once per fine step
 *   lds    both matrices stream through the shared double buffer, the fine state in LDS, 8 waves (two per SIMD,
 *          z and y in 128 of their 256 registers) -- KS = 32 when the constants leave the room
 *   regs   the coarse matrix resident in LDS, the fine one streamed, the fine state in registers; 4 waves at
 *          KS = 32 (one per SIMD, 512 registers), 8 below
 *   plain  as regs, every wave reading the fine matrix from L2 itself (no room for the double buffer)
 *   stream both matrices stream, the fine state in memory as in mem; 4 waves at KS >= 32: what runs where the coarse
 *          matrix does not fit the LDS, and at KS = 64 (d <= 256: z and y alone are 256 registers) */
template <int KS>
constexpr int mlda2_waves() { return KS >= 32 ? 4 : 8; }

template <int KS, bool STAGED, int XM>
static hipError_t launch_tile_mlda2_as(const TileArgs &a, size_t lds, hipStream_t s)
{
    constexpr int W = (XM == kXRegs || XM == kXMemC) ? mlda2_waves<KS>() : 8;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_tile_mlda2_kernel<KS, STAGED, W, XM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t nct = (a.K + 15) / 16;
    const unsigned grid = (unsigned)((nct + W - 1) / W);
    hipLaunchKernelGGL((linear_tile_mlda2_kernel<KS, STAGED, W, XM>), dim3(grid), dim3(64 * W), lds, s, a);
    return hipGetLastError();
}

template <int KS>
static hipError_t launch_tile_mlda2(const TileArgs &a, hipStream_t s)
{
    const size_t unit2 = 2 * (size_t)ymc_stream_unit<KS, 4>() * sizeof(double);
    const size_t consts = ymc_tile_mlda2_lds_bytes(KS, a.MT, a.MT_f) - (size_t)a.MT * KS * 64 * sizeof(double);
    const size_t plain = ymc_tile_mlda2_lds_bytes(KS, a.MT, a.MT_f), staged = plain + unit2;
    {
        const char *env = std::getenv("YMC_TILE_MLDA2");
        if (KS >= 64 || plain > kTileLdsLimit || (env && env[0] == 's')) {
            if (consts + unit2 > kTileLdsLimit) return hipErrorInvalidValue;
            return launch_tile_mlda2_as<KS, true, kXMemC>(a, consts + unit2, s);
        }
    }
    if constexpr (KS < 64) {
        const size_t xlds = consts + unit2 + (size_t)8 * KS * 64 * sizeof(double);
        bool stage = staged <= kTileLdsLimit;
        if (const char *env = std::getenv("YMC_TILE_STAGE")) stage = stage && env[0] != '0';
        char mode = 'r';
        if (const char *env = std::getenv("YMC_TILE_MLDA2")) mode = env[0];
        if (mode == 'p') stage = false;
        if (mode == 'l' && stage && xlds <= kTileLdsLimit) return launch_tile_mlda2_as<KS, true, kXLds>(a, xlds, s);
        if (mode == 'm')
            return stage ? launch_tile_mlda2_as<KS, true, kXMem>(a, staged, s) : launch_tile_mlda2_as<KS, false, kXMem>(a, plain, s);
        return stage ? launch_tile_mlda2_as<KS, true, kXRegs>(a, staged, s) : launch_tile_mlda2_as<KS, false, kXRegs>(a, plain, s);
    }
    return hipErrorInvalidValue;
}

#ifdef YMC_TILE_PROFILE
extern "C" int ymc_debug_tile_sections(unsigned long long *out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(ymc_tile_prof), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -2;
}
#endif

hipError_t ymc_launch_linear_tile_mlda2(int KS, const TileArgs &a, hipStream_t s)
{
    switch (KS) {
    case 4: return launch_tile_mlda2<4>(a, s);
    case 8: return launch_tile_mlda2<8>(a, s);
    case 16: return launch_tile_mlda2<16>(a, s);
    case 32: return launch_tile_mlda2<32>(a, s);
    case 64: return launch_tile_mlda2<64>(a, s);
    default: return hipErrorInvalidValue;
    }
}

size_t ymc_tile_mlda2_stream_lds_bytes(int KS, int MT_coarse, int MT_fine)
{
    const size_t unit2 = 2 * (size_t)2 * (KS / 4) * 64 * sizeof(double);     /* 2 ymc_stream_unit<KS, 4>() */
    return ymc_tile_mlda2_lds_bytes(KS, MT_coarse, MT_fine) - (size_t)MT_coarse * KS * 64 * sizeof(double) + unit2;
}

template <int KS>
static hipError_t launch_tile(const TileArgs &a, hipStream_t s)
{
    /* the forward matrix streamed from L2 where it does not fit the LDS (YMC_TILE_ASTREAM=1 forces it: the tests
     * compare both forms); one wave per SIMD at KS = 64 */
    bool astream = KS >= 64 || ymc_tile_lds_bytes(KS, a.MT) > kTileLdsLimit;
    if (const char *env = std::getenv("YMC_TILE_ASTREAM")) astream = astream || env[0] == '1';
    if (astream) {
        constexpr int W = KS >= 64 ? 4 : kTileWaves;
        if (a.chol_frag) /* the proposal's factor, then the forward matrix, through the one double buffer */
            return a.pcn ? hipErrorInvalidValue : launch_tile_as<KS, kPropDenseStaged, W, true>(a, s);
        return a.pcn ? launch_tile_as<KS, kPropPcn, W, true>(a, s) : launch_tile_as<KS, kPropDiag, W, true>(a, s);
    }
    if constexpr (KS < 64) {
        if (a.chol_frag) {
            if (a.pcn) return hipErrorInvalidValue;
            /* L through the shared double buffer when the LDS has room beside the forward map (identical results;
             * YMC_TILE_STAGE=0 switches it off, the tests run both) */
            const size_t extra = 2 * (size_t)ymc_stream_unit<KS, 4>() * sizeof(double);
            bool stage = ymc_tile_lds_bytes(KS, a.MT) + extra <= kTileLdsLimit;
            if (const char *env = std::getenv("YMC_TILE_STAGE")) stage = stage && env[0] != '0';
            return stage ? launch_tile_as<KS, kPropDenseStaged>(a, s, extra) : launch_tile_as<KS, kPropDense>(a, s);
        }
        return a.pcn ? launch_tile_as<KS, kPropPcn>(a, s) : launch_tile_as<KS, kPropDiag>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t ymc_launch_linear_tile(int KS, const TileArgs &a, hipStream_t s)
{
    switch (KS) {
    case 4: return launch_tile<4>(a, s);
    case 8: return launch_tile<8>(a, s);
    case 16: return launch_tile<16>(a, s);
    case 32: return launch_tile<32>(a, s);
    case 64: return launch_tile<64>(a, s);
    default: return hipErrorInvalidValue;
    }
}

hipError_t ymc_launch_untile(const double *tiled, uint64_t row0, uint64_t rows, uint64_t K, int d, int KS,
                             double *out, hipStream_t s)
{
    const uint64_t n = rows * K * (uint64_t)d;
    hipLaunchKernelGGL(untile_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tiled, row0, rows, K, d, KS,
                       out);
    return hipGetLastError();
}

hipError_t ymc_launch_welford_replay(const double *tiled, const double *x_before, uint64_t rows, uint64_t K, int d,
                                     int KS, uint64_t wf_n0, double *wf_mean, double *wf_m2, hipStream_t s)
{
    const uint64_t per_row = ((K + 15) / 16) * (uint64_t)KS * 64;
    hipLaunchKernelGGL(welford_replay_kernel, dim3((unsigned)((per_row + 255) / 256)), dim3(256), 0, s, tiled, x_before,
                       rows, K, d, KS, wf_n0, wf_mean, wf_m2);
    return hipGetLastError();
}

hipError_t ymc_launch_linear_big_eval(const BigLinearArgs &p, uint64_t n, const double *x, double *lp, double *fwd,
                                      hipStream_t s)
{
    hipLaunchKernelGGL(linear_big_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, x, lp, fwd);
    return hipGetLastError();
}
