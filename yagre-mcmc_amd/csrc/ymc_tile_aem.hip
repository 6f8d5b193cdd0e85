/*
 * ymc_tile_aem.hip -- two-level delayed acceptance with the ADAPTIVE ERROR MODEL over two linear models at
 * large dimension (d, data_dim <= 128): the matrix-core counterpart of ymc_aem.hip
 * (chain/method/aem.py:6-58; statistics/likelihood.py:90-155; noise.py:25-61; utility/memoisation.py:76-149).
 *
 * Mapping of linear_tile_mlda2_kernel (one wave = 16 chains, lane (g, c) owns chain c and the components /
 * outputs congruent g mod 4; coarse matrix in LDS, fine matrix fragments from L2).  What a chain's coarse
 * AEMLikelihood object holds lives with its four lanes:
 *   - the Welford accumulator of the model error F_fine(y) - F_coarse(y) over the ACCEPTED fine proposals,
 *     the residual shift and the inflated noise precision: one value per output, so lane (g, c) keeps the
 *     entries of ITS outputs in arrays laid out like the record ([tile][slot][lane], coalesced), read at every
 *     coarse evaluation and rewritten when the chain accepts;
 *   - the three-entry AEMCache of log-likelihood values, most recently used last (a hit returns the value
 *     stored when the entry was made, possibly under an older error model -- the reference's chains depend on
 *     that).  Its keys are parameter vectors; comparing 128-component keys at every query would cost more
 *     than the likelihood, so a key is the IDENTITY of the point: every proposal is a new point (id = its
 *     transition index + 1, the initial state has id 0), accepting hands the id on.  Two ids are equal iff
 *     the vectors are the same point of the walk; distinct proposals that coincide in all d components have
 *     probability zero, and a proposal equal to its state is rejected before any query (as in the reference).
 * A miss needs the likelihood under the CURRENT error model: it is evaluated for the whole wave whenever
 * any of its chains misses (a ballot), and only the chains that missed take the value.
 * Same draws and the same arithmetic in the same order as ymo_aem_advance / aem2_kernel: bit-identical.
 */
#define YMC_COEFF_IN_SGPR 1
#include <cstdlib>

#include "ymc_kernels.h"
#include "ymc_tile.h"
#include "ymc_tile_dev.h"

namespace {

constexpr int kAemRows = 8; /* data rows whose partial sums a lane keeps in registers */
/* four waves per workgroup = one per SIMD: sub-chain state, proposal, the row sums and the cache take more
 * than the 256 registers a wave has at two per SIMD (345 spills at d = 128) */
constexpr int kAemWaves = 4, kAemThreads = 64 * kAemWaves;

/* the three-entry cache in NAMED registers: with arrays the one run-time index (put at position nkeys, a per-lane
 * value) made the whole object private memory -- 64 B of scratch per lane and a memory round trip per cache
 * operation on the critical path of every evaluation */
struct AemCache {
    int32_t nkeys;
    uint64_t k0, k1, k2;
    double v0, v1, v2;

    __device__ __forceinline__ int find(uint64_t id) const
    {
        if (nkeys > 0 && k0 == id) return 0;
        if (nkeys > 1 && k1 == id) return 1;
        if (nkeys > 2 && k2 == id) return 2;
        return -1;
    }
    /* AEMLikelihood.query_log_likelihood + update_cache (likelihood.py:121-126,135-141) given where the key
     * was found (-1: not) and, for a miss, the likelihood under the current error model */
    __device__ __forceinline__ double take(uint64_t id, int hit, double fresh)
    {
        if (hit == 0) { /* retrieve: the entry moves to the back (memoisation.py:98-103) */
            const double v = v0;
            if (nkeys == 2) {
                k0 = k1, v0 = v1;
                k1 = id, v1 = v;
            } else if (nkeys == 3) {
                k0 = k1, v0 = v1;
                k1 = k2, v1 = v2;
                k2 = id, v2 = v;
            }
            return v;
        }
        if (hit == 1) {
            const double v = v1;
            if (nkeys == 3) {
                k1 = k2, v1 = v2;
                k2 = id, v2 = v;
            }
            return v;
        }
        if (hit == 2) return v2;
        if (nkeys == 3) { /* evict the oldest (memoisation.py:92-96) */
            k0 = k1, v0 = v1;
            k1 = k2, v1 = v2;
            k2 = id, v2 = fresh;
        } else { /* append at position nkeys */
            k0 = nkeys == 0 ? id : k0, v0 = nkeys == 0 ? fresh : v0;
            k1 = nkeys == 1 ? id : k1, v1 = nkeys == 1 ? fresh : v1;
            k2 = nkeys == 2 ? id : k2, v2 = nkeys == 2 ? fresh : v2;
            nkeys += 1;
        }
        return fresh;
    }
};

/* STG: 0 = every wave reads the fine matrix's fragments from L2 itself; 4 / 8 = the workgroup's four waves (in
 * lockstep: same S for all) share one stream of it through an LDS double buffer, in units of KS/4 or KS/8 k-steps
 * of a pair of output tiles (16 or 8 KB at KS = 32 -- the data rows leave 11 KB at d = m = 128), one barrier per
 * unit, the next unit fetched into registers while the MFMAs of the current one run.  With one wave per SIMD
 * nothing else hides the L2 latency (ymc_tile_nested.hip: +67 %).  Waves past the last tile stay in the loop for
 * the barriers and touch no memory; the update of the error model is then evaluated by every wave on every fine
 * step (it applies to accepted chains only, as before). */
template <int KS, int STG, bool ROWBLOCKS = false>
__global__ __launch_bounds__(kAemThreads) void linear_tile_aem2_kernel(const TileArgs a)
{
    constexpr bool STAGED = STG != 0;
    constexpr int kDiv = STAGED ? STG : 4;
    static_assert(KS % kDiv == 0, "staged units are whole k-steps");

    ymc_stage_tables();
    extern __shared__ double lds[];
    const int MT = a.MT, m_pad = 16 * MT, d_pad = 4 * KS, Q = 4 * MT; /* both levels have a.m outputs */
    /* coarse level: A fragments | bias | (n * noise_prec, unused) | (row mean, unused) | prior mean | prior prec | sigma */
    double *Ac = lds;
    double *bias_c = Ac + (size_t)MT * KS * 64;
    double *pmean_c = bias_c + 3 * m_pad;
    double *pprec_c = pmean_c + d_pad;
    double *sigma = pprec_c + d_pad;
    const size_t blob_c = (size_t)MT * KS * 64 + (size_t)m_pad * 3 + 3 * (size_t)d_pad;
    /* fine level constants: bias | n * noise_prec | row mean | prior mean | prior prec */
    double *bias_f = lds + blob_c;
    double *nprec_f = bias_f + m_pad;
    double *dbar_f = nprec_f + m_pad;
    double *pmean_f = dbar_f + m_pad;
    double *pprec_f = pmean_f + d_pad;
    const size_t consts_f = (size_t)m_pad * 3 + 2 * (size_t)d_pad;
    /* the coarse likelihood row by row: noise precision | data rows [n_data][m_pad] */
    double *nprec_c = bias_f + consts_f;
    double *data_c = nprec_c + m_pad;
    const size_t rows_c = (size_t)m_pad * (1 + (size_t)a.n_data);
    const double *Af = a.blob_f;
    double *stage = nprec_c + rows_c;                            /* STAGED: the stream's double buffer (ymc_tile_dev.h) */

    const int tid = threadIdx.x;
    ymc_stage_block<kAemThreads>(lds, a.blob, blob_c / 2, tid);
    for (size_t i = tid; i < consts_f; i += kAemThreads) bias_f[i] = a.blob_f[(size_t)MT * KS * 64 + i];
    for (size_t i = tid; i < rows_c; i += kAemThreads) nprec_c[i] = a.aem_rows[i];
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const uint64_t nct = (a.K + 15) / 16;
    const uint64_t tile_raw = (uint64_t)blockIdx.x * kAemWaves + wave;
    const bool wave_live = tile_raw < nct;
    if (!STAGED && !wave_live) return;
    const uint64_t tile = wave_live ? tile_raw : 0; /* a wave past the last tile reads tile 0's arrays and writes nothing */
    const uint64_t k = tile_raw * 16 + c;
    const bool live = wave_live && k < a.K;
    const uint64_t kk = live ? k : 0;
    const uint32_t chain = (uint32_t)(a.chain0 + kk);
    const int d = a.d, S = a.S, m_real = a.m, n_data = a.n_data;
    const uint32_t pair_of_g = (g == 0) ? 0u : (g == 1) ? 2u : (g == 2) ? 1u : 3u;
    double *xt = a.x_tiled + (tile * KS) * 64 + lane;
    /* this lane's column of the per-output error-model arrays: slot q = 4 mt + r <-> output 16 mt + 4 r + g */
    /* mean / M2 of the chain's error model: the copy it reads (em, e2) and the one the fine pass writes the updated
     * values into before the decision is known (em_o, e2_o); an accepting chain swaps them.  The run's arrays are the
     * first copy: a chain that ends a launch on the second one copies it back */
    double *const em_run = a.err_mean_t + (tile * Q) * 64 + lane, *const e2_run = a.err_m2_t + (tile * Q) * 64 + lane;
    double *em = em_run, *e2 = e2_run;
    double *em_o = a.err_mean_alt_t + (tile * Q) * 64 + lane, *e2_o = a.err_m2_alt_t + (tile * Q) * 64 + lane;
    double *ep = a.prec_t + (tile * Q) * 64 + lane;

    double z[KS], y[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int j = 4 * s + g;
        z[s] = (live && j < d) ? a.x[k * d + j] : 0.0;
        if (wave_live) xt[s * 64] = z[s];
    }
    double lpf = live ? a.lp[k] : 0.0;
    uint64_t accepted = live ? a.acc_count[k] : 0;
    uint32_t st = live ? a.status[k] : 0;
    uint64_t n_err = a.err_n[kk];
    bool has_prec = a.aem_flags[2 * kk] != 0;
    AemCache cache;
    cache.nkeys = a.aem_flags[2 * kk + 1];
    cache.k0 = a.cache_id[kk * 4 + 0], cache.v0 = a.cache_val[kk * 3 + 0];
    cache.k1 = a.cache_id[kk * 4 + 1], cache.v1 = a.cache_val[kk * 3 + 1];
    cache.k2 = a.cache_id[kk * 4 + 2], cache.v2 = a.cache_val[kk * 3 + 2];
    uint64_t x_id = a.cache_id[kk * 4 + 3];
    UniformCache uc;
    uc.valid = false;
    uc.pair = 0;
    RecordClock clk(a.rec_stride, a.rec_since);

    /* AEMLikelihood.compute_log_likelihood under the CURRENT error model of every chain (likelihood.py:77-84,
     * 128-133): residual F - D_i (+ error mean once n >= minDataSize), norm in the (inflated) noise precision,
     * row by row; every row's terms in the canonical four partial sums */
    auto now = [&](const double (&v)[KS]) {
        const bool shift = n_err >= (uint64_t)a.min_data;
        /* the rows' partial sums as two 4-vectors: an array here stays a private-memory object (64 B of scratch per
         * lane, a serialised load / add / store per row and output -- it made an evaluation 2.5 x as long as its matrix
         * product, profiles/r03_tile_experiments.txt), vector elements are always registers */
        static_assert(kAemRows == 8, "two 4-vectors of row sums");
        /* ROWBLOCKS (more than eight data rows): one pass over the forward map per block of eight rows -- the rows' sums
         * join the total in row order either way, the same bits a single pass would give.  Its own instantiation: the
         * loop's bookkeeping cost the eight-row kernel 13 % when it carried it (22 spilled registers at KS = 32) */
        double total = 0.0;
#pragma unroll 1
        for (int r0 = 0; r0 < (ROWBLOCKS ? n_data : 1); r0 += kAemRows) {
            const double *rows = ROWBLOCKS ? data_c + (size_t)r0 * m_pad : data_c;
            const int nr = ROWBLOCKS && n_data - r0 >= kAemRows ? kAemRows : n_data - r0;
            d4 SA = {0.0, 0.0, 0.0, 0.0}, SB = {0.0, 0.0, 0.0, 0.0};
            int row_off[kAemRows];
#pragma unroll
            for (int i = 0; i < kAemRows; ++i) row_off[i] = (i < nr ? i : nr - 1) * m_pad;
#pragma unroll 1
            for (int mt = 0; mt < MT; mt += 2) {
                d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                const double *A0 = Ac + (size_t)mt * KS * 64 + lane;
                const double *A1 = A0 + (size_t)KS * 64;
                /* this pair's entries of the chain's error model, requested BEFORE its 64 MFMAs: they come from HBM (134 MB
                 * of them at 65 536 chains) and with one wave per SIMD nothing else hides that latency -- read in the
                 * epilogue they made an evaluation six times as long as its matrix product (profiles/r03_c3a_*) */
                double em_v[8], ep_v[8];
#pragma unroll
                for (int hq = 0; hq < 8; ++hq) {
                    em_v[hq] = em[(4 * mt + hq) * 64];
                    ep_v[hq] = has_prec ? ep[(4 * mt + hq) * 64] : 0.0;
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A0[s * 64], v[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1[s * 64], v[s], acc1, 0, 0, 0);
                }
                /* the pair's eight outputs against the block's rows, all eight row sums whatever the row count (a sum
                 * beyond the last row repeats that row and is never read): as a conditional per row the compiler chained a
                 * branch, an LDS read and its wait per row and output -- 160 serial round trips per evaluation, 46.8 k
                 * cycles around a 17 k matrix product; here an output's eight reads are in flight together.  Without the
                 * error mean res + 0.0 is res (a -0.0 becomes +0.0; its square is +0.0 either way) */
                double gate = 0.0, gate_next = 0.0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = 16 * (mt + h) + 4 * r + g;
                        const double F = (h ? acc1[r] : acc0[r]) + bias_c[m];
                        const double mean = shift ? em_v[4 * h + r] : 0.0;
                        const double pe = has_prec ? ep_v[4 * h + r] : nprec_c[m];
                        double dv[kAemRows];
                        int mo = m;
                        /* this output's reads wait for the sums of the one before the previous: two outputs' values in
                         * registers at most, everything else free to move */
                        asm volatile("" : "+v"(mo) : "v"(gate));
#pragma unroll
                        for (int i = 0; i < kAemRows; ++i) dv[i] = rows[row_off[i] + mo];
#pragma unroll
                        for (int i = 0; i < kAemRows; ++i) {
                            const double res = (F - dv[i]) + mean;
                            const double px = pe * res;
                            if (i < 4) SA[i & 3] = SA[i & 3] + res * px;
                            else SB[i & 3] = SB[i & 3] + res * px;
                        }
                        gate = gate_next;
                        gate_next = SA[0];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kAemRows; ++i)
                if (i < nr) total = total + combine4(i < 4 ? SA[i & 3] : SB[i & 3]);
        }
        return -0.5 * total;
    };
    /* the fine forward map of v, a pair of output tiles at a time: emit(mt, acc0, acc1) receives the MFMA
     * accumulators of tiles mt and mt + 1 (k ascending per accumulator in both forms: the same bits).  STAGED:
     * EVERY wave of the workgroup calls this at the same point of the loop. */
    auto fine_pairs = [&](const double (&v)[KS], auto &&emit) {
        if constexpr (STAGED) {
            ymc_stream_tile_pairs<KS, kDiv, kAemThreads>(Af, MT / 2, stage, tid, lane, v,
                                                         [&](int pair, const d4 &acc0, const d4 &acc1) { emit(2 * pair, acc0, acc1); });
        } else {
#pragma unroll 1
            for (int mt = 0; mt < MT; mt += 2) {
                d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                const double *A0 = Af + (size_t)mt * KS * 64 + lane;
                const double *A1 = A0 + (size_t)KS * 64;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A0[s * 64], v[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1[s * 64], v[s], acc1, 0, 0, 0);
                }
                emit(mt, acc0, acc1);
            }
        }
    };
    auto logprior = [&](const double *pmean, const double *pprec, const double (&v)[KS]) {
        return ymc_tile_logprior<KS>(pmean, pprec, g, v);
    };
    auto any_differs = [&](const double (&p)[KS], const double (&q)[KS]) { return ymc_tile_any_differs<KS>(p, q); };
    /* one cache query of every chain of the wave (`active` lanes only): the current-model likelihood is
     * evaluated for the wave iff some active chain misses */
    auto query = [&](uint64_t id, const double (&v)[KS], bool active) {
        const int hit = cache.find(id);
        const bool miss = active && hit < 0;
        double fresh = 0.0;
        if (__builtin_amdgcn_ballot_w64(miss) != 0) fresh = now(v);
        return active ? cache.take(id, hit, fresh) : 0.0;
    };
    /* The fine accept step in ONE pass over the fine matrix: the plain fine likelihood of v (for the decision) and
     * update_error_estimate(F_fine(v) - F_coarse(v)) (aem.py:44-56, likelihood.py:142-150, estimation.py:36-53) prepared for
     * every chain -- the updated mean / M2 go to the copy the chain is not reading, the range of the marginal variances
     * is kept -- so that an accepting chain only swaps its copies (commit_error_model).  The two used to be two passes:
     * the second one, after the decision, streamed the fine matrix again for the same F_fine. */
    double upd_lo = 0.0, upd_hi = 0.0;
    bool upd_first = true;
    auto fine_likelihood_and_error = [&](const double (&v)[KS]) {
        const uint64_t n_new = n_err + 1;
        const double dn = (double)n_new, nm1 = (double)(n_new - 1);
        double lo = 0.0, hi = 0.0, Ssum = 0.0;
        bool first = true;
        fine_pairs(v, [&](int mt, const d4 &f0, const d4 &f1) {
            ymc_tile_add_residuals(mt, f0, f1, bias_f, nprec_f, dbar_f, g, Ssum);
            d4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
            const double *C0 = Ac + (size_t)mt * KS * 64 + lane, *C1 = C0 + (size_t)KS * 64;
            double em_v[8], e2_v[8];        /* requested before the coarse MFMAs, as in now() */
#pragma unroll
            for (int hq = 0; hq < 8; ++hq) {
                em_v[hq] = em[(4 * mt + hq) * 64];
                e2_v[hq] = e2[(4 * mt + hq) * 64];
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(C0[s * 64], v[s], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(C1[s * 64], v[s], c1, 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * (mt + h) + 4 * r + g, q = 4 * (mt + h) + r;
                    const double Ff = (h ? f1[r] : f0[r]) + bias_f[m];
                    const double Fc = (h ? c1[r] : c0[r]) + bias_c[m];
                    const double e = Ff - Fc;
                    double mean = em_v[4 * h + r], m2 = e2_v[4 * h + r];
                    const double delta = e - mean; /* estimation.py:36-53 */
                    mean = mean + delta / dn;
                    const double delta2 = e - mean;
                    m2 = m2 + delta * delta2;
                    if (wave_live && m < m_real) {
                        em_o[q * 64] = mean;
                        e2_o[q * 64] = m2;
                        const double mvar = m2 / nm1;
                        lo = (first || mvar < lo) ? mvar : lo;
                        hi = (first || mvar > hi) ? mvar : hi;
                        first = false;
                    }
                }
            }
        });
        upd_lo = lo, upd_hi = hi, upd_first = first;
        return -0.5 * (combine4(Ssum) + a.lik_const_f);
    };
    /* the chains with `mine` take the prepared update: their copies swap; beyond minDataSize the precision is inflated
     * from the new variances (noise.py:41-54) */
    auto commit_error_model = [&](bool mine) {
        const uint64_t n_new = n_err + 1;
        const double nm1 = (double)(n_new - 1);
        const bool inflate = n_new > (uint64_t)a.min_data;
        double lo = upd_lo, hi = upd_hi;
        bool first = upd_first;
        {
            double *t0 = mine ? em_o : em, *t1 = mine ? e2_o : e2;
            em_o = mine ? em : em_o, e2_o = mine ? e2 : e2_o;
            em = t0, e2 = t1;
        }
        if (__builtin_amdgcn_ballot_w64(mine && inflate) != 0) {
            /* min / max of the marginal variances over ALL outputs of the chain: the four lanes' ranges joined
             * (a lane without a real output holds none) */
#pragma unroll
            for (int mask = 16; mask <= 32; mask <<= 1) {
                const double olo = shfl_xor_f64(lo, mask), ohi = shfl_xor_f64(hi, mask);
                const int ofirst = __shfl_xor(first ? 1 : 0, mask, 64);
                if (!ofirst) {
                    lo = (first || olo < lo) ? olo : lo;
                    hi = (first || ohi > hi) ? ohi : hi;
                    first = false;
                }
            }
            double scaling = 1.0;
            if (a.use_heuristic) { /* noise.py:41-46 */
                const double minval = lo > 1e-6 ? lo : 1e-6;
                scaling = 2.0 * hi / minval;
                scaling = scaling < 100.0 ? scaling : 100.0;
            }
            if (mine && inflate) {
                /* eight outputs at a time: their M2 entries requested together (one at a time, each waited for its own
                 * trip to L2 -- 32 in a row on the accepting chains' critical path) */
#pragma unroll 1
                for (int q0 = 0; q0 < Q; q0 += 8) {
                    double m2v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) m2v[i] = (q0 + i < Q) ? e2[(q0 + i) * 64] : 0.0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int q = q0 + i, m = 16 * (q >> 2) + 4 * (q & 3) + g;
                        if (q < Q && m < m_real) {
                            const double mvar = m2v[i] / nm1;
                            ep[q * 64] = 1.0 / (scaling * mvar + 1.0 / nprec_c[m]); /* noise.py:51-53, covariance.py:36-41 */
                        }
                    }
                }
                has_prec = true;
            }
        }
        if (mine) n_err = n_new;
    };

    YMC_PROF_DECL;
#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t tbase = (a.t0 + n) * (uint64_t)(S + 1);
        uint64_t z_id = x_id;
        YMC_PROF_START();
        /* surrogate.run(S + 1, state), mlda.py:102-110 */
#pragma unroll 1
        for (int sub = 0; sub < S; ++sub) {
            const uint64_t t = tbase + (uint64_t)sub;
#pragma unroll
            for (int q = 0; q < KS / 2; ++q) {
                double xi0, xi1;
                ymc_draw_normal_pair(a.seed, chain, t, 4u * (uint32_t)q + pair_of_g, &xi0, &xi1);
                swap_odd_even_rows(xi0, xi1);
                y[2 * q] = z[2 * q] + sigma[8 * q + g] * xi0;
                y[2 * q + 1] = z[2 * q + 1] + sigma[8 * q + 4 + g] * xi1;
            }
            const double u = ymc_uniform_cached(uc, a.seed, chain, t);
            const bool moved = any_differs(y, z);
            YMC_PROF(0);
            /* mrw.py:51-57: evaluate_log(proposal), then evaluate_log(state) */
            const uint64_t y_id = t + 1;
            const double ly = query(y_id, y, moved) + logprior(pmean_c, pprec_c, y);
            YMC_PROF(1);
            const double lz = query(z_id, z, moved) + logprior(pmean_c, pprec_c, z);
            YMC_PROF(2);
            const bool acc = moved && ymc_accept(ly - lz, true, ymc_finite(ly), u, st);
#pragma unroll
            for (int s = 0; s < KS; ++s) z[s] = acc ? y[s] : z[s];
            z_id = acc ? y_id : z_id;
            if (a.sub_acc && live && g == 0) a.sub_acc[((uint64_t)n * S + sub) * a.K + k] = acc ? 1 : 0;
            YMC_PROF(3);
        }
        /* fine accept / reject, mlda.py:146-154: tgt(y), finest(x), finest(y), tgt(x); y <- the fine state x */
#pragma unroll
        for (int s = 0; s < KS; ++s) y[s] = wave_live ? xt[s * 64] : 0.0;
        const double u = ymc_uniform_cached(uc, a.seed, chain, tbase + (uint64_t)S);
        const bool moved = any_differs(z, y);
        YMC_PROF(4);
        const double lpfy = fine_likelihood_and_error(z) + logprior(pmean_f, pprec_f, z);
        YMC_PROF(5);
        const double cx = query(x_id, y, moved) + logprior(pmean_c, pprec_c, y);
        const double cy = query(z_id, z, moved) + logprior(pmean_c, pprec_c, z);
        YMC_PROF(6);
        const double log_ratio = ((lpfy + cx) - cy) - lpf;
        const bool acc = wave_live && moved && ymc_accept(log_ratio, true, ymc_finite(lpfy) && ymc_finite(cy), u, st);
        /* aem.py:26-56, before the chain moves */
        if (__builtin_amdgcn_ballot_w64(acc) != 0) commit_error_model(acc);
        YMC_PROF(7);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            z[s] = acc ? z[s] : y[s];
            if (wave_live) xt[s * 64] = z[s];
        }
        lpf = acc ? lpfy : lpf;
        x_id = acc ? z_id : x_id;
        accepted += acc ? 1u : 0u;

        const bool rec = clk.due();
        if (rec && a.traj && wave_live) {
            double *row = a.traj + (((uint64_t)clk.row * nct + tile) * KS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) row[s * 64] = z[s];
        }
        if (live && g == 0) {
            if (a.acc) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
            if (rec && a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + k] = lpf;
        }
        if (rec) clk.wrote();
        YMC_PROF(8);
    }
    YMC_PROF_FLUSH();

    if (wave_live && em != em_run) { /* the run's arrays are the first copy */
#pragma unroll 1
        for (int q = 0; q < Q; ++q) {
            em_run[q * 64] = em[q * 64];
            e2_run[q * 64] = e2[q * 64];
        }
    }
    if (live) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int j = 4 * s + g;
            if (j < d) a.x[k * d + j] = z[s];
        }
        if (g == 0) {
            a.lp[k] = lpf;
            a.acc_count[k] = accepted;
            a.status[k] = st;
            a.err_n[k] = n_err;
            a.aem_flags[2 * k] = has_prec ? 1 : 0;
            a.aem_flags[2 * k + 1] = cache.nkeys;
            a.cache_id[k * 4 + 0] = cache.k0, a.cache_val[k * 3 + 0] = cache.v0;
            a.cache_id[k * 4 + 1] = cache.k1, a.cache_val[k * 3 + 1] = cache.v1;
            a.cache_id[k * 4 + 2] = cache.k2, a.cache_val[k * 3 + 2] = cache.v2;
            a.cache_id[k * 4 + 3] = x_id;
        }
    }
}

/*
 * The same sampler with the chains' ERROR MODEL resident in LDS (round 4).  The kernel above keeps the coarse matrix in
 * LDS and reads a chain's learned state -- mean shift and inflated precision, 2 KB at 128 outputs -- from HBM at every
 * coarse evaluation: 25 x the record's bytes at C3's shape, the state of 65 536 chains being far beyond any cache.
 * Here the roles are swapped: what is PER CHAIN lives in LDS for the whole launch (mean | precision of the workgroup's 64
 * chains: 128 KB), what is THE SAME FOR EVERY CHAIN -- both forward matrices, L2-resident -- streams through the
 * workgroup's shared double buffer (ymc_stream_tile_pairs).  The Welford M2 stays in the run's array: only an ACCEPTED
 * fine proposal touches it, so the update (estimation.py:36-53) is no longer prepared for every chain on every fine step
 * but applied after the decision by the accepting chains, from the model error F_fine - F_coarse the fine pass parked in
 * a scratch (1 KB per chain and fine step, the only per-step traffic besides the record).  The fine pass streams the two
 * matrices interleaved pair by pair (blob_fc: fine tiles 2p, 2p + 1, coarse tiles 2p, 2p + 1), one stream for both maps.
 * The four waves run in lockstep (every stream is a workgroup matter): a cache miss anywhere in the workgroup evaluates
 * the current-model likelihood in all four waves, and only the chains that missed take the value -- per chain the same
 * queries, the same values, the same bits as above.
 */
template <int KS, int DIV>
__global__ __launch_bounds__(kAemThreads) void linear_tile_aem2_lds_kernel(const TileArgs a)
{
    static_assert(KS % DIV == 0, "staged units are whole k-steps");
    ymc_stage_tables();
    extern __shared__ double lds[];
    const int MT = a.MT, m_pad = 16 * MT, d_pad = 4 * KS, Q = 4 * MT; /* both levels have a.m outputs */
    /* constants: coarse bias | coarse prior mean | prior prec | sigma | fine bias | n * noise_prec | row mean | fine prior
     * mean | prior prec | coarse noise precision | coarse data rows [n_data][m_pad] */
    double *bias_c = lds;
    double *pmean_c = bias_c + m_pad;
    double *pprec_c = pmean_c + d_pad;
    double *sigma = pprec_c + d_pad;
    double *bias_f = sigma + d_pad;
    const double *nprec_f = bias_f + m_pad;
    const double *dbar_f = nprec_f + m_pad;
    const double *pmean_f = dbar_f + m_pad;
    const double *pprec_f = pmean_f + d_pad;
    double *nprec_c = bias_f + 3 * m_pad + 2 * d_pad;
    double *inv_nprec_c = nprec_c + m_pad;                    /* 1 / noise precision: the data variance of noise.py:51-53 */
    double *data_c = inv_nprec_c + m_pad;
    const size_t rows_c = (size_t)m_pad * (size_t)a.n_data;
    double *em_all = data_c + rows_c;                         /* [waves][Q][64] mean of the model error */
    double *ep_all = em_all + (size_t)kAemWaves * Q * 64;     /* [waves][Q][64] inflated precision */
    double *stage = ep_all + (size_t)kAemWaves * Q * 64;      /* the streams' double buffer */
    int *vote = reinterpret_cast<int *>(stage + 2 * ymc_stream_unit<KS, DIV>());

    const int tid = threadIdx.x;
    {
        const double *cc = a.blob + (size_t)MT * KS * 64;       /* bias | n prec | row mean | prior mean | prior prec | sigma */
        const double *fc = a.blob_f + (size_t)MT * KS * 64;
        for (int i = tid; i < m_pad; i += kAemThreads) bias_c[i] = cc[i];
        for (int i = tid; i < 3 * d_pad; i += kAemThreads) pmean_c[i] = cc[3 * (size_t)m_pad + i];
        for (int i = tid; i < 3 * m_pad + 2 * d_pad; i += kAemThreads) bias_f[i] = fc[i];
        for (int i = tid; i < m_pad; i += kAemThreads) {
            const double np = a.aem_rows[i];
            nprec_c[i] = np;
            inv_nprec_c[i] = 1.0 / np; /* padding: 1 / 0, never read */
        }
        for (size_t i = tid; i < rows_c; i += kAemThreads) data_c[i] = a.aem_rows[(size_t)m_pad + i];
        if (tid < 3) vote[tid] = 0;
    }

    const int lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const uint64_t nct = (a.K + 15) / 16;
    const uint64_t tile_raw = (uint64_t)blockIdx.x * kAemWaves + wave;
    const bool wave_live = tile_raw < nct;
    const uint64_t tile = wave_live ? tile_raw : 0; /* a wave past the last tile reads tile 0's arrays and writes nothing */
    const uint64_t k = tile_raw * 16 + c;
    const bool live = wave_live && k < a.K;
    const uint64_t kk = live ? k : 0;
    const uint32_t chain = (uint32_t)(a.chain0 + kk);
    const int d = a.d, S = a.S, m_real = a.m;
    const uint32_t pair_of_g = (g == 0) ? 0u : (g == 1) ? 2u : (g == 2) ? 1u : 3u;
    double *xt = a.x_tiled + (tile * KS) * 64 + lane;
    /* this lane's column of the per-output arrays: slot q = 4 mt + r <-> output 16 mt + 4 r + g */
    double *const em = em_all + (size_t)wave * Q * 64 + lane, *const ep = ep_all + (size_t)wave * Q * 64 + lane;
    double *const em_run = a.err_mean_t + (tile * Q) * 64 + lane, *const ep_run = a.prec_t + (tile * Q) * 64 + lane;
    double *const e2 = a.err_m2_t + (tile * Q) * 64 + lane;
    double *const escr = a.err_mean_alt_t + (tile * Q) * 64 + lane; /* F_fine - F_coarse of the current fine proposal */
#pragma unroll 1
    for (int q0 = 0; q0 < Q; q0 += 8) { /* Q is a multiple of 8 (MT is even) */
        double mv[8], pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) mv[i] = em_run[(q0 + i) * 64], pv[i] = ep_run[(q0 + i) * 64];
#pragma unroll
        for (int i = 0; i < 8; ++i) em[(q0 + i) * 64] = mv[i], ep[(q0 + i) * 64] = pv[i];
    }
    __syncthreads();

    double z[KS], y[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int j = 4 * s + g;
        z[s] = (live && j < d) ? a.x[k * d + j] : 0.0;
        if (wave_live) xt[s * 64] = z[s];
    }
    double lpf = live ? a.lp[k] : 0.0;
    uint64_t accepted = live ? a.acc_count[k] : 0;
    uint32_t st = live ? a.status[k] : 0;
    uint64_t n_err = a.err_n[kk];
    bool has_prec = a.aem_flags[2 * kk] != 0;
    AemCache cache;
    cache.nkeys = a.aem_flags[2 * kk + 1];
    cache.k0 = a.cache_id[kk * 4 + 0], cache.v0 = a.cache_val[kk * 3 + 0];
    cache.k1 = a.cache_id[kk * 4 + 1], cache.v1 = a.cache_val[kk * 3 + 1];
    cache.k2 = a.cache_id[kk * 4 + 2], cache.v2 = a.cache_val[kk * 3 + 2];
    uint64_t x_id = a.cache_id[kk * 4 + 3];
    UniformCache uc;
    uc.valid = false;
    uc.pair = 0;
    RecordClock clk(a.rec_stride, a.rec_since);
    uint32_t votes = 0;

    /* does any chain of the WORKGROUP have `p`?  One barrier: call n writes slot n mod 3 and clears slot (n + 1) mod 3,
     * which was last read before everybody's arrival at the previous call's barrier */
    auto wg_any = [&](bool p) {
        const uint32_t cur = votes % 3u, nxt = (votes + 1u) % 3u;
        votes += 1u;
        if (tid == 0) vote[nxt] = 0;
        if (__builtin_amdgcn_ballot_w64(p) != 0 && lane == 0) vote[cur] = 1;
        __syncthreads();
        return vote[cur] != 0;
    };
    /* units of KS/8 k-steps (8 MFMAs) are shorter than a trip to L2: their fragments are requested two units ahead */
    auto stream_pairs = [&](const double *A, int n_pairs, const double (&v)[KS], auto &&emit) {
        if constexpr (DIV >= 8) ymc_stream_tile_pairs_ahead2<KS, DIV, kAemThreads>(A, n_pairs, stage, tid, lane, v, emit);
        else ymc_stream_tile_pairs<KS, DIV, kAemThreads>(A, n_pairs, stage, tid, lane, v, emit);
    };
    /* AEMLikelihood.compute_log_likelihood under the CURRENT error model (likelihood.py:77-84, 128-133), the coarse matrix
     * streamed: residual F - D_i (+ error mean once n >= minDataSize), norm in the (inflated) noise precision, row by row;
     * every row's terms in the canonical four partial sums.  EVERY wave of the workgroup calls this together */
    auto now = [&](const double (&v)[KS]) {
        const bool shift = n_err >= (uint64_t)a.min_data;
        static_assert(kAemRows == 8, "two 4-vectors of row sums");
        const int nr = a.n_data;
        d4 SA = {0.0, 0.0, 0.0, 0.0}, SB = {0.0, 0.0, 0.0, 0.0};
        int row_off[kAemRows];
#pragma unroll
        for (int i = 0; i < kAemRows; ++i) row_off[i] = (i < nr ? i : nr - 1) * m_pad;
        stream_pairs(a.blob, MT / 2, v, [&](int pair, const d4 &acc0, const d4 &acc1) {
            const int mt = 2 * pair;
            /* an output's reads wait for the sums of the one before the previous (the address passes through an empty asm
             * that takes that sum as an input): two outputs' values in registers at most -- unbounded, the compiler hoists
             * all 64 reads of a pair and spills (profiles/r03_tile_experiments.txt, experiment 9) */
            double gate = 0.0, gate_next = 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * (mt + h) + 4 * r + g, q = 4 * (mt + h) + r;
                    const double F = (h ? acc1[r] : acc0[r]) + bias_c[m];
                    int mo = m;
                    asm volatile("" : "+v"(mo) : "v"(gate));
                    const double mean = shift ? em[q * 64] : 0.0;
                    const double pe = has_prec ? ep[q * 64] : nprec_c[mo];
                    double dv[kAemRows];
#pragma unroll
                    for (int i = 0; i < kAemRows; ++i) dv[i] = data_c[row_off[i] + mo];
#pragma unroll
                    for (int i = 0; i < kAemRows; ++i) {
                        const double res = (F - dv[i]) + mean;
                        const double px = pe * res;
                        if (i < 4) SA[i & 3] = SA[i & 3] + res * px;
                        else SB[i & 3] = SB[i & 3] + res * px;
                    }
                    gate = gate_next;
                    gate_next = SA[0];
                }
            }
        });
        double total = 0.0;
#pragma unroll
        for (int i = 0; i < kAemRows; ++i)
            if (i < nr) total = total + combine4(i < 4 ? SA[i & 3] : SB[i & 3]);
        return -0.5 * total;
    };
    auto logprior = [&](const double *pmean, const double *pprec, const double (&v)[KS]) {
        return ymc_tile_logprior<KS>(pmean, pprec, g, v);
    };
    auto any_differs = [&](const double (&p)[KS], const double (&q)[KS]) { return ymc_tile_any_differs<KS>(p, q); };
    auto query = [&](uint64_t id, const double (&v)[KS], bool active) {
        const int hit = cache.find(id);
        const bool miss = active && hit < 0;
        double fresh = 0.0;
        if (wg_any(miss)) fresh = now(v);
        return active ? cache.take(id, hit, fresh) : 0.0;
    };
    /* the fine likelihood of v (for the decision) and the model error F_fine(v) - F_coarse(v) (aem.py:44-56), parked for
     * the chains that will accept: ONE stream of both matrices */
    auto fine_likelihood_and_error = [&](const double (&v)[KS]) {
        double Ssum = 0.0;
        d4 f0 = {0.0, 0.0, 0.0, 0.0}, f1 = {0.0, 0.0, 0.0, 0.0};
        stream_pairs(a.blob_fc, MT, v, [&](int P, const d4 &acc0, const d4 &acc1) {
            if ((P & 1) == 0) {
                f0 = acc0, f1 = acc1;
                ymc_tile_add_residuals(P, acc0, acc1, bias_f, nprec_f, dbar_f, g, Ssum);
            } else {
                const int mt = P - 1;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = 16 * (mt + h) + 4 * r + g, q = 4 * (mt + h) + r;
                        const double Ff = (h ? f1[r] : f0[r]) + bias_f[m];
                        const double Fc = (h ? acc1[r] : acc0[r]) + bias_c[m];
                        if (wave_live) escr[q * 64] = Ff - Fc;
                    }
                }
            }
        });
        return -0.5 * (combine4(Ssum) + a.lik_const_f);
    };
    /* update_error_estimate for the chains with `mine` (likelihood.py:142-150, estimation.py:36-53): Welford step on every
     * output from the parked model error, then -- beyond minDataSize -- the precision inflated from the new variances
     * (noise.py:41-54) */
    auto commit_error_model = [&](bool mine) {
        const uint64_t n_new = n_err + 1;
        const double dn = (double)n_new, nm1 = (double)(n_new - 1);
        const bool inflate = n_new > (uint64_t)a.min_data;
        double lo = 0.0, hi = 0.0;
        bool first = true;
        if (mine) {
#pragma unroll 1
            for (int q0 = 0; q0 < Q; q0 += 8) {
                double ev[8], m2v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) ev[i] = escr[(q0 + i) * 64], m2v[i] = e2[(q0 + i) * 64];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = q0 + i, m = 16 * (q >> 2) + 4 * (q & 3) + g;
                    double mean = em[q * 64], m2 = m2v[i];
                    const double delta = ev[i] - mean; /* estimation.py:36-53 */
                    mean = mean + delta / dn;
                    const double delta2 = ev[i] - mean;
                    m2 = m2 + delta * delta2;
                    if (m < m_real) {
                        em[q * 64] = mean;
                        e2[q * 64] = m2;
                        m2v[i] = m2;
                        const double mvar = m2 / nm1;
                        lo = (first || mvar < lo) ? mvar : lo;
                        hi = (first || mvar > hi) ? mvar : hi;
                        first = false;
                    }
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(mine && inflate) != 0) {
            /* min / max of the marginal variances over ALL outputs of the chain: the four lanes' ranges joined
             * (a lane without a real output holds none) */
#pragma unroll
            for (int mask = 16; mask <= 32; mask <<= 1) {
                const double olo = shfl_xor_f64(lo, mask), ohi = shfl_xor_f64(hi, mask);
                const int ofirst = __shfl_xor(first ? 1 : 0, mask, 64);
                if (!ofirst) {
                    lo = (first || olo < lo) ? olo : lo;
                    hi = (first || ohi > hi) ? ohi : hi;
                    first = false;
                }
            }
            double scaling = 1.0;
            if (a.use_heuristic) { /* noise.py:41-46 */
                const double minval = lo > 1e-6 ? lo : 1e-6;
                scaling = 2.0 * hi / minval;
                scaling = scaling < 100.0 ? scaling : 100.0;
            }
            if (mine && inflate) {
#pragma unroll 1
                for (int q0 = 0; q0 < Q; q0 += 8) {
                    double m2v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) m2v[i] = e2[(q0 + i) * 64];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int q = q0 + i, m = 16 * (q >> 2) + 4 * (q & 3) + g;
                        if (m < m_real) {
                            const double mvar = m2v[i] / nm1;
                            ep[q * 64] = 1.0 / (scaling * mvar + inv_nprec_c[m]); /* noise.py:51-53, covariance.py:36-41 */
                        }
                    }
                }
                has_prec = true;
            }
        }
        if (mine) n_err = n_new;
    };

    YMC_PROF_DECL;
#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t tbase = (a.t0 + n) * (uint64_t)(S + 1);
        uint64_t z_id = x_id;
        YMC_PROF_START();
        /* surrogate.run(S + 1, state), mlda.py:102-110 */
#pragma unroll 1
        for (int sub = 0; sub < S; ++sub) {
            const uint64_t t = tbase + (uint64_t)sub;
#pragma unroll
            for (int q = 0; q < KS / 2; ++q) {
                double xi0, xi1;
                ymc_draw_normal_pair(a.seed, chain, t, 4u * (uint32_t)q + pair_of_g, &xi0, &xi1);
                swap_odd_even_rows(xi0, xi1);
                y[2 * q] = z[2 * q] + sigma[8 * q + g] * xi0;
                y[2 * q + 1] = z[2 * q + 1] + sigma[8 * q + 4 + g] * xi1;
            }
            const double u = ymc_uniform_cached(uc, a.seed, chain, t);
            const bool moved = wave_live && any_differs(y, z);
            YMC_PROF(0);
            /* mrw.py:51-57: evaluate_log(proposal), then evaluate_log(state) */
            const uint64_t y_id = t + 1;
            const double ly = query(y_id, y, moved) + logprior(pmean_c, pprec_c, y);
            YMC_PROF(1);
            const double lz = query(z_id, z, moved) + logprior(pmean_c, pprec_c, z);
            YMC_PROF(2);
            const bool acc = moved && ymc_accept(ly - lz, true, ymc_finite(ly), u, st);
#pragma unroll
            for (int s = 0; s < KS; ++s) z[s] = acc ? y[s] : z[s];
            z_id = acc ? y_id : z_id;
            if (a.sub_acc && live && g == 0) a.sub_acc[((uint64_t)n * S + sub) * a.K + k] = acc ? 1 : 0;
            YMC_PROF(3);
        }
        /* fine accept / reject, mlda.py:146-154: tgt(y), finest(x), finest(y), tgt(x); y <- the fine state x */
#pragma unroll
        for (int s = 0; s < KS; ++s) y[s] = wave_live ? xt[s * 64] : 0.0;
        const double u = ymc_uniform_cached(uc, a.seed, chain, tbase + (uint64_t)S);
        const bool moved = wave_live && any_differs(z, y);
        YMC_PROF(4);
        const double lpfy = fine_likelihood_and_error(z) + logprior(pmean_f, pprec_f, z);
        YMC_PROF(5);
        const double cx = query(x_id, y, moved) + logprior(pmean_c, pprec_c, y);
        const double cy = query(z_id, z, moved) + logprior(pmean_c, pprec_c, z);
        YMC_PROF(6);
        const double log_ratio = ((lpfy + cx) - cy) - lpf;
        const bool acc = wave_live && moved && ymc_accept(log_ratio, true, ymc_finite(lpfy) && ymc_finite(cy), u, st);
        /* aem.py:26-56, before the chain moves */
        if (__builtin_amdgcn_ballot_w64(acc) != 0) commit_error_model(acc);
        YMC_PROF(7);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            z[s] = acc ? z[s] : y[s];
            if (wave_live) xt[s * 64] = z[s];
        }
        lpf = acc ? lpfy : lpf;
        x_id = acc ? z_id : x_id;
        accepted += acc ? 1u : 0u;

        const bool rec = clk.due();
        if (rec && a.traj && wave_live) {
            double *row = a.traj + (((uint64_t)clk.row * nct + tile) * KS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) row[s * 64] = z[s];
        }
        if (live && g == 0) {
            if (a.acc) a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
            if (rec && a.lp_out) a.lp_out[(uint64_t)clk.row * a.K + k] = lpf;
        }
        if (rec) clk.wrote();
        YMC_PROF(8);
    }
    YMC_PROF_FLUSH();

    if (wave_live) { /* the learned state goes back to the run's arrays */
#pragma unroll 1
        for (int q0 = 0; q0 < Q; q0 += 8) {
            double mv[8], pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) mv[i] = em[(q0 + i) * 64], pv[i] = ep[(q0 + i) * 64];
#pragma unroll
            for (int i = 0; i < 8; ++i) em_run[(q0 + i) * 64] = mv[i], ep_run[(q0 + i) * 64] = pv[i];
        }
    }
    if (live) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int j = 4 * s + g;
            if (j < d) a.x[k * d + j] = z[s];
        }
        if (g == 0) {
            a.lp[k] = lpf;
            a.acc_count[k] = accepted;
            a.status[k] = st;
            a.err_n[k] = n_err;
            a.aem_flags[2 * k] = has_prec ? 1 : 0;
            a.aem_flags[2 * k + 1] = cache.nkeys;
            a.cache_id[k * 4 + 0] = cache.k0, a.cache_val[k * 3 + 0] = cache.v0;
            a.cache_id[k * 4 + 1] = cache.k1, a.cache_val[k * 3 + 1] = cache.v1;
            a.cache_id[k * 4 + 2] = cache.k2, a.cache_val[k * 3 + 2] = cache.v2;
            a.cache_id[k * 4 + 3] = x_id;
        }
    }
}

/* bytes of LDS the LDS-resident-error-model build wants with units of KS / div k-steps */
size_t aem2_lds_resident_bytes(int KS, int MT, int n_data, int div)
{
    const size_t m_pad = 16 * (size_t)MT, d_pad = 4 * (size_t)KS, Q = 4 * (size_t)MT;
    const size_t consts = 6 * m_pad + 5 * d_pad + m_pad * (size_t)n_data;
    const size_t unit = 2 * (size_t)(KS / div) * 64;
    return (consts + 2 * (size_t)kAemWaves * Q * 64 + 2 * unit) * sizeof(double) + 16;
}

template <int KS, int DIV>
hipError_t launch_lds_model(const TileArgs &a, size_t lds, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_tile_aem2_lds_kernel<KS, DIV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t nct = (a.K + 15) / 16;
    const unsigned grid = (unsigned)((nct + kAemWaves - 1) / kAemWaves);
    hipLaunchKernelGGL((linear_tile_aem2_lds_kernel<KS, DIV>), dim3(grid), dim3(kAemThreads), lds, s, a);
    return hipGetLastError();
}

template <int KS, int STG, bool ROWBLOCKS = false>
hipError_t launch_staged(const TileArgs &a, size_t lds, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_tile_aem2_kernel<KS, STG, ROWBLOCKS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint64_t nct = (a.K + 15) / 16;
    const unsigned grid = (unsigned)((nct + kAemWaves - 1) / kAemWaves);
    hipLaunchKernelGGL((linear_tile_aem2_kernel<KS, STG, ROWBLOCKS>), dim3(grid), dim3(kAemThreads), lds, s, a);
    return hipGetLastError();
}

template <int KS>
hipError_t launch_as(const TileArgs &a, hipStream_t s)
{
    /* the fine matrix through the shared double buffer in the largest unit the LDS has room for (identical
     * results; YMC_TILE_STAGE=0 switches it off, the tests run both) */
    const size_t plain = ymc_tile_aem2_lds_bytes(KS, a.MT, a.n_data);
    if (a.n_data > kAemRows) return launch_staged<KS, 0, true>(a, plain, s);
    /* YMC_TILE_AEM=lds: the chains' error model in LDS and both matrices streamed (identical results, the tests run both;
     * 1/8 of the HBM traffic but 9 % slower at C3's shape -- profiles/r04_c3a_builds.txt -- so the coarse matrix in LDS
     * stays the default) */
    bool lds_model = false;
    if (const char *env = std::getenv("YMC_TILE_AEM")) lds_model = a.blob_fc != nullptr && env[0] == 'l';
    if (lds_model) {
        /* the largest stream unit the LDS has room for (at d = m = 128: KS/8 k-steps; with the fine level's constants
         * read from global memory instead KS/4 would fit -- measured: the fine pass then waits for them, 15.6 against
         * 13.0 ms per launch of c3a) */
        if (aem2_lds_resident_bytes(KS, a.MT, a.n_data, 4) <= kTileLdsLimit)
            return launch_lds_model<KS, 4>(a, aem2_lds_resident_bytes(KS, a.MT, a.n_data, 4), s);
        if constexpr (KS >= 8)
            if (aem2_lds_resident_bytes(KS, a.MT, a.n_data, 8) <= kTileLdsLimit)
                return launch_lds_model<KS, 8>(a, aem2_lds_resident_bytes(KS, a.MT, a.n_data, 8), s);
    }
    const size_t unit4 = 2 * (size_t)ymc_stream_unit<KS, 4>() * sizeof(double);
    bool stage = true;
    if (const char *env = std::getenv("YMC_TILE_STAGE")) stage = env[0] != '0';
    if (stage && plain + unit4 <= kTileLdsLimit) return launch_staged<KS, 4>(a, plain + unit4, s);
    if constexpr (KS >= 8)
        if (stage && plain + unit4 / 2 <= kTileLdsLimit) return launch_staged<KS, 8>(a, plain + unit4 / 2, s);
    return launch_staged<KS, 0>(a, plain, s);
}

}  // namespace

size_t ymc_tile_aem2_lds_bytes(int KS, int MT, int n_data)
{
    return ymc_tile_mlda2_lds_bytes(KS, MT, MT) + 16 * (size_t)MT * (1 + (size_t)n_data) * sizeof(double);
}

int ymc_tile_aem2_max_rows(void) { return kAemRows; } /* rows per pass over the forward map (more rows: more passes) */

#ifdef YMC_TILE_PROFILE
extern "C" int ymc_debug_tile_aem_sections(unsigned long long *out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(ymc_tile_prof), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -2;
}
#endif

hipError_t ymc_launch_linear_tile_aem2(int KS, const TileArgs &a, hipStream_t s)
{
    switch (KS) {
    case 4: return launch_as<4>(a, s);
    case 8: return launch_as<8>(a, s);
    case 16: return launch_as<16>(a, s);
    case 32: return launch_as<32>(a, s);
    default: return hipErrorInvalidValue;
    }
}
