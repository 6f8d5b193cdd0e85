/* ymc_tile.h -- launch interface of the large-dimension linear-posterior
 * kernels (ymc_tile.hip). */
#ifndef YMC_TILE_H
#define YMC_TILE_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ymc.h"

constexpr int kTileWaves = 8;                 /* waves per workgroup (one per SIMD), 16 chains each */
constexpr int kTileThreads = 64 * kTileWaves;
constexpr size_t kTileLdsLimit = 156 * 1024;  /* LDS per CU on gfx950 (160 KB) less the 4 KB of static draw tables */

/* one posterior of a level above the base of the nested sampler (ymc_tile_nested.hip): a block in `blob`'s
 * layout in global memory */
struct TileLevel {
    const double *blob;
    double lik_const;
    int32_t MT, pad_;
};

struct TileArgs {
    /* one device block, staged into LDS by every workgroup:
     * A fragments [MT][KS][64] | bias[m_pad] | n * noise_prec[m_pad] | row mean of the data[m_pad] |
     * prior_mean[d_pad] | prior_prec[d_pad] | sigma[d_pad]   (padding = zeros) */
    const double *blob;
    double lik_const;      /* sum_i |D_i - Dbar|^2 (ymc_linear_data_summary) */
    int32_t d, m, pad0_, MT;
    uint64_t seed, chain0, K, t0;
    uint32_t nsteps, flags;
    int32_t pcn, pad_;      /* 1: preconditioned Crank-Nicolson (pcn.py:9-57): sigma is the PRIOR's, lp the log-likelihood */
    uint32_t rec_stride, rec_since; /* thinned record (RecordClock) */
    double pcn_c1, pcn_c2;  /* sqrt(1 - 2h), sqrt(2h) */
    double *x, *lp;
    uint64_t *acc_count;
    uint32_t *status;
    double *traj; /* tiled layout [rows][ceil(K/16)][KS][64] */
    uint8_t *acc;
    double *lp_out;
    /* two-level delayed acceptance (mlda.py:100-110,146-154): `blob` is the COARSE level (staged in LDS, it is
     * evaluated S times per fine step), blob_f the FINE level in the same layout (its A fragments are read
     * from global memory / L2, its constants are staged) */
    const double *blob_f;
    double lik_const_f;
    int32_t m_f, MT_f, S, pad1_;
    double *lpc;      /* coarse log-density of the carried (fine) state */
    double *x_tiled;  /* the carried state in the record's tiled layout [ceil(K/16)][KS][64] (scratch) */
    uint8_t *sub_acc; /* [rows][S][K] or null */
    /* random walk with a DENSE proposal covariance (covariance.py:69-94): its lower Cholesky factor as MFMA
     * A-operand fragments [MT_L][KS][64] in global memory (rows = components), and a scratch in the record's
     * tiled layout for L xi; null = the diagonal proposal of sigma */
    const double *chol_frag;
    double *w_tiled;
    int32_t MT_L, pad3_;
    /* adaptive error model (ymc_tile_aem.hip; both levels have m outputs): the coarse likelihood's state per
     * chain.  Per-output arrays are laid out like the record, [ceil(K/16)][4 MT][64]: slot 4 mt + r of lane
     * (g, c) belongs to output 16 mt + 4 r + g of chain c */
    int32_t n_data, min_data, use_heuristic, pad2_;
    const double *aem_rows;  /* noise precision [m_pad] | data rows [n_data][m_pad] of the coarse level */
    uint64_t *err_n;         /* [K] accepted fine proposals */
    double *err_mean_t, *err_m2_t, *prec_t;
    double *err_mean_alt_t, *err_m2_alt_t; /* the second copy of mean / M2 (the kernel updates into the copy a chain is not reading;
                                            * the LDS-resident build parks the model error of a fine proposal in the first) */
    const double *blob_fc;   /* both levels' A fragments interleaved pair by pair: fine tiles 2p, 2p + 1, coarse tiles 2p, 2p + 1
                              * ([2 MT][KS][64]; one stream for the fine pass), or null */
    int32_t *aem_flags;      /* [K][2] {precision set, cache fill} */
    uint64_t *cache_id;      /* [K][4]: the three cached points' ids (oldest first) and the carried state's id */
    double *cache_val;       /* [K][3] */
    /* nested delayed acceptance (ymc_tile_nested.hip, YMC_METHOD_MLDA_NESTED): `blob` is level 0's posterior,
     * levels[2 l] / levels[2 l + 1] are P / Q of level l >= 1 (device array), lp / lpc the top level's P / Q of
     * the carried state, x_tiled holds E - 1 parked states */
    const TileLevel *levels;
    int32_t E, steps[YMC_MAX_LEVELS], pad4_;
    uint64_t cost_top;       /* transitions one top-level step spans */
    uint64_t *level_acc;     /* [E-1][K] accepted transitions of the lower levels */
};

struct BigLinearArgs {
    const double *A, *b, *dbar, *nprec, *prior_mean, *prior_prec; /* unpadded, row-major */
    double lik_const;
    int32_t d, m, pad0_, likelihood_only;
};

/* Linear posteriors beyond the matrix-core kernels' 256 parameters (random walk with a diagonal proposal, pCN): one
 * transition as a sequence of plain kernels over [K][d] arrays -- draws (ymc_launch_rng_fill), proposal, the canonical-order
 * evaluation (ymc_launch_linear_big_eval), decision, move + record + moments.  Any d, any number of outputs; the record is
 * in the lane kernels' plain layout. */
struct WideArgs {
    uint64_t K;
    int32_t d, pcn;
    double pcn_c1, pcn_c2;
    const double *sigma, *pmean;   /* [d] */
    const double *xi, *u;          /* the transition's draws: [K][d], [K] (dense proposal: xi holds L xi, sigma ones) */
    double *x, *y;                 /* [K][d]: the recorded chain's state, the proposal */
    /* what one decision works on (random walk: from = dst = x, cur_lp = lp, count = acc_count; a sub-step of the two-level
     * sampler: from = dst = z, cur_lp = the coarse log-density of z, no count) */
    const double *from;            /* the state the proposal starts from and is compared with */
    double *cur_lp;                /* [K]: its log-density, replaced on acceptance */
    double *dst, *dst_tiled;       /* the state that moves (and its copy in the matrix-core layout, or null) */
    uint64_t *count;               /* accepted moves (or null) */
    /* two-level delayed acceptance: the sub-chain's state z (from / dst above), the coarse log-densities of x and z, the
     * fine log-density of z */
    const double *z, *lpfz;
    double *lpc, *lpcz;
    double *y_tiled;               /* the proposal once more in the matrix-core layout [ceil(K/16)][KS][64] (padding zero) */
    int32_t KS, pad_;
    double *lp;                    /* [K] */
    const double *lpy;             /* [K]: log-density of the proposals */
    uint64_t *acc_count;
    uint32_t *status;
    uint8_t *accflag;              /* [K]: this transition's decisions */
    uint32_t *differs;             /* [K]: set (by plain stores of 1) where the proposal differs from the state it is compared
                                      with; the deciding kernel reads and clears it */
    uint8_t *acc_row;              /* record rows of this transition (null: not recorded / not due) */
    double *traj_row, *lp_row;
    double *wf_mean, *wf_m2;       /* [K][d] (null: no moments) */
    double wf_dn, wf_rdn;          /* the moments' sample count after this transition and RN(1 / count) */
};
/* The random walk (diagonal proposal) and pCN beyond 256 parameters as ONE persistent launch per advance call
 * (linear_wide_fused_kernel, round 4): a workgroup owns NT tiles of 16 chains for the whole launch; the proposal of
 * the moment lives in LDS in the matrix cores' B-operand layout, the state in registers, a slice of the k-steps per
 * wave; every wave multiplies ITS output tiles over all k (the matrix's fragments from L2, each feeding NT MFMAs), and
 * the residual terms join the chains' sums in the canonical order by a hand-over from wave to wave. */
struct WideFusedArgs {
    const double *blob;            /* TileArgs::blob layout with KS k-steps and MT output tiles, in global memory */
    double lik_const;
    int32_t KS, MT, d, likelihood_only, pcn, pad_;
    double pcn_c1, pcn_c2;
    uint64_t seed, chain0, K, t0;
    uint32_t nsteps, rec_stride, rec_since, pad1_;
    double *x, *lp;                /* [K][d], [K]: carried */
    uint64_t *acc_count;
    uint32_t *status;
    double *traj;                  /* plain record rows of this launch [rows][K][d], or null */
    uint8_t *acc;                  /* [nsteps][K] or null */
    double *lp_out;                /* [rows][K] or null */
    const double *chol_frag;       /* dense proposal: the factor as A-operand fragments [MT_L][KS][64], else null */
    int32_t MT_L, pad2_;
};
/* Two-level delayed acceptance (mlda.py:100-110,146-154) beyond 256 parameters the same way (linear_wide_fused2_kernel):
 * the sub-chain's state z as a slice of registers per wave (the fine state x parked in x_park meanwhile), ONE tile of 16 chains per workgroup;
 * the S coarse transitions and the fine decision each one evaluation of the tile in LDS (the fine one after z has been
 * written into it). */
struct WideFused2Args {
    const double *blob_c, *blob_f; /* coarse / fine posterior, TileArgs::blob layout, same KS */
    double lik_const_c, lik_const_f;
    int32_t KS, MT_c, MT_f, d, S, pad_;
    uint64_t seed, chain0, K, t0;  /* t0: fine steps done */
    uint32_t nsteps, rec_stride, rec_since, pad1_;
    double *x, *lp, *lpc;          /* [K][d]; fine and coarse log-density of the carried state */
    double *x_park;                /* [tiles][KS][64] scratch: the fine state during a fine step's sub-chain (tiled layout) */
    uint64_t *acc_count;
    uint32_t *status;
    double *traj;
    uint8_t *acc, *sub_acc;        /* [nsteps][K], [nsteps][S][K] or null */
    double *lp_out;
};
size_t ymc_wide_fused2_lds_bytes(int KS, int MT_c, int MT_f, uint64_t K, int cus);
hipError_t ymc_launch_wide_fused2(const WideFused2Args &a, int cus, hipStream_t s);
/* Small ensembles (fewer tiles than half the CUs): a tile of 16 chains over a GROUP of workgroups, each multiplying its share of
 * the output tiles, the per-output terms exchanged through global memory once per transition (linear_wide_split_kernel). */
struct WideSplitArgs {
    WideFusedArgs f;
    int32_t G, chunk;              /* members of a group; output tiles per round of the leader's sum: set by the launch function */
    double *xch;                   /* [tiles][2][MT][256]: the exchange slots, ymc_wide_split_exchange_bytes */
    uint32_t *arrived;             /* [tiles][64]: word 0 = members through with their stores, words 32-33 = {transition, sixteen accept
                                    * flags} from the group's leader (the systolic build); zeroed by the launch function */
    uint32_t poll_limit, pad1_;
    uint32_t *fault;               /* the run's count of abandoned hand-overs (page-locked) */
};
/* the group's size for this shape and ensemble; 0 = the fused kernel (or the launch-per-stage path) runs it */
int ymc_wide_split_group(int KS, int MT, uint64_t K, int cus);
size_t ymc_wide_split_exchange_bytes(int MT, uint64_t K);
size_t ymc_wide_split_counter_bytes(uint64_t K);
hipError_t ymc_launch_wide_split(WideSplitArgs a, int cus, hipStream_t s);
/* 0 = this shape / ensemble does not take the fused kernel (the caller runs the launch-per-stage path) */
size_t ymc_wide_fused_lds_bytes(int KS, int MT, uint64_t K, int cus);
hipError_t ymc_launch_wide_fused(const WideFusedArgs &a, int cus, hipStream_t s);
/* the random walk with a dense proposal covariance as one persistent launch (0 = does not fit: launches per stage) */
size_t ymc_wide_fused_dense_lds_bytes(int KS, int MT, int MT_L);
hipError_t ymc_launch_wide_fused_dense(const WideFusedArgs &a, hipStream_t s);
hipError_t ymc_launch_wide_propose(const WideArgs &a, hipStream_t s);
/* the normals of transition t in the matrix-core layout (xi_tiled) and the accept uniforms (a.u): the dense proposal's input */
hipError_t ymc_launch_wide_draws(const WideArgs &a, uint64_t seed, uint64_t chain0, uint64_t t, double *xi_tiled, hipStream_t s);
/* the draws of transition t AND the proposal in one launch (a thread per chain and pair of components; a.u is written) */
hipError_t ymc_launch_wide_draw_propose(const WideArgs &a, uint64_t seed, uint64_t chain0, uint64_t t, hipStream_t s);
/* log-densities of the K points of y_tiled on the matrix cores: `blob` as TileArgs::blob with KS k-steps (any number) and MT
 * output tiles, everything read from global memory */
hipError_t ymc_launch_wide_eval(const double *blob, int KS, int MT, double lik_const, int likelihood_only, const double *y_tiled,
                                uint64_t K, double *lp_out, hipStream_t s, double *terms = nullptr);
/* scratch doubles the evaluation of a FEW points wants (a wave per tile AND pair of output tiles, the residual terms summed
 * in order by a second kernel); 0 for ensembles that fill the card with a wave per tile */
size_t ymc_wide_eval_terms(int MT, uint64_t K);
/* out[K][d] = L in_tiled: the rows of a d x d factor (MFMA fragments [MT_L][KS][64]) times the 16 points of a tile, row i one
 * k-ordered fma chain (the dense proposal's L xi, as linear_tile_kernel) */
hipError_t ymc_launch_wide_rows(const double *frag, int KS, int MT_L, int d, const double *in_tiled, uint64_t K, double *out,
                                hipStream_t s);
hipError_t ymc_launch_wide_decide(const WideArgs &a, hipStream_t s);
hipError_t ymc_launch_wide_apply(const WideArgs &a, hipStream_t s);
hipError_t ymc_launch_wide_copy(const WideArgs &a, const double *src, double *dst, double *dst_tiled, hipStream_t s);
hipError_t ymc_launch_wide_differs(const WideArgs &a, const double *p, const double *q, hipStream_t s); /* differs[k] = 1 where p != q */
hipError_t ymc_launch_wide_fine_decide(const WideArgs &a, hipStream_t s); /* mlda.py:146-154: z against x */
hipError_t ymc_launch_wide_fine_apply(const WideArgs &a, hipStream_t s);  /* x <- z where accepted; record, moments */

size_t ymc_tile_lds_bytes(int KS, int MT);
size_t ymc_tile_stream_lds_bytes(int KS, int MT); /* the forward matrix streamed from L2: constants + double buffer */
size_t ymc_tile_mlda2_lds_bytes(int KS, int MT_coarse, int MT_fine);
size_t ymc_tile_mlda2_stream_lds_bytes(int KS, int MT_coarse, int MT_fine); /* both matrices streamed: constants + double buffer */
hipError_t ymc_launch_linear_tile(int KS, const TileArgs &a, hipStream_t s);
hipError_t ymc_launch_linear_tile_mlda2(int KS, const TileArgs &a, hipStream_t s);
size_t ymc_tile_aem2_lds_bytes(int KS, int MT, int n_data);
int ymc_tile_aem2_max_rows(void);
hipError_t ymc_launch_linear_tile_aem2(int KS, const TileArgs &a, hipStream_t s);
size_t ymc_tile_nested_stream_lds_bytes(int KS, int MT); /* level 0's matrix streamed as the upper ones */
hipError_t ymc_launch_linear_tile_nested(int KS, const TileArgs &a, hipStream_t s);
hipError_t ymc_launch_untile(const double *tiled, uint64_t row0, uint64_t rows, uint64_t K, int d, int KS,
                             double *out, hipStream_t s);
hipError_t ymc_launch_welford_replay(const double *tiled, const double *x_before, uint64_t rows, uint64_t K, int d,
                                     int KS, uint64_t wf_n0, double *wf_mean, double *wf_m2, hipStream_t s);
hipError_t ymc_launch_linear_big_eval(const BigLinearArgs &p, uint64_t n, const double *x, double *lp, double *fwd,
                                      hipStream_t s);

#endif /* YMC_TILE_H */
