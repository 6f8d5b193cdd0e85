/*
 * ymc_numerics.h -- the normative numeric specification of the chain-ensemble
 * hot path: counter-based random draws and the fp64 elementary functions the
 * Metropolis-Hastings kernels use.
 *
 * Everything in this header is written with IEEE-754 binary64 basic operations
 * only (+, -, *, /, sqrt, fma -- each correctly rounded), integer arithmetic
 * and two small constant tables (log, sincospi).  (On the device the one square
 * root is formed from v_rsq_f64 by Goldschmidt's corrections, which deliver the
 * correctly rounded value: ymc_sqrt_radius, checked against the host's sqrt.)  The same source is compiled for gfx950 by hipcc (device code of
 * the HIP kernels) and for x86-64 by gcc (oracle/, the CPU checker).  Both are
 * built with -ffp-contract=off, and every fused multiply-add below is an
 * explicit ymc_fma(), so the two produce IDENTICAL bits.  That is what makes
 * accept masks and trajectories comparable bit-for-bit between the GPU and the
 * CPU restatement ("Bit-exact masks require bit-exact
 * states" / "Whose normals?").
 *
 * The reference (yagremcmc) draws from NumPy's global legacy generator:
 *   xi = standard_normal(d)            yagremcmc/statistics/gaussian.py:63
 *   u  = uniform(0, 1, size=1)[0]      yagremcmc/chain/metropolisHastings.py:68
 * A Mersenne-Twister stream cannot be sharded over 10^5..10^6 chains, so the
 * ensemble replaces it by a pure function  (seed, chain, transition) -> draws
 * defined here; parity with the reference is established on identical injected
 * draws (tests/golden/make_golden.py replays this stream into the reference).
 */
#ifndef YMC_NUMERICS_H
#define YMC_NUMERICS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define YMC_FN __host__ __device__ static inline __attribute__((always_inline))
#else
#define YMC_FN static inline __attribute__((always_inline))
#endif

/* ------------------------------------------------------------------------ */
/* bit helpers                                                              */
/* ------------------------------------------------------------------------ */

YMC_FN double ymc_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

/* fma whose ADDEND is a polynomial coefficient (a Horner step).  Same value as
 * ymc_fma; on the device it is pinned to the three-address v_fma_f64, because
 * hipcc (ROCm 7.2) otherwise selects the two-address v_fmac_f64 and pays a
 * v_mov_b64 of the coefficient per step.  The coefficient sits in a VGPR pair
 * by default; a translation unit whose kernels loop over transitions defines
 * YMC_COEFF_HOISTED to keep every coefficient in VGPRs across the loop, one
 * that is short of VGPRs defines YMC_COEFF_IN_SGPR and gets it from an SGPR
 * pair instead.  All three give the same bits. */
YMC_FN double ymc_fma_c(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(YMC_COEFF_IN_SGPR)
    double d;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
#elif defined(__HIP_DEVICE_COMPILE__) && defined(YMC_COEFF_HOISTED)
    /* v_div_fixup_f64(c, +-1, 1) returns c for a finite c (its sign is taken from the "quotient"
     * of the two other operands, hence the -1 for a negative c): an opaque, speculatable,
     * non-convergent copy of the coefficient into a VGPR pair, which LICM lifts out of the
     * transition loop.  (An asm statement is convergent in HIP and stays in the loop; a plain
     * literal is re-materialised there with two s_mov_b32 and a v_mov_b64 per use.) */
    const double cv = __builtin_amdgcn_div_fixup(c, c < 0.0 ? -1.0 : 1.0, 1.0);
    double d;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(cv));
    return d;
#elif defined(__HIP_DEVICE_COMPILE__)
    double d;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return __builtin_fma(a, b, c);
#endif
}

/* A polynomial's LEADING coefficient (the first multiplicand of a Horner chain): the value c;
 * in a YMC_COEFF_HOISTED unit it is kept in a VGPR pair across the loop like the addends above. */
YMC_FN double ymc_coeff(double c)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(YMC_COEFF_HOISTED)
    return __builtin_amdgcn_div_fixup(c, c < 0.0 ? -1.0 : 1.0, 1.0);
#else
    return c;
#endif
}

YMC_FN uint64_t ymc_bits(double x)
{
    uint64_t u;
    __builtin_memcpy(&u, &x, sizeof u);
    return u;
}

YMC_FN double ymc_from_bits(uint64_t u)
{
    double x;
    __builtin_memcpy(&x, &u, sizeof x);
    return x;
}

/* ------------------------------------------------------------------------ */
/* a / b given y = RN(1/b): Markstein's correction, twice.  q0 = RN(a y) is   */
/* within 1.5 ulp of a/b; the first step makes it faithful; for a faithful q  */
/* and a correctly rounded reciprocal  RN(q + (a - b q) y)  IS  RN(a/b)       */
/* (P. Markstein, IBM J. R&D 34(1), 1990; the final step of the IA-64 divide).*/
/* Used where many lanes divide by the same b (Welford: b = sample count):    */
/* 5 fma-class operations instead of the ~14 of an IEEE division sequence.    */
/* Requires a, b, a/b in the normal range; tests/test_numerics.py checks it   */
/* against the hardware division on random and near-midpoint quotients.       */
/* ------------------------------------------------------------------------ */
YMC_FN double ymc_div_by_recip(double a, double b, double y)
{
    double q = a * y;
    double r = ymc_fma(-q, b, a);
    q = ymc_fma(r, y, q);
    r = ymc_fma(-q, b, a);
    return ymc_fma(r, y, q);
}

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as   */
/* easy as 1, 2, 3", SC'11).  Pinned by the Random123 known-answer vectors  */
/* in tests/test_numerics.py.                                               */
/* ------------------------------------------------------------------------ */

#define YMC_PHILOX_M0 0xD2511F53u
#define YMC_PHILOX_M1 0xCD9E8D57u
#define YMC_PHILOX_W0 0x9E3779B9u
#define YMC_PHILOX_W1 0xBB67AE85u

typedef struct ymc_u32x4 {
    uint32_t v[4];
} ymc_u32x4;

/* a ^ b ^ c: one v_bitop3_b32 on gfx950 (the compiler pairs the xors into two instructions) */
YMC_FN uint32_t ymc_xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

YMC_FN ymc_u32x4 ymc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1)
{
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)YMC_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)YMC_PHILOX_M1 * c2;
        uint32_t n0 = ymc_xor3((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = ymc_xor3((uint32_t)(p0 >> 32), c3, k1);
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += YMC_PHILOX_W0;
        k1 += YMC_PHILOX_W1;
    }
    ymc_u32x4 out;
    out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
    return out;
}

/* ------------------------------------------------------------------------ */
/* Draw addressing.                                                         */
/*                                                                          */
/* A chain consumes draws per MH *transition*.  Transitions of one chain    */
/* are numbered t = 0, 1, 2, ... (for the single-level random walk t is the */
/* step index; for two-level delayed acceptance with sub-chain length S,    */
/* fine step n owns t = n*(S+1) + s, s = 0..S-1 the coarse sub-steps and    */
/* s = S the fine accept/reject; for the nested scheme every accept/reject  */
/* decision of any level takes the next t in execution order, so one step   */
/* of level e spans c_e = S_{e-1} c_{e-1} + 1 transitions, c_0 = 1, and top */
/* step n starts at t = n c_{E-1}).  Every draw is addressed, never         */
/* sequenced, so the reference's "proposal == state => no uniform drawn"    */
/* rule (metropolisHastings.py:60-61) does not shift any later draw.        */
/*                                                                          */
/*   key     = (seed[31:0], seed[63:32])                                    */
/*   uniform of transition t:  counter (chain, 0, p[31:0], p[63:32]),       */
/*       p = t >> 1; words {0,1} serve even t, words {2,3} odd t            */
/*   normals 2j, 2j+1 of transition t:                                      */
/*       counter (chain, 1 + j, t[31:0], t[63:32]); Box-Muller on the block */
/* chain is the GLOBAL chain id, so a chain's stream does not depend on how */
/* the ensemble is sharded over GPUs.                                       */
/* ------------------------------------------------------------------------ */

/* m * 2^-53 for 0 <= m <= 2^53, exactly: both halves convert exactly, both products are exact
 * and their sum has at most 53 significant bits (one instruction fewer than a u64 conversion). */
YMC_FN double ymc_u53_to_unit(uint64_t m)
{
    double hi = (double)(uint32_t)(m >> 32);
    double lo = (double)(uint32_t)m;
    return ymc_fma(hi, 0x1p-21, lo * 0x1p-53);
}

/* The 53-bit mantissa m = (hi:lo) >> 11 of two words, as m 2^-53, WITHOUT forming m: m = hi 2^21 + (lo >> 11), so
 * m 2^-53 = hi 2^-32 + (lo >> 11) 2^-53 -- two exact conversions, an exact product and a sum of at most 53
 * significant bits: the same double as ymc_u53_to_unit(m) with a 32-bit shift in place of the 64-bit one
 * (`add` = 1 gives (m + 1) 2^-53: lo >> 11 is at most 2^21 - 1, so the increment stays in its word). */
YMC_FN double ymc_words_to_unit(uint32_t lo, uint32_t hi, uint32_t add)
{
    double h = (double)hi;
    double l = (double)((lo >> 11) + add);
    return ymc_fma(h, 0x1p-32, l * 0x1p-53);
}

/* 53-bit mantissa from two words -> [0,1) on the 2^-53 grid */
YMC_FN double ymc_u01_co(uint32_t lo, uint32_t hi)
{
    return ymc_words_to_unit(lo, hi, 0u);
}

/* 53-bit mantissa from two words -> (0,1] on the 2^-53 grid */
YMC_FN double ymc_u01_oc(uint32_t lo, uint32_t hi)
{
    return ymc_words_to_unit(lo, hi, 1u);
}

YMC_FN ymc_u32x4 ymc_uniform_block(uint64_t seed, uint32_t chain, uint64_t t)
{
    uint64_t p = t >> 1;
    return ymc_philox4x32_10(chain, 0u, (uint32_t)p, (uint32_t)(p >> 32),
                             (uint32_t)seed, (uint32_t)(seed >> 32));
}

YMC_FN double ymc_uniform_from_block(ymc_u32x4 b, uint64_t t)
{
    return (t & 1u) ? ymc_u01_co(b.v[2], b.v[3]) : ymc_u01_co(b.v[0], b.v[1]);
}

#if defined(YMC_DRAW_SPEC_V2X)
/* TIMING EXPERIMENT ONLY (make v2x, tools/ab_spec_v2.sh, profiles/r04_spec_v2.txt): a cheaper draw
 * specification costed against the product's.  ONE Philox block per transition carries the first normal pair (two 42-bit
 * mantissas: words 0, 1 + 10 bits each of word 3) AND the accept uniform (44 bits: word 2 + 12 bits of word 3); beyond the
 * first pair three normal pairs share two blocks (40-bit mantissas).  Results differ from every fixture by construction. */
YMC_FN double ymc_v2x_unit(uint32_t hi, uint32_t low, double low_scale, uint32_t add)
{
    return ymc_fma((double)hi, 0x1p-32, (double)(low + add) * low_scale);
}
#endif

/* the accept/reject uniform of transition t (u ~ U[0,1), metropolisHastings.py:68) */
YMC_FN double ymc_draw_uniform(uint64_t seed, uint32_t chain, uint64_t t)
{
#if defined(YMC_DRAW_SPEC_V2X)
    ymc_u32x4 b = ymc_philox4x32_10(chain, 1u, (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    return ymc_v2x_unit(b.v[2], b.v[3] >> 20, 0x1p-44, 0u);
#else
    return ymc_uniform_from_block(ymc_uniform_block(seed, chain, t), t);
#endif
}

/* ------------------------------------------------------------------------ */
/* log(x), x a positive normal double: table-driven, no division.           */
/*   x = 2^k z,  z in [0.6855, 1.3711);  interval i of 128 holds z, with    */
/*   centre c_i;  r = z / c_i - 1 (one fma with the tabulated 1/c_i, |r| <= */
/*   2^-8);  log x = k ln2 + log c_i + r + r^2 g(r),  g a degree-4 fit of   */
/*   (log1p(r) - r) / r^2.  1.0 is the centre of its interval (c = 1, log c */
/*   = 0), so log(1) = 0 exactly and arguments next to 1 lose nothing.      */
/* Table and coefficients: tools/gen_numerics_tables.py (mpmath).  Measured */
/* error < 1.7 ulp (tests/test_numerics.py); it replaced an fdlibm-style    */
/* log with a division (11 fewer VALU instructions per normal pair).        */
/* On the device the tables (this one and sincospi's, 2 KB each) are read   */
/* from LDS: every kernel that draws calls ymc_stage_tables() first.        */
/* ------------------------------------------------------------------------ */
#define YMC_LOG_OFF 0x3fe5f00000000000ull /* 0x3fe6000000000000 - 2^44 */

#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const double ymc_logtab_rom[256] = {
#include "ymc_logtab.inc"
};
static __device__ const double ymc_sincostab_rom[256] = {
#include "ymc_sincostab.inc"
};
__shared__ double ymc_logtab_lds[256];
__shared__ double ymc_sincostab_lds[256];
#define YMC_LOGTAB ymc_logtab_lds
#define YMC_SINCOSTAB ymc_sincostab_lds
#else
static const double ymc_logtab_host[256] = {
#include "ymc_logtab.inc"
};
static const double ymc_sincostab_host[256] = {
#include "ymc_sincostab.inc"
};
#define YMC_LOGTAB ymc_logtab_host
#define YMC_SINCOSTAB ymc_sincostab_host
#endif

#if defined(__HIPCC__)
/* every thread of the workgroup, before the kernel's first draw (and before any early return) */
__device__ __forceinline__ void ymc_stage_tables()
{
#if defined(__HIP_DEVICE_COMPILE__)
    for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) {
        ymc_logtab_lds[i] = ymc_logtab_rom[i];
        ymc_sincostab_lds[i] = ymc_sincostab_rom[i];
    }
    __syncthreads();
#endif
}
#endif

YMC_FN double ymc_log(double x)
{
    const double Ln2hi = 0x1.62e42fefa3800p-1; /* 42 bits: k * Ln2hi is exact */
    const double Ln2lo = 0x1.ef35793c76730p-45;
    /* everything the reduction needs sits in the HIGH word (YMC_LOG_OFF's low word is zero): the interval index, the
     * exponent k and the exponent bits to clear -- 32-bit operations, the low word of x passes through */
    uint64_t ix = ymc_bits(x);
    uint32_t hx = (uint32_t)(ix >> 32);
    uint32_t ht = hx - (uint32_t)(YMC_LOG_OFF >> 32);
    int32_t i = (int32_t)((ht >> 13) & 127u);
    int32_t k = (int32_t)ht >> 20;
    double z = ymc_from_bits(((uint64_t)(hx - (ht & 0xfff00000u)) << 32) | (uint32_t)ix);
    double invc = YMC_LOGTAB[2 * i], logc = YMC_LOGTAB[2 * i + 1];
    double r = ymc_fma(z, invc, -1.0);
    double kd = (double)k;
    double w = ymc_fma(kd, Ln2hi, logc);
    double hi = w + r;
    double lo = ymc_fma(kd, Ln2lo, (w - hi) + r);
    double r2 = r * r;
    double p = ymc_fma_c(r, ymc_coeff(-0x1.55569bc4f7488p-3), 0x1.999b0eab6ecccp-3);
    double q = ymc_fma_c(r, ymc_coeff(-0x1.ffffffffacc02p-3), 0x1.5555555525c32p-2);
    p = ymc_fma(r2, p, q);
    return ymc_fma(r * r2, p, ymc_fma(r2, -0.5, lo)) + hi;
}

/* ------------------------------------------------------------------------ */
/* exp(x) for any double.  x = k ln2 + r, |r| <= ln2/2; degree-13 Taylor    */
/* polynomial in r by Horner/fma; 2^k applied through the exponent field.   */
/* Tables: tools/gen_numerics_tables.py.  < 1 ulp (tests/test_numerics.py). */
/* ------------------------------------------------------------------------ */
YMC_FN double ymc_scale2(double p, int32_t k) /* p * 2^k, result must be normal */
{
    return ymc_from_bits(ymc_bits(p) + ((uint64_t)(int64_t)k << 52));
}

YMC_FN double ymc_exp(double x)
{
    const double inv_ln2 = 0x1.71547652b82fep+0;
    const double ln2_hi = 0x1.62e42fee00000p-1;
    const double ln2_lo = 0x1.a39ef35793c76p-33;

    /* clamp so that k stays in int range; the special cases are selected at  */
    /* the end (straight-line code: no divergent branches on the GPU)          */
    double xc = (x > 710.0) ? 710.0 : ((x < -746.0) ? -746.0 : x);
    double kd = __builtin_rint(xc * inv_ln2);
    int32_t k = (int32_t)kd;
    double r = ymc_fma(-kd, ln2_hi, xc);
    r = ymc_fma(-kd, ln2_lo, r);

    double p = ymc_coeff(0x1.6124613a86d09p-33);                 /* 1/13! */
    p = ymc_fma_c(p, r, 0x1.1eed8eff8d898p-29);         /* 1/12! */
    p = ymc_fma_c(p, r, 0x1.ae64567f544e4p-26);         /* 1/11! */
    p = ymc_fma_c(p, r, 0x1.27e4fb7789f5cp-22);         /* 1/10! */
    p = ymc_fma_c(p, r, 0x1.71de3a556c734p-19);         /* 1/9!  */
    p = ymc_fma_c(p, r, 0x1.a01a01a01a01ap-16);         /* 1/8!  */
    p = ymc_fma_c(p, r, 0x1.a01a01a01a01ap-13);         /* 1/7!  */
    p = ymc_fma_c(p, r, 0x1.6c16c16c16c17p-10);         /* 1/6!  */
    p = ymc_fma_c(p, r, 0x1.1111111111111p-7);          /* 1/5!  */
    p = ymc_fma_c(p, r, 0x1.5555555555555p-5);          /* 1/4!  */
    p = ymc_fma_c(p, r, 0x1.5555555555555p-3);          /* 1/3!  */
    p = ymc_fma_c(p, r, 0.5);
    p = ymc_fma_c(p, r, 1.0);
    p = ymc_fma_c(p, r, 1.0);                           /* p in [0.70, 1.42] */

    /* 2^k in one or two exactly representable factors */
    int32_t k1 = (k < -1021) ? k + 1000 : ((k > 1022) ? k - 2 : k);
    double f2 = (k < -1021) ? 0x1p-1000 : ((k > 1022) ? 4.0 : 1.0);
    double y = ymc_scale2(p, k1) * f2;

    y = (x > 0x1.62e42fefa39efp+9) ? __builtin_inf() : y;
    y = (x < -745.2) ? 0.0 : y;
    return (x == x) ? y : x;                          /* NaN propagates */
}

/* The Metropolis acceptance probability as far as a DECISION needs it              */
/*   a = exp(log_ratio); a = a if a < 1 else 1;  accept iff u <= a  (mrw.py:54-57) */
/* u lies on the 2^-53 grid of [0, 1), so every a below 2^-53 decides the same way  */
/* (accept iff u == 0).  ymc_accept_prob(d) therefore returns min(1, ymc_exp(d))    */
/* exactly for d >= -64, and exp(-64) (< 2^-92) for d < -64, NaN and +inf giving 1   */
/* like the reference's comparison: same decisions as the full-range exp for every  */
/* input, without its overflow / subnormal / NaN selects (two min/max, one exponent */
/* add instead of fourteen compares and selects).                                  */
YMC_FN double ymc_min_nonan(double a, double b) /* a unless a is NaN or a > b */
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmin(a, b); /* v_min_f64: the other operand when one is NaN */
#else
    return (a < b) ? a : b;
#endif
}

YMC_FN double ymc_accept_prob(double d)
{
    const double inv_ln2 = 0x1.71547652b82fep+0;
    const double ln2_hi = 0x1.62e42fee00000p-1;
    const double ln2_lo = 0x1.a39ef35793c76p-33;

    double x = ymc_min_nonan(d, 0.0);                 /* NaN, +inf, d >= 0 -> 0 -> exp = 1 */
#if defined(__HIP_DEVICE_COMPILE__)
    double xc = __builtin_fmax(x, -64.0);
#else
    double xc = (x < -64.0) ? -64.0 : x;
#endif
    double kd = __builtin_rint(xc * inv_ln2);
    int32_t k = (int32_t)kd;                          /* -93 <= k <= 0 */
    double r = ymc_fma(-kd, ln2_hi, xc);
    r = ymc_fma(-kd, ln2_lo, r);

    double p = ymc_coeff(0x1.6124613a86d09p-33);
    p = ymc_fma_c(p, r, 0x1.1eed8eff8d898p-29);
    p = ymc_fma_c(p, r, 0x1.ae64567f544e4p-26);
    p = ymc_fma_c(p, r, 0x1.27e4fb7789f5cp-22);
    p = ymc_fma_c(p, r, 0x1.71de3a556c734p-19);
    p = ymc_fma_c(p, r, 0x1.a01a01a01a01ap-16);
    p = ymc_fma_c(p, r, 0x1.a01a01a01a01ap-13);
    p = ymc_fma_c(p, r, 0x1.6c16c16c16c17p-10);
    p = ymc_fma_c(p, r, 0x1.1111111111111p-7);
    p = ymc_fma_c(p, r, 0x1.5555555555555p-5);
    p = ymc_fma_c(p, r, 0x1.5555555555555p-3);
    p = ymc_fma_c(p, r, 0.5);
    p = ymc_fma_c(p, r, 1.0);
    p = ymc_fma_c(p, r, 1.0);                           /* p in [0.70, 1.42] */
    return ymc_min_nonan(ymc_scale2(p, k), 1.0);
}

/* ------------------------------------------------------------------------ */
/* sin(pi t), cos(pi t) for 0 <= t < 2^24, table-driven: t = j/64 + r        */
/* exactly, |r| <= 1/128; (S, C) = (sin, cos)(pi j/64) from a 128-entry     */
/* table (exact 0 and +-1 on the axes); sp = sin(pi r) and cm = cos(pi r)-1 */
/* by short Taylor polynomials in r^2 (next terms 4e-19 and 3e-18);         */
/*   sin(pi t) = S + (C sp + S cm),   cos(pi t) = C + (C cm - S sp).        */
/* No quadrant selects, 17 fewer VALU instructions than the 17-term version */
/* it replaced; error < 1.6 ulp (tests/test_numerics.py).  Table:           */
/* tools/gen_numerics_tables.py sincostab; in LDS on the device.            */
/* ------------------------------------------------------------------------ */
YMC_FN void ymc_sincospi(double t, double *sn, double *cs)
{
    double qd = __builtin_rint(t * 64.0);
    double r = ymc_fma(-qd, 0x1p-6, t);               /* exact */
    int32_t j = (int32_t)qd & 127;
    double S = YMC_SINCOSTAB[2 * j], C = YMC_SINCOSTAB[2 * j + 1];
    double r2 = r * r;

    double p = ymc_fma_c(r2, ymc_coeff(-0x1.32d2cce62bd86p-1), 0x1.466bc6775aae2p+1); /* -pi^7/7!, pi^5/5! */
    p = ymc_fma_c(r2, p, -0x1.4abbce625be53p+2);        /* -pi^3/3! */
    p = ymc_fma_c(r2, p, 0x1.921fb54442d18p+1);         /* pi */
    double sp = r * p;
    double q = ymc_fma_c(r2, ymc_coeff(-0x1.55d3c7e3cbffap+0), 0x1.03c1f081b5ac4p+2); /* -pi^6/6!, pi^4/4! */
    q = ymc_fma_c(r2, q, -0x1.3bd3cc9be45dep+2);        /* -pi^2/2 */
    double cm = r2 * q;

    *sn = S + ymc_fma(C, sp, S * cm);
    *cs = C + ymc_fma(-S, sp, C * cm);
}

/* ------------------------------------------------------------------------ */
/* The Box-Muller radius sqrt(w), w = -2 log(u) in [0, 74]: the correctly   */
/* rounded square root of max(w, 2^-100).  The floor only matters for       */
/* u = 1 (w = 0, probability 2^-53: radius 2^-50 instead of 0); with it the */
/* argument never needs the sub-2^-767 rescaling or the zero / infinity     */
/* special cases of a general sqrt, so the device runs the bare Goldschmidt */
/* sequence on v_rsq_f64 (the compiler's own f64 sqrt minus that handling:  */
/* 11 instead of 18 instructions, same bits as the host's sqrt).            */
/* ------------------------------------------------------------------------ */
YMC_FN double ymc_sqrt_radius(double w)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double x = __builtin_fmax(w, 0x1p-100);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = ymc_fma(-h, g, 0.5);
    g = ymc_fma(g, r, g);
    h = ymc_fma(h, r, h);
    double d = ymc_fma(-g, g, x);
    g = ymc_fma(d, h, g);
    d = ymc_fma(-g, g, x);
    return ymc_fma(d, h, g);
#else
    return __builtin_sqrt(w > 0x1p-100 ? w : 0x1p-100);
#endif
}

/* ------------------------------------------------------------------------ */
/* The normal pair (2j, 2j+1) of transition t: Box-Muller on one Philox     */
/* block.  Replaces numpy.random.standard_normal(d) (gaussian.py:63).       */
/* ------------------------------------------------------------------------ */
YMC_FN void ymc_draw_normal_pair(uint64_t seed, uint32_t chain, uint64_t t, uint32_t j,
                                 double *z0, double *z1)
{
    ymc_u32x4 b = ymc_philox4x32_10(chain, 1u + j, (uint32_t)t, (uint32_t)(t >> 32),
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
#if defined(YMC_DRAW_SPEC_V2X)
    double u1 = ymc_v2x_unit(b.v[0], b.v[3] & 0x3ffu, 0x1p-42, 1u);
    double u2 = ymc_v2x_unit(b.v[1], (b.v[3] >> 10) & 0x3ffu, 0x1p-42, 0u);
#else
    double u1 = ymc_u01_oc(b.v[0], b.v[1]);           /* (0,1] */
    double u2 = ymc_u01_co(b.v[2], b.v[3]);           /* [0,1) */
#endif
    double rad = ymc_sqrt_radius(-2.0 * ymc_log(u1));
    double sn, cs;
    ymc_sincospi(u2 + u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}

#if defined(YMC_DRAW_SPEC_V2X)
/* experiment: the normal pairs j0, j0 + 1, j0 + 2 (as z[0..5]) of transition t from TWO blocks, 40-bit mantissas */
YMC_FN void ymc_v2x_normal_triple(uint64_t seed, uint32_t chain, uint64_t t, uint32_t j0, double *za, double *zb, double *zc,
                                  double *zd, double *ze, double *zf)
{
    ymc_u32x4 a = ymc_philox4x32_10(chain, 0x40000000u + j0, (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    ymc_u32x4 b = ymc_philox4x32_10(chain, 0x40000001u + j0, (uint32_t)t, (uint32_t)(t >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
#define YMC_V2X_PAIR(o0, o1, h1, h2, l1, l2)                              \
    {                                                                     \
        double u1 = ymc_v2x_unit(h1, l1, 0x1p-40, 1u);                    \
        double u2 = ymc_v2x_unit(h2, l2, 0x1p-40, 0u);                    \
        double rad = ymc_sqrt_radius(-2.0 * ymc_log(u1));                 \
        double sn, cs;                                                    \
        ymc_sincospi(u2 + u2, &sn, &cs);                                  \
        *(o0) = rad * cs;                                                 \
        *(o1) = rad * sn;                                                 \
    }
    YMC_V2X_PAIR(za, zb, a.v[0], a.v[1], b.v[2] & 0xffu, (b.v[2] >> 8) & 0xffu)
    YMC_V2X_PAIR(zc, zd, a.v[2], a.v[3], (b.v[2] >> 16) & 0xffu, b.v[2] >> 24)
    YMC_V2X_PAIR(ze, zf, b.v[0], b.v[1], b.v[3] & 0xffu, (b.v[3] >> 8) & 0xffu)
#undef YMC_V2X_PAIR
}
#endif

/* ------------------------------------------------------------------------ */
/* Gaussian-noise likelihood of a forward value F shared by all n data rows */
/* (the linear model: likelihood.py:74-84 broadcasts F against every row):  */
/*   sum_i |F - D_i|^2_P  =  |F - Dbar|^2_{nP}  +  sum_i |D_i - Dbar|^2_P   */
/* with Dbar the row mean.  The right-hand side costs one */
/* row instead of n per evaluation; its two data-only parts are computed    */
/* once, here, in the ensemble's canonical order (four interleaved partial  */
/* sums).  For n = 1 both sides are the same operations.                    */
/* ------------------------------------------------------------------------ */
YMC_FN void ymc_linear_data_summary(const double *data, const double *prec, int32_t n, int32_t m,
                                    double *dbar, double *nprec, double *cst)
{
    for (int32_t j = 0; j < m; ++j) {
        double sum = data[j];
        for (int32_t i = 1; i < n; ++i) sum = sum + data[(int64_t)i * m + j];
        dbar[j] = sum / (double)n;
        nprec[j] = (double)n * prec[j];
    }
    double total = 0.0;
    for (int32_t i = 0; i < n; ++i) {
        double S[4] = {0.0, 0.0, 0.0, 0.0};
        for (int32_t j = 0; j < m; ++j) {
            double r = data[(int64_t)i * m + j] - dbar[j];
            double pr = prec[j] * r;
            S[j & 3] = S[j & 3] + r * pr;
        }
        total = total + ((S[0] + S[1]) + (S[2] + S[3]));
    }
    *cst = total;
}

/* ------------------------------------------------------------------------ */
/* |r|^2 in the norm of a DENSE covariance Sigma = L L^T                     */
/* (CovarianceMatrix.induced_norm_squared over DenseCovarianceMatrix,        */
/* covariance.py:19-22,69-94) from Linv = L^-1 (lower triangular, row-major  */
/* n x n, formed once on the host):  w = Linv r,  |r|^2 = |w|^2.  The        */
/* reference solves two triangular systems with LAPACK and takes a BLAS dot  */
/* product -- orders that are not specified; the ensemble's order is: row i  */
/* of Linv r as a left-to-right fma chain, the squares as the canonical four */
/* interleaved partial sums.  Values agree with the reference to rounding.   */
/* ------------------------------------------------------------------------ */
YMC_FN double ymc_whitened_norm_sq(const double *linv, const double *r, int32_t n)
{
    double S[4] = {0.0, 0.0, 0.0, 0.0};
    for (int32_t i = 0; i < n; ++i) {
        double w = linv[(int64_t)i * n] * r[0];
        for (int32_t j = 1; j <= i; ++j) w = ymc_fma(linv[(int64_t)i * n + j], r[j], w);
        S[i & 3] = S[i & 3] + w * w;
    }
    return (S[0] + S[1]) + (S[2] + S[3]);
}

/* ------------------------------------------------------------------------ */
/* A dense NOISE covariance beyond the lane kernels' sizes (matrix-core       */
/* path): |L^-1 (A v + b - D_i)|^2 = |(L^-1 A) v + L^-1 b - L^-1 D_i|^2, so the */
/* problem is whitened ONCE when it is lowered -- A, b and the data rows are  */
/* multiplied by L^-1 and the noise becomes the identity -- and the kernels   */
/* run the diagonal case.  out[i][c] = sum_{k <= i} linv[i][k] src[k][c], one  */
/* k-ordered fma chain per entry (host and CPU checker share this order).     */
/* ------------------------------------------------------------------------ */
YMC_FN void ymc_lower_times(const double *linv, int32_t m, const double *src, int32_t cols, double *out)
{
    for (int32_t i = 0; i < m; ++i)
        for (int32_t c = 0; c < cols; ++c) {
            double acc = 0.0;
            for (int32_t k = 0; k <= i; ++k) acc = ymc_fma(linv[(int64_t)i * m + k], src[(int64_t)k * cols + c], acc);
            out[(int64_t)i * cols + c] = acc;
        }
}

/* ------------------------------------------------------------------------ */
/* Lotka-Volterra flow (testSetup.py:96-99) and one step of classical RK4,  */
/* the ensemble's integrator (the reference integrates adaptively with      */
/* scipy's solve_ivp; agreement of the forward map is checked to 1e-3, so   */
/* the arithmetic below is the build's own): written with fused multiply-   */
/* adds, dx = x (alpha - beta y), dy = y (delta x - gamma) -- 30 operations */
/* per step instead of 58.                                                  */
/* ------------------------------------------------------------------------ */
YMC_FN void ymc_lv_flow(double x, double y, double alpha, double beta, double gamma, double delta,
                        double *dx, double *dy)
{
    *dx = x * ymc_fma(-beta, y, alpha);
    *dy = y * ymc_fma(delta, x, -gamma);
}

/* h2 = h / 2, h6 = h / 6 */
YMC_FN void ymc_lv_rk4_step(double *x, double *y, double alpha, double beta, double gamma, double delta,
                            double h, double h2, double h6)
{
    double k1x, k1y, k2x, k2y, k3x, k3y, k4x, k4y;
    ymc_lv_flow(*x, *y, alpha, beta, gamma, delta, &k1x, &k1y);
    ymc_lv_flow(ymc_fma(h2, k1x, *x), ymc_fma(h2, k1y, *y), alpha, beta, gamma, delta, &k2x, &k2y);
    ymc_lv_flow(ymc_fma(h2, k2x, *x), ymc_fma(h2, k2y, *y), alpha, beta, gamma, delta, &k3x, &k3y);
    ymc_lv_flow(ymc_fma(h, k3x, *x), ymc_fma(h, k3y, *y), alpha, beta, gamma, delta, &k4x, &k4y);
    *x = ymc_fma(h6, ymc_fma(2.0, k2x, k1x) + ymc_fma(2.0, k3x, k4x), *x);
    *y = ymc_fma(h6, ymc_fma(2.0, k2y, k1y) + ymc_fma(2.0, k3y, k4y), *y);
}

/* The same flow integrated adaptively over [0, T]: the Dormand-Prince 5(4) pair (the tableau of scipy's RK45;
 * the reference's LotkaVolterraSolver hands `method` and `rtol` to scipy.integrate.solve_ivp, testSetup.py:118-141)
 * with first-same-as-last and a controller that only halves and doubles the step -- every quantity an IEEE sum,
 * product or fma in a fixed order, so that device and checker agree to the bit.  A step is accepted when every
 * component's error estimate is within atol + rtol |y|; it doubles after a step whose estimate is within
 * 1/64 of that bound (the estimate scales with h^5: x 32), halves after a rejected one (to a quarter beyond 32 x the
 * bound).  Returns false -- state unspecified -- when the step has shrunk below T 2^-40 (a singular right-hand side
 * or a non-finite state) or after max_attempts steps.  Classical RK4 in 50 steps overflows at the reference
 * example's own starting point, theta = (-7, 2.8), where the predator first grows at rate 16 (h lambda = 3.3). */
#define YMC_LV_MAX_ATTEMPTS 100000 /* steps, accepted or not, one solve may take */

YMC_FN int ymc_lv_dp45_solve(double *x, double *y, double alpha, double beta, double gamma, double delta, double T,
                             double rtol, double atol, int max_attempts)
{
    const double a21 = 1.0 / 5.0;
    const double a31 = 3.0 / 40.0, a32 = 9.0 / 40.0;
    const double a41 = 44.0 / 45.0, a42 = -56.0 / 15.0, a43 = 32.0 / 9.0;
    const double a51 = 19372.0 / 6561.0, a52 = -25360.0 / 2187.0, a53 = 64448.0 / 6561.0, a54 = -212.0 / 729.0;
    const double a61 = 9017.0 / 3168.0, a62 = -355.0 / 33.0, a63 = 46732.0 / 5247.0, a64 = 49.0 / 176.0,
                 a65 = -5103.0 / 18656.0;
    const double b1 = 35.0 / 384.0, b3 = 500.0 / 1113.0, b4 = 125.0 / 192.0, b5 = -2187.0 / 6784.0, b6 = 11.0 / 84.0;
    const double e1 = 71.0 / 57600.0, e3 = -71.0 / 16695.0, e4 = 71.0 / 1920.0, e5 = -17253.0 / 339200.0,
                 e6 = 22.0 / 525.0, e7 = -1.0 / 40.0;
    const double h_min = T * 0x1p-40, huge = 1.79769313486231570815e308;
    double u = *x, v = *y, t = 0.0, h = T * 0x1p-7;
    double k1u, k1v;
    ymc_lv_flow(u, v, alpha, beta, gamma, delta, &k1u, &k1v);
    for (int attempt = 0; attempt < max_attempts; ++attempt) {
        const double left = T - t;
        const int last = !(h < left);
        const double hs = last ? left : h;
        double k2u, k2v, k3u, k3v, k4u, k4v, k5u, k5v, k6u, k6v, k7u, k7v;
        ymc_lv_flow(ymc_fma(hs, a21 * k1u, u), ymc_fma(hs, a21 * k1v, v), alpha, beta, gamma, delta, &k2u, &k2v);
        ymc_lv_flow(ymc_fma(hs, ymc_fma(a32, k2u, a31 * k1u), u), ymc_fma(hs, ymc_fma(a32, k2v, a31 * k1v), v), alpha, beta,
                    gamma, delta, &k3u, &k3v);
        ymc_lv_flow(ymc_fma(hs, ymc_fma(a43, k3u, ymc_fma(a42, k2u, a41 * k1u)), u),
                    ymc_fma(hs, ymc_fma(a43, k3v, ymc_fma(a42, k2v, a41 * k1v)), v), alpha, beta, gamma, delta, &k4u, &k4v);
        ymc_lv_flow(ymc_fma(hs, ymc_fma(a54, k4u, ymc_fma(a53, k3u, ymc_fma(a52, k2u, a51 * k1u))), u),
                    ymc_fma(hs, ymc_fma(a54, k4v, ymc_fma(a53, k3v, ymc_fma(a52, k2v, a51 * k1v))), v), alpha, beta, gamma,
                    delta, &k5u, &k5v);
        ymc_lv_flow(ymc_fma(hs, ymc_fma(a65, k5u, ymc_fma(a64, k4u, ymc_fma(a63, k3u, ymc_fma(a62, k2u, a61 * k1u)))), u),
                    ymc_fma(hs, ymc_fma(a65, k5v, ymc_fma(a64, k4v, ymc_fma(a63, k3v, ymc_fma(a62, k2v, a61 * k1v)))), v),
                    alpha, beta, gamma, delta, &k6u, &k6v);
        const double un = ymc_fma(hs, ymc_fma(b6, k6u, ymc_fma(b5, k5u, ymc_fma(b4, k4u, ymc_fma(b3, k3u, b1 * k1u)))), u);
        const double vn = ymc_fma(hs, ymc_fma(b6, k6v, ymc_fma(b5, k5v, ymc_fma(b4, k4v, ymc_fma(b3, k3v, b1 * k1v)))), v);
        ymc_lv_flow(un, vn, alpha, beta, gamma, delta, &k7u, &k7v);
        const double eu = __builtin_fabs(
            hs * ymc_fma(e7, k7u, ymc_fma(e6, k6u, ymc_fma(e5, k5u, ymc_fma(e4, k4u, ymc_fma(e3, k3u, e1 * k1u))))));
        const double ev = __builtin_fabs(
            hs * ymc_fma(e7, k7v, ymc_fma(e6, k6v, ymc_fma(e5, k5v, ymc_fma(e4, k4v, ymc_fma(e3, k3v, e1 * k1v))))));
        /* the bound is relative to the state the step STARTS from: with max(|y|, |y_new|) -- scipy's choice -- a step that
         * explodes inflates its own tolerance (at rtol 1e-2 one in eighty solves around theta = (-7, 2.8) accepted a jump to
         * 1e10 and a negative population that way) */
        const double su = ymc_fma(rtol, __builtin_fabs(u), atol), sv = ymc_fma(rtol, __builtin_fabs(v), atol);
        const double aun = __builtin_fabs(un), avn = __builtin_fabs(vn);
        /* (a population does not change sign: the exact flow keeps the axes invariant) */
        const int keeps_sign = !(u > 0.0 && un < 0.0) && !(v > 0.0 && vn < 0.0) && !(u < 0.0 && un > 0.0) && !(v < 0.0 && vn > 0.0);
        if (eu <= su && ev <= sv && aun <= huge && avn <= huge && keeps_sign) { /* false for a NaN or an overflow anywhere */
            u = un, v = vn, k1u = k7u, k1v = k7v;
            if (last) {
                *x = u, *y = v;
                return 1;
            }
            t = t + hs;
            if (64.0 * eu <= su && 64.0 * ev <= sv) h = 2.0 * h;
        } else {
            h = (eu <= 32.0 * su && ev <= 32.0 * sv) ? 0.5 * hs : 0.25 * hs;
            if (!(h >= h_min)) return 0;
        }
    }
    return 0;
}

#endif /* YMC_NUMERICS_H */
