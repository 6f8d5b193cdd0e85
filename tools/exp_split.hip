// Experiment harness: producer/consumer variants of the single-level random-walk kernel for the
// 2-D Gaussian target at K = 65 536 chains (1 wave per SIMD when one lane owns one chain).
// All variants must give the fused kernel's bits; they are timed in interleaved rounds in ONE process.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../yagre-mcmc_amd/csrc exp_split.hip -o exp_split
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ymc_device.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int D = 2;
using Dens = Density<YMC_DENSITY_GAUSS_WHITENED, D>;

struct Args {
    DensityDev<D> tgt;
    double sigma[D];
    uint64_t seed, K, t0;
    uint32_t nsteps;
    double *x, *lp, *wmean, *wm2, *traj;
    const double* recip;   // recip[n] = 1 / (t0 + n + 1)
    uint64_t* acc_count;
    uint8_t* acc;
};

__device__ __forceinline__ void tail(const Args& a, uint64_t k, uint32_t n, double (&x)[D], double& lp, uint64_t& accepted,
                                     double (&wmean)[D], double (&wm2)[D], const double (&delta)[D], double u)
{
    double y[D];
#pragma unroll
    for (int j = 0; j < D; ++j) y[j] = x[j] + delta[j];
    uint32_t st = 0;
    const bool moved = !ymc_states_equal<D>(y, x, 0);
    const double lpy = Dens::eval(a.tgt, y, st);
    const bool acc = ymc_accept(lpy - lp, moved, ymc_finite(lpy), u, st);
    ymc_welford<D>(x, (double)(a.t0 + n + 1), a.recip[n], wmean, wm2);
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = acc ? y[j] : x[j];
    lp = acc ? lpy : lp;
    accepted += acc ? 1u : 0u;
    ymc_store_row<D>(a.traj + (uint64_t)n * a.K * D, k, x);
    a.acc[(uint64_t)n * a.K + k] = acc ? 1 : 0;
}

__global__ __launch_bounds__(256) void fused(const Args a)
{
    ymc_stage_tables();
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t chain = (uint32_t)k;
    double x[D], wmean[D], wm2[D], sigma[D] = {a.sigma[0], a.sigma[1]};
    ymc_load_row<D>(a.x, k, x);
    ymc_load_row<D>(a.wmean, k, wmean);
    ymc_load_row<D>(a.wm2, k, wm2);
    double lp = a.lp[k];
    uint64_t accepted = a.acc_count[k];
    UniformCache uc;
    uc.valid = false;
    uc.pair = 0;
#pragma unroll 1
    for (uint32_t n = 0; n < a.nsteps; ++n) {
        const uint64_t t = a.t0 + n;
        double z0, z1;
        ymc_draw_normal_pair(a.seed, chain, t, 0, &z0, &z1);
        const double delta[D] = {sigma[0] * z0, sigma[1] * z1};
        const double u = ymc_uniform_cached(uc, a.seed, chain, t);
        tail(a, k, n, x, lp, accepted, wmean, wm2, delta, u);
    }
    ymc_store_row<D>(a.x, k, x);
    ymc_store_row<D>(a.wmean, k, wmean);
    ymc_store_row<D>(a.wm2, k, wm2);
    a.lp[k] = lp;
    a.acc_count[k] = accepted;
}

// NP producer waves + 1 consumer wave per 64 chains. Ring buffer of B = 2*NP*R transitions, double-buffered.
// The consumer reads a whole buffer into registers right after the barrier (one LDS latency per buffer).
template <int NP, int R, int PRIO>
__global__ __launch_bounds__(64 * (NP + 1)) void split(const Args a)
{
    ymc_stage_tables();
    constexpr int B = 2 * NP * R;
    __shared__ double ring[2][B][D + 1][64];
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const uint64_t k = (uint64_t)blockIdx.x * 64 + lane;
    const uint32_t chain = (uint32_t)k;
    const uint32_t nb = a.nsteps / B;  // harness: nsteps is a multiple of B, t0 even
    if (role != 0) {
        const int p = role - 1;
        const double s0 = a.sigma[0], s1 = a.sigma[1];
        auto fill = [&](uint32_t b) {
#pragma unroll 1
            for (int r = 0; r < R; ++r) {
                const int q = 2 * (NP * r + p);
                const uint64_t t = a.t0 + b * B + q;
                const ymc_u32x4 ub = ymc_uniform_block(a.seed, chain, t);
                ring[b & 1][q][D][lane] = ymc_u01_co(ub.v[0], ub.v[1]);
                ring[b & 1][q + 1][D][lane] = ymc_u01_co(ub.v[2], ub.v[3]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double z0, z1;
                    ymc_draw_normal_pair(a.seed, chain, t + h, 0, &z0, &z1);
                    ring[b & 1][q + h][0][lane] = s0 * z0;
                    ring[b & 1][q + h][1][lane] = s1 * z1;
                }
            }
        };
        fill(0);
        __syncthreads();
        for (uint32_t b = 0; b < nb; ++b) {
            if (b + 1 < nb) fill(b + 1);
            __syncthreads();
        }
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    double x[D], wmean[D], wm2[D];
    ymc_load_row<D>(a.x, k, x);
    ymc_load_row<D>(a.wmean, k, wmean);
    ymc_load_row<D>(a.wm2, k, wm2);
    double lp = a.lp[k];
    uint64_t accepted = a.acc_count[k];
    __syncthreads();
    for (uint32_t b = 0; b < nb; ++b) {
        double dl[B][D], u[B];
#pragma unroll
        for (int s = 0; s < B; ++s) {
            dl[s][0] = ring[b & 1][s][0][lane];
            dl[s][1] = ring[b & 1][s][1][lane];
            u[s] = ring[b & 1][s][D][lane];
        }
#pragma unroll
        for (int s = 0; s < B; ++s) tail(a, k, b * B + s, x, lp, accepted, wmean, wm2, dl[s], u[s]);
        __syncthreads();
    }
    ymc_store_row<D>(a.x, k, x);
    ymc_store_row<D>(a.wmean, k, wmean);
    ymc_store_row<D>(a.wm2, k, wm2);
    a.lp[k] = lp;
    a.acc_count[k] = accepted;
}


// SIMD-balanced variant: one 1024-thread workgroup = 16 waves = 4 per SIMD (wave w runs on SIMD w % 4) for
// 256 chains.  Chain group g = w & 3 (64 chains) lives on SIMD g: wave g is its consumer, waves 4+g, 8+g,
// 12+g its three producers, so every SIMD carries 1 consumer + 3 producers (in `split` the consumers of
// four workgroups pile up on SIMD 0).
template <int R, int PRIO, int LIGHT>
__global__ __launch_bounds__(1024) void split16(const Args a)
{
    ymc_stage_tables();
    constexpr int NP = 3;
    constexpr int B = 2 * NP * R;
    __shared__ double ring[4][2][B][D + 1][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = w & 3, role = w >> 2;
    const uint64_t k = (uint64_t)blockIdx.x * 256 + g * 64 + lane;
    const uint32_t chain = (uint32_t)k;
    const uint32_t nb = a.nsteps / B;
    if (role != 0) {
        const int p = role - 1;
        const double s0 = a.sigma[0], s1 = a.sigma[1];
        auto fill = [&](uint32_t b) {
#pragma unroll 1
            for (int r = 0; r < R; ++r) {
                const int q = 2 * (NP * r + p);
                const uint64_t t = a.t0 + b * B + q;
                const ymc_u32x4 ub = ymc_uniform_block(a.seed, chain, t);
                ring[g][b & 1][q][D][lane] = ymc_u01_co(ub.v[0], ub.v[1]);
                ring[g][b & 1][q + 1][D][lane] = ymc_u01_co(ub.v[2], ub.v[3]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double z0, z1;
                    ymc_draw_normal_pair(a.seed, chain, t + h, 0, &z0, &z1);
                    ring[g][b & 1][q + h][0][lane] = s0 * z0;
                    ring[g][b & 1][q + h][1][lane] = s1 * z1;
                }
            }
        };
        fill(0);
        __syncthreads();
        for (uint32_t b = 0; b < nb; ++b) {
            if (b + 1 < nb) fill(b + 1);
            __syncthreads();
        }
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    double x[D], wmean[D], wm2[D];
    ymc_load_row<D>(a.x, k, x);
    ymc_load_row<D>(a.wmean, k, wmean);
    ymc_load_row<D>(a.wm2, k, wm2);
    double lp = a.lp[k];
    uint64_t accepted = a.acc_count[k];
    __syncthreads();
    for (uint32_t b = 0; b < nb; ++b) {
        double dl[B][D], u[B];
#pragma unroll
        for (int s = 0; s < B; ++s) {
            dl[s][0] = ring[g][b & 1][s][0][lane];
            dl[s][1] = ring[g][b & 1][s][1][lane];
            u[s] = ring[g][b & 1][s][D][lane];
        }
#pragma unroll
        for (int s = 0; s < B; ++s) tail(a, k, b * B + s, x, lp, accepted, wmean, wm2, dl[s], u[s]);
        __syncthreads();
    }
    ymc_store_row<D>(a.x, k, x);
    ymc_store_row<D>(a.wmean, k, wmean);
    ymc_store_row<D>(a.wm2, k, wm2);
    a.lp[k] = lp;
    a.acc_count[k] = accepted;
}


// Four roles per SIMD ("quad"): 1024-thread workgroup, 256 chains, chain group g = w & 3 on SIMD g.
//   role 0  C  consumer: y = x + delta, density, accept, select; writes x', acc to the out ring
//   role 1  R  recorder: Welford + global stores of the buffer before last, and the uniforms of the next
//   role 2  PA normals (delta = sigma z) of the even transitions of the next buffer
//   role 3  PB normals of the odd transitions
// One barrier per B transitions; per transition the four waves carry ~50 / ~52 / ~56 / ~56 VALU instructions.
__device__ __forceinline__ void ctail(const Args& a, double (&x)[D], double& lp, uint64_t& accepted, const double (&delta)[D],
                                      double u, bool& acc_out)
{
    double y[D];
#pragma unroll
    for (int j = 0; j < D; ++j) y[j] = x[j] + delta[j];
    uint32_t st = 0;
    const bool moved = !ymc_states_equal<D>(y, x, 0);
    const double lpy = Dens::eval(a.tgt, y, st);
    const bool acc = ymc_accept(lpy - lp, moved, ymc_finite(lpy), u, st);
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = acc ? y[j] : x[j];
    lp = acc ? lpy : lp;
    accepted += acc ? 1u : 0u;
    acc_out = acc;
}

template <int B, int PRIO>
__global__ __launch_bounds__(1024) void quad(const Args a)
{
    ymc_stage_tables();
    __shared__ double draws[4][2][B][D + 1][64];
    __shared__ double outs[4][2][B][D][64];
    __shared__ uint8_t oacc[4][2][B][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = w & 3, role = w >> 2;
    const uint64_t k = (uint64_t)blockIdx.x * 256 + g * 64 + lane;
    const uint32_t chain = (uint32_t)k;
    const uint32_t nb = a.nsteps / B;  // harness: nsteps multiple of B, B even, t0 even
    if (role >= 2) {
        const int par = role - 2;
        const double s0 = a.sigma[0], s1 = a.sigma[1];
        for (uint32_t i = 0; i < nb + 2; ++i) {
            if (i < nb) {
#pragma unroll
                for (int s = par; s < B; s += 2) {
                    double z0, z1;
                    ymc_draw_normal_pair(a.seed, chain, a.t0 + (uint64_t)i * B + s, 0, &z0, &z1);
                    draws[g][i & 1][s][0][lane] = s0 * z0;
                    draws[g][i & 1][s][1][lane] = s1 * z1;
                }
            }
            __syncthreads();
        }
        return;
    }
    if (role == 1) {
        double xp[D], wmean[D], wm2[D];
        ymc_load_row<D>(a.x, k, xp);
        ymc_load_row<D>(a.wmean, k, wmean);
        ymc_load_row<D>(a.wm2, k, wm2);
        for (uint32_t i = 0; i < nb + 2; ++i) {
            if (i < nb) {
#pragma unroll
                for (int s = 0; s < B; s += 2) {
                    const ymc_u32x4 ub = ymc_uniform_block(a.seed, chain, a.t0 + (uint64_t)i * B + s);
                    draws[g][i & 1][s][D][lane] = ymc_u01_co(ub.v[0], ub.v[1]);
                    draws[g][i & 1][s + 1][D][lane] = ymc_u01_co(ub.v[2], ub.v[3]);
                }
            }
            if (i >= 2) {
                const uint32_t b = i - 2;
#pragma unroll
                for (int s = 0; s < B; ++s) {
                    const uint32_t n = b * B + s;
                    double xn[D] = {outs[g][b & 1][s][0][lane], outs[g][b & 1][s][1][lane]};
                    const uint8_t ac = oacc[g][b & 1][s][lane];
                    ymc_welford<D>(xp, (double)(a.t0 + n + 1), a.recip[n], wmean, wm2);
                    ymc_store_row<D>(a.traj + (uint64_t)n * a.K * D, k, xn);
                    a.acc[(uint64_t)n * a.K + k] = ac;
                    xp[0] = xn[0];
                    xp[1] = xn[1];
                }
            }
            __syncthreads();
        }
        ymc_store_row<D>(a.wmean, k, wmean);
        ymc_store_row<D>(a.wm2, k, wm2);
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    double x[D];
    ymc_load_row<D>(a.x, k, x);
    double lp = a.lp[k];
    uint64_t accepted = a.acc_count[k];
    for (uint32_t i = 0; i < nb + 2; ++i) {
        if (i >= 1 && i <= nb) {
            const uint32_t b = i - 1;
            double dl[B][D], u[B];
#pragma unroll
            for (int s = 0; s < B; ++s) {
                dl[s][0] = draws[g][b & 1][s][0][lane];
                dl[s][1] = draws[g][b & 1][s][1][lane];
                u[s] = draws[g][b & 1][s][D][lane];
            }
#pragma unroll
            for (int s = 0; s < B; ++s) {
                bool ac;
                ctail(a, x, lp, accepted, dl[s], u[s], ac);
                outs[g][b & 1][s][0][lane] = x[0];
                outs[g][b & 1][s][1][lane] = x[1];
                oacc[g][b & 1][s][lane] = ac ? 1 : 0;
            }
        }
        __syncthreads();
    }
    ymc_store_row<D>(a.x, k, x);
    a.lp[k] = lp;
    a.acc_count[k] = accepted;
}

// Phase-locked variant: ONE workgroup of NW waves per CU-sized slice of NW/4*... chains.  All NW waves
// first PRODUCE the draws of B transitions of the workgroup's CW*64 chains into LDS (high occupancy,
// no serial work), barrier, then the first CW waves CONSUME them (one lane = one chain) while the
// other waves wait at the next barrier: the serial part never competes for issue slots.
template <int NW, int CW, int B>
__global__ __launch_bounds__(64 * NW) void phased(const Args a)
{
    ymc_stage_tables();
    extern __shared__ double ring[];   // [B][D+1][CW*64]
    constexpr int NC = CW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t kbase = (uint64_t)blockIdx.x * NC;
    const uint32_t nb = a.nsteps / B;  // harness: multiple of B, t0 even, B even
    const double s0 = a.sigma[0], s1 = a.sigma[1];
    // consumer state
    double x[D] = {0, 0}, wmean[D] = {0, 0}, wm2[D] = {0, 0};
    double lp = 0;
    uint64_t accepted = 0;
    const uint64_t kc = kbase + (uint64_t)wave * 64 + lane;
    if (wave < CW) {
        ymc_load_row<D>(a.x, kc, x);
        ymc_load_row<D>(a.wmean, kc, wmean);
        ymc_load_row<D>(a.wm2, kc, wm2);
        lp = a.lp[kc];
        accepted = a.acc_count[kc];
    }
    for (uint32_t b = 0; b < nb; ++b) {
        // produce: work items = (pair of transitions, chain); NC chains x B/2 pairs over NW*64 threads
        for (int item = tid; item < NC * (B / 2); item += 64 * NW) {
            const int c = item % NC, pr = item / NC;
            const uint32_t chain = (uint32_t)(kbase + c);
            const uint64_t t = a.t0 + (uint64_t)b * B + 2 * pr;
            const ymc_u32x4 ub = ymc_uniform_block(a.seed, chain, t);
            ring[((2 * pr) * (D + 1) + D) * NC + c] = ymc_u01_co(ub.v[0], ub.v[1]);
            ring[((2 * pr + 1) * (D + 1) + D) * NC + c] = ymc_u01_co(ub.v[2], ub.v[3]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double z0, z1;
                ymc_draw_normal_pair(a.seed, chain, t + h, 0, &z0, &z1);
                ring[((2 * pr + h) * (D + 1) + 0) * NC + c] = s0 * z0;
                ring[((2 * pr + h) * (D + 1) + 1) * NC + c] = s1 * z1;
            }
        }
        __syncthreads();
        if (wave < CW) {
            const int c = wave * 64 + lane;
#pragma unroll 2
            for (int s = 0; s < B; ++s) {
                const double dl[D] = {ring[(s * (D + 1) + 0) * NC + c], ring[(s * (D + 1) + 1) * NC + c]};
                const double u = ring[(s * (D + 1) + D) * NC + c];
                tail(a, kc, b * B + s, x, lp, accepted, wmean, wm2, dl, u);
            }
        }
        __syncthreads();
    }
    if (wave < CW) {
        ymc_store_row<D>(a.x, kc, x);
        ymc_store_row<D>(a.wmean, kc, wmean);
        ymc_store_row<D>(a.wm2, kc, wm2);
        a.lp[kc] = lp;
        a.acc_count[kc] = accepted;
    }
}

template <int NW, int CW, int B>
void launch_phased(const Args& a, uint64_t K)
{
    const size_t lds = (size_t)B * (D + 1) * CW * 64 * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&phased<NW, CW, B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((phased<NW, CW, B>), dim3(K / (CW * 64)), dim3(64 * NW), lds, 0, a);
}

struct Bufs {
    double *x, *lp, *wmean, *wm2, *traj;
    uint64_t* acc_count;
    uint8_t* acc;
};

int main(int argc, char** argv)
{
    const uint64_t K = argc > 1 ? atoll(argv[1]) : 65536;
    const uint32_t T = 960;  // multiple of 4, 6, 12, 16, 24, 32
    Args a;
    memset(&a, 0, sizeof a);
    a.tgt.mean[0] = 1.0; a.tgt.mean[1] = 1.5;
    a.tgt.whiten[0] = -0.34990314; a.tgt.whiten[1] = -0.60586986; a.tgt.whiten[2] = -1.2849522; a.tgt.whiten[3] = 0.16498339;
    a.tgt.log_norm = 4.0;
    a.sigma[0] = a.sigma[1] = 1.0;
    a.seed = 20250103; a.K = K; a.t0 = 0; a.nsteps = T;
    auto alloc = [&](Bufs& b) {
        CK(hipMalloc(&b.x, K * D * 8)); CK(hipMalloc(&b.lp, K * 8)); CK(hipMalloc(&b.wmean, K * D * 8)); CK(hipMalloc(&b.wm2, K * D * 8));
        CK(hipMalloc(&b.traj, (size_t)T * K * D * 8)); CK(hipMalloc(&b.acc_count, K * 8)); CK(hipMalloc(&b.acc, (size_t)T * K));
    };
    auto reset = [&](Bufs& b) {
        std::vector<double> x0(K * D); for (uint64_t i = 0; i < K; ++i) { x0[2 * i] = -8; x0[2 * i + 1] = -7; }
        std::vector<double> lp0(K, -50.0);
        CK(hipMemcpy(b.x, x0.data(), K * D * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(b.lp, lp0.data(), K * 8, hipMemcpyHostToDevice));
        CK(hipMemset(b.wmean, 0, K * D * 8)); CK(hipMemset(b.wm2, 0, K * D * 8)); CK(hipMemset(b.acc_count, 0, K * 8));
    };
    Bufs ref, alt;
    alloc(ref); alloc(alt);
    {
        std::vector<double> rc(T);
        for (uint32_t n = 0; n < T; ++n) rc[n] = 1.0 / (double)(n + 1);
        double* d; CK(hipMalloc(&d, T * 8)); CK(hipMemcpy(d, rc.data(), T * 8, hipMemcpyHostToDevice));
        a.recip = d;
    }
    auto bind = [&](Bufs& b) { a.x = b.x; a.lp = b.lp; a.wmean = b.wmean; a.wm2 = b.wm2; a.traj = b.traj; a.acc_count = b.acc_count; a.acc = b.acc; };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Variant { const char* name; void (*launch)(const Args&, uint64_t); };
    Variant vars[] = {
        {"fused", [](const Args& a, uint64_t K) { hipLaunchKernelGGL(fused, dim3(K / 256), dim3(256), 0, 0, a); }},
        {"split NP2 R2", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((split<2, 2, 0>), dim3(K / 64), dim3(192), 0, 0, a); }},
        {"split NP2 R2 prio", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((split<2, 2, 1>), dim3(K / 64), dim3(192), 0, 0, a); }},
        {"split NP3 R2 prio", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((split<3, 2, 1>), dim3(K / 64), dim3(256), 0, 0, a); }},
        {"split NP3 R1 prio", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((split<3, 1, 1>), dim3(K / 64), dim3(256), 0, 0, a); }},
        {"split NP4 R1 prio", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((split<4, 1, 1>), dim3(K / 64), dim3(320), 0, 0, a); }},
        {"split NP5 R1 prio", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((split<5, 1, 1>), dim3(K / 64), dim3(384), 0, 0, a); }},
        {"split16 R1", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((split16<1, 0, 0>), dim3(K / 256), dim3(1024), 0, 0, a); }},
        {"split16 R1 prio", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((split16<1, 1, 0>), dim3(K / 256), dim3(1024), 0, 0, a); }},
        {"split16 R2 prio", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((split16<2, 1, 0>), dim3(K / 256), dim3(1024), 0, 0, a); }},
        {"quad B4", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((quad<4, 0>), dim3(K / 256), dim3(1024), 0, 0, a); }},
        {"quad B6", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((quad<6, 0>), dim3(K / 256), dim3(1024), 0, 0, a); }},
        {"quad B6 prio", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((quad<6, 1>), dim3(K / 256), dim3(1024), 0, 0, a); }},
        {"quad B2", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((quad<2, 0>), dim3(K / 256), dim3(1024), 0, 0, a); }},
        {"split NP4 R1", [](const Args& a, uint64_t K) { hipLaunchKernelGGL((split<4, 1, 0>), dim3(K / 64), dim3(320), 0, 0, a); }},
        {"phased NW16 CW4 B16", [](const Args& a, uint64_t K) { launch_phased<16, 4, 16>(a, K); }},
        {"phased NW16 CW4 B24", [](const Args& a, uint64_t K) { launch_phased<16, 4, 24>(a, K); }},
        {"phased NW16 CW4 B8", [](const Args& a, uint64_t K) { launch_phased<16, 4, 8>(a, K); }},
        {"phased NW8 CW2 B16", [](const Args& a, uint64_t K) { launch_phased<8, 2, 16>(a, K); }},
        {"phased NW8 CW1 B16", [](const Args& a, uint64_t K) { launch_phased<8, 1, 16>(a, K); }},
        {"phased NW16 CW2 B32", [](const Args& a, uint64_t K) { launch_phased<16, 2, 32>(a, K); }},
    };
    const int nv = sizeof vars / sizeof vars[0];
    // reference output
    reset(ref); bind(ref); vars[0].launch(a, K); CK(hipDeviceSynchronize());
    std::vector<double> tref((size_t)T * K * D), talt((size_t)T * K * D);
    CK(hipMemcpy(tref.data(), ref.traj, tref.size() * 8, hipMemcpyDeviceToHost));
    std::vector<std::vector<float>> times(nv);
    for (int round = 0; round < 7; ++round)
        for (int v = 0; v < nv; ++v) {
            reset(alt); bind(alt);
            CK(hipEventRecord(e0)); vars[v].launch(a, K); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            times[v].push_back(ms);
            if (round == 0) {
                CK(hipMemcpy(talt.data(), alt.traj, talt.size() * 8, hipMemcpyDeviceToHost));
                if (memcmp(talt.data(), tref.data(), talt.size() * 8) != 0) printf("!! %s differs from fused\n", vars[v].name);
            }
        }
    for (int v = 0; v < nv; ++v) {
        std::sort(times[v].begin(), times[v].end());
        float med = times[v][times[v].size() / 2];
        printf("%-22s median %.3f ms  min %.3f ms  -> %.3e steps/s\n", vars[v].name, med, times[v][0], (double)K * T / (med * 1e-3));
    }
    return 0;
}
