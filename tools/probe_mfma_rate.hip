// Probe: what a 16-chain wave's forward map (256 x v_mfma_f64_16x16x4_f64, A fragments from LDS) costs per MFMA at one and two
// waves per SIMD, against the matrix pipe's 64 cycles -- and which instruction order gets closest:
//   mode 0  the product's loop as the compiler schedules it (ymc_tile_residual_sum: each fragment read one MFMA before its use)
//   mode 1  no LDS at all (the A operand is a register): the pipe's own ceiling
//   mode 2  hand-scheduled blocks of 16 MFMAs, the fragments of block b + 1 read while block b runs (inline asm)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_rate.hip -o tools/probe_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int KS = 32, MT = 8;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(unsigned long long *out, double *sink, int reps)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < MT * KS * 64; i += 64 * WAVES) lds[i] = 1.0 + 1e-9 * (i % 977);
    __syncthreads();
    double v[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) v[s] = 0.5 + 1e-6 * (lane + s);
    double total = 0.0;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
#pragma unroll 1
    for (int rep = 0; rep < reps; ++rep) {
        if constexpr (MODE == 3 || MODE == 4 || MODE == 5) {
            // 3: A from eight different registers, no LDS; 4: the same plus one (unused) ds_read_b64 per MFMA;
            // 5: A from LDS by single ds_read_b64 (no read2 pairing), two k-steps ahead
            const unsigned base = (unsigned)(size_t)(lds) + lane * 8;
#pragma unroll 1
            for (int mt = 0; mt < MT; mt += 2) {
                d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
                double ar[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) ar[i] = v[(mt + i) & 31] * 1.0001;
                const unsigned pa = base + (unsigned)mt * KS * 512;
                if constexpr (MODE == 5) {
                    double f0[KS], f1[KS];
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f0[s]) : "v"(pa), "n"(s * 512));
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f1[s]) : "v"(pa), "n"((KS + s) * 512));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        asm volatile("" : "+v"(f0[s]), "+v"(f1[s]));
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0[s], v[s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1[s], v[s], acc1, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        if constexpr (MODE == 4) {
                            double junk;
                            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(junk) : "v"(pa), "n"(s * 512));
                            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(junk) : "v"(pa), "n"((KS + s) * 512));
                        }
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[s & 7], v[s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[(s + 3) & 7], v[s], acc1, 0, 0, 0);
                    }
                    if constexpr (MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                total += (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]) + (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
            }
        } else if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll 1
            for (int mt = 0; mt < MT; mt += 2) {
                d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
                const double *A0 = lds + (size_t)mt * KS * 64 + lane, *A1 = A0 + (size_t)KS * 64;
                double areg = v[mt];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(MODE == 1 ? areg : A0[s * 64], v[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(MODE == 1 ? areg : A1[s * 64], v[s], acc1, 0, 0, 0);
                }
                total += (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]) + (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
            }
        } else {
            // hand-scheduled: per pair of tiles 4 blocks of 8 k-steps; fragments of the next block are read while the
            // current block's 16 MFMAs run.  f[b][t][i]: block parity b, tile t, two k-steps per d2
            const unsigned base = (unsigned)(size_t)(lds) + lane * 8;   // LDS byte address of this lane's column
            d2 fa0[4], fa1[4], fb0[4], fb1[4];
            auto read8 = [&](d2 (&f0)[4], d2 (&f1)[4], unsigned addr) {
                asm volatile("ds_read2st64_b64 %0, %8 offset0:0 offset1:1\n\t"
                             "ds_read2st64_b64 %4, %8 offset0:32 offset1:33\n\t"
                             "ds_read2st64_b64 %1, %8 offset0:2 offset1:3\n\t"
                             "ds_read2st64_b64 %5, %8 offset0:34 offset1:35\n\t"
                             "ds_read2st64_b64 %2, %8 offset0:4 offset1:5\n\t"
                             "ds_read2st64_b64 %6, %8 offset0:36 offset1:37\n\t"
                             "ds_read2st64_b64 %3, %8 offset0:6 offset1:7\n\t"
                             "ds_read2st64_b64 %7, %8 offset0:38 offset1:39"
                             : "=&v"(f0[0]), "=&v"(f0[1]), "=&v"(f0[2]), "=&v"(f0[3]), "=&v"(f1[0]), "=&v"(f1[1]), "=&v"(f1[2]),
                               "=&v"(f1[3])
                             : "v"(addr));
            };
            // 16 MFMAs on fragments (f0, f1) with B operands vv[0..7]; waits for the reads issued BEFORE the previous
            // read8 (lgkmcnt counts in order: 8 newer reads may stay in flight)
            auto mfma16 = [&](d4 &acc0, d4 &acc1, const d2 (&f0)[4], const d2 (&f1)[4], const double *vv, bool last) {
                if (last) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0[i][0], vv[2 * i], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1[i][0], vv[2 * i], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0[i][1], vv[2 * i + 1], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1[i][1], vv[2 * i + 1], acc1, 0, 0, 0);
                }
            };
            read8(fa0, fa1, base);
#pragma unroll
            for (int p = 0; p < MT / 2; ++p) {
                d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
                const unsigned pa = base + (unsigned)(2 * p) * KS * 512;
                read8(fb0, fb1, pa + 8 * 512);
                mfma16(acc0, acc1, fa0, fa1, v + 0, false);
                read8(fa0, fa1, pa + 16 * 512);
                mfma16(acc0, acc1, fb0, fb1, v + 8, false);
                read8(fb0, fb1, pa + 24 * 512);
                mfma16(acc0, acc1, fa0, fa1, v + 16, false);
                if (p + 1 < MT / 2) read8(fa0, fa1, pa + 2 * KS * 512);
                mfma16(acc0, acc1, fb0, fb1, v + 24, p + 1 >= MT / 2);
                total += (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]) + (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1));
    if (lane == 0) {
        out[2 * (blockIdx.x * WAVES + (tid >> 6))] = t1 - t0;
        out[2 * (blockIdx.x * WAVES + (tid >> 6)) + 1] = r1 - r0;
    }
    if (total == 12345.678) sink[0] = total;
}

template <int MODE, int WAVES>
int run(const char *name)
{
    const int blocks = 256, reps = 200;
    unsigned long long *d;
    double *sink;
    CHECK(hipMalloc(&d, sizeof(unsigned long long) * 2 * blocks * WAVES));
    CHECK(hipMalloc(&sink, 8));
    const size_t lds = (size_t)MT * KS * 64 * 8;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), lds, 0, d, sink, reps);
        hipEventRecord(e1);
        CHECK(hipEventSynchronize(e1));
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> h(2 * blocks * WAVES);
    CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int i = 0; i < blocks * WAVES; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
    cyc /= blocks * WAVES; real /= blocks * WAVES;
    const double mfmas_per_wave = 256.0 * reps, waves_per_simd = WAVES / 4.0;
    const double clock_ghz = cyc / (real * 10.0);   // s_memrealtime ticks at 100 MHz
    printf("%-34s waves/SIMD %.0f  wave cycles per MFMA %.1f  SIMD cycles per MFMA %.1f  (clock %.2f GHz, kernel %.3f ms)\n", name,
           waves_per_simd, cyc / mfmas_per_wave, cyc / mfmas_per_wave / waves_per_simd, clock_ghz, best);
    hipFree(d);
    hipFree(sink);
    return 0;
}

int main()
{
    if (run<1, 4>("registers only")) return 1;
    if (run<1, 8>("registers only")) return 1;
    if (run<3, 4>("registers, 8 different A")) return 1;
    if (run<3, 8>("registers, 8 different A")) return 1;
    if (run<4, 4>("registers + unused ds_read_b64")) return 1;
    if (run<4, 8>("registers + unused ds_read_b64")) return 1;
    if (run<5, 4>("LDS, whole pair read first (b64)")) return 1;
    if (run<5, 8>("LDS, whole pair read first (b64)")) return 1;
    if (run<0, 4>("LDS fragments, compiler order")) return 1;
    if (run<0, 8>("LDS fragments, compiler order")) return 1;
    if (run<2, 4>("LDS fragments, one block ahead")) return 1;
    if (run<2, 8>("LDS fragments, one block ahead")) return 1;
    return 0;
}
