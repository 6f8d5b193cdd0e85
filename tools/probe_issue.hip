// Probe: issue cost (cycles per wave-instruction) of the VALU ops the MH kernels are made of, at 1..8
// waves per SIMD, and the clock the chip holds under an fp64 VALU load.
// Build: hipcc --offload-arch=gfx950 -O2 probe_issue.hip -o probe_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

template <int OP>
__global__ void k(unsigned long long* out, int iters, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 0.5;
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned long long m0 = threadIdx.x, m1 = 5, m2 = 7, m3 = 9;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { // 4 independent fp64 fma chains
            REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (OP == 1) { // one dependent fp64 fma chain
            REP16(asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));)
        } else if (OP == 2) { // 4 independent xor
            REP16(asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x3 | 1));)
        } else if (OP == 3) { // dependent xor
            REP16(asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(x0) : "v"(x1));)
        } else if (OP == 4) { // 4 independent mad_u64_u32
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\n\tv_mad_u64_u32 %1, vcc, %4, %5, 0\n\tv_mad_u64_u32 %2, vcc, %4, %5, 0\n\tv_mad_u64_u32 %3, vcc, %4, %5, 0" : "=v"(m0), "=v"(m1), "=v"(m2), "=v"(m3) : "v"(x0), "v"(x1) : "vcc");)
        } else if (OP == 5) { // dependent mad_u64_u32 (lo word feeds next)
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m0) : "v"(x0), "v"(x1) : "vcc");)
        } else if (OP == 6) { // 4 independent fp64 mul
            REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if (OP == 7) { // mixed: fma64, xor, fma64, xor
            REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_xor_b32 %2, %2, %3\n\tv_fma_f64 %1, %1, %4, %5\n\tv_xor_b32 %3, %3, %2" : "+v"(a0), "+v"(a1), "+v"(x0), "+v"(x1) : "v"(b), "v"(c));)
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1));
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (a0 + a1 + a2 + a3 == 12345.678 || x0 + x1 + x2 + x3 == 0xdeadbeef || m0 + m1 + m2 + m3 == 77) out[0] = 0;
}

template <int OP>
void run(const char* name, int ninstr_per_iter)
{
    unsigned long long* d;
    const int cus = 256;
    hipMalloc(&d, 16 * 8192);
    for (int wps : {1, 2, 4, 8}) {
        int blocks = cus * wps;  // 256-thread blocks: 4 waves per block = 1 per SIMD
        int iters = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float wall_ms = 0.f;
        hipEventElapsedTime(&wall_ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, clk;
        for (int b = 0; b < blocks; ++b) {
            cyc.push_back((double)h[2 * b] / ((double)iters * ninstr_per_iter));
            clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);  // GHz (memrealtime = 100 MHz)
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        /* the wall-clock rate of the whole launch: what a SIMD really sustains (the per-wave figure divided by the
         * nominal wave count overstates it when fewer waves than launched are resident together) */
        const double ghz = clk[clk.size() / 2];
        const double wall_cyc = (double)wall_ms * 1e-3 * ghz * 1e9 * (cus * 4.0) / ((double)blocks * 4.0 * iters * ninstr_per_iter);
        printf("%-34s waves/SIMD %d: %.2f cycles per wave-instr (per wave), /waves %.2f, WALL-CLOCK %.2f cycles per instr per SIMD "
               "(%.3g wave-instr/s on the chip), clock %.2f GHz\n",
               name, wps, cyc[cyc.size() / 2], cyc[cyc.size() / 2] / wps, wall_cyc,
               (double)blocks * 4.0 * iters * ninstr_per_iter / ((double)wall_ms * 1e-3), ghz);
    }
    hipFree(d);
}

int main()
{
    run<0>("v_fma_f64 x4 independent", 64);
    run<1>("v_fma_f64 dependent chain", 64);
    run<6>("v_mul_f64 x4 independent", 64);
    run<2>("v_xor_b32 x4 independent", 64);
    run<3>("v_xor_b32 dependent chain", 64);
    run<4>("v_mad_u64_u32 x4 independent", 64);
    run<5>("v_mad_u64_u32 dependent (addend)", 32);
    run<7>("fma64/xor interleaved", 64);
    return 0;
}
