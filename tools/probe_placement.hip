// Probe: where does the hardware place the workgroups of a 256- and a 512-workgroup launch (256 threads each,
// ~170 VGPRs, 4 KB LDS -- the shape of the random-walk kernel at 65 536 and 131 072 chains)?
// Build: hipcc --offload-arch=gfx950 -O3 probe_placement.hip -o probe_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k(unsigned *out, int spin, double seed)
{
    __shared__ double lds[512];
    lds[threadIdx.x] = seed;
    __syncthreads();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep ~170 VGPRs alive
    double r[80];
#pragma unroll
    for (int i = 0; i < 80; ++i) r[i] = lds[(threadIdx.x + i) & 511] + i;
    for (int it = 0; it < spin; ++it) {
#pragma unroll
        for (int i = 0; i < 80; ++i) r[i] = __builtin_fma(r[i], 1.0000001, r[(i + 1) % 80] * 1e-9);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 80; ++i) s += r[i];
    if (s == 12345.678) out[0] = 1;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = hwid;
        out[2 * w + 1] = xcc;
    }
}

int main()
{
    for (int blocks : {256, 512, 1024}) {
        unsigned *d;
        hipMalloc(&d, blocks * 4 * 2 * sizeof(unsigned));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 3000, 1.0);
        hipDeviceSynchronize();
        std::vector<unsigned> h(blocks * 8);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_cu, per_simd;
        for (int w = 0; w < blocks * 4; ++w) {
            const unsigned id = h[2 * w], xcc = h[2 * w + 1] & 0xf;
            const unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 7;
            const unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            per_cu[cukey] += 1;
            per_simd[(cukey << 2) | simd] += 1;
        }
        std::map<int, int> hist_cu, hist_simd;
        for (auto &p : per_cu) hist_cu[p.second] += 1;
        for (auto &p : per_simd) hist_simd[p.second] += 1;
        printf("%d workgroups: %zu distinct CUs; waves per CU histogram:", blocks, per_cu.size());
        for (auto &p : hist_cu) printf(" %d waves x %d CUs;", p.first, p.second);
        printf("  waves per SIMD:");
        for (auto &p : hist_simd) printf(" %d x %d;", p.first, p.second);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
