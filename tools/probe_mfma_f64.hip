// Probe: is v_mfma_f64_16x16x4_f64 bit-identical to a k-ordered fma chain?
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off probe_mfma_f64.hip -o probe_mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k(const double* A, const double* B, const double* C, double* D, int ksteps)
{
    // A: [16][4*ksteps] row-major (i,k); B: [4*ksteps][16] (k,j); C,D: [16][16]
    int l = threadIdx.x, K = 4 * ksteps;
    d4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    for (int s = 0; s < ksteps; ++s) {
        double a = A[(l & 15) * K + 4 * s + (l >> 4)];
        double b = B[(4 * s + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

int main()
{
    const int ksteps = 32, K = 4 * ksteps;
    std::vector<double> A(16 * K), B(K * 16), C(256), D(256);
    srand(7);
    auto rnd = [] { return (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 40 - 20) * 0.3); };
    long bad_chain = 0, bad_pair = 0, bad_rev = 0, total = 0;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    for (int trial = 0; trial < 200; ++trial) {
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd();
        for (auto& v : C) v = (trial % 2) ? rnd() : 0.0;
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, ksteps);
        hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double c1 = C[i * 16 + j], c2 = c1, c3 = c1;
                for (int kk = 0; kk < K; ++kk) c1 = fma(A[i * K + kk], B[kk * 16 + j], c1);
                for (int s = 0; s < ksteps; ++s) {  // per MFMA: sum of 4 products then add
                    double p = 0;
                    for (int q = 0; q < 4; ++q) p = fma(A[i * K + 4 * s + q], B[(4 * s + q) * 16 + j], p);
                    c2 += p;
                    for (int q = 3; q >= 0; --q) c3 = fma(A[i * K + 4 * s + q], B[(4 * s + q) * 16 + j], c3);
                }
                double d = D[i * 16 + j];
                ++total;
                if (d != c1) ++bad_chain;
                if (d != c2) ++bad_pair;
                if (d != c3) ++bad_rev;
            }
    }
    printf("total %ld  mismatch vs k-ordered fma chain: %ld  vs (sum4 then add): %ld  vs reversed chain: %ld\n", total,
           bad_chain, bad_pair, bad_rev);
    return 0;
}
