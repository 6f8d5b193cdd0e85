/* A client of libymc.so written against include/ymc.h alone (plain C99, no Python, no HIP headers):
 * the 2-D Gaussian random-walk ensemble of example_mcmc_2d_singleLevel.py.  Prints the FNV-1a hash
 * of the trajectory and decision records plus the pooled acceptance rate; tests/test_gpu_c_abi.py
 * compares them with the same run driven through the ctypes layer.
 *   gcc -std=c99 -O2 -I include tests/c_abi/smoke.c -o smoke -L yagre-mcmc_amd -lymc -Wl,-rpath,...  */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ymc.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != YMC_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ymc_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static uint64_t fnv1a(const void *p, size_t n, uint64_t h)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

int main(int argc, char **argv)
{
    const uint64_t K = argc > 1 ? strtoull(argv[1], NULL, 10) : 4096;
    const uint64_t T = argc > 2 ? strtoull(argv[2], NULL, 10) : 200;
    if (ymc_abi_version() != YMC_ABI_VERSION) {
        fprintf(stderr, "ABI mismatch: header %d, library %d\n", YMC_ABI_VERSION, ymc_abi_version());
        return 1;
    }
    /* mean, whitening matrix and log-normaliser of N((1, 1.5), [[2.4, -0.5], [-0.5, 0.7]]) come from the caller */
    double mean[2] = {1.0, 1.5};
    double whiten[4], log_norm;
    if (argc < 8) {
        fprintf(stderr, "usage: smoke K T u00 u01 u10 u11 log_norm\n");
        return 2;
    }
    for (int i = 0; i < 4; ++i) whiten[i] = strtod(argv[3 + i], NULL);
    log_norm = strtod(argv[7], NULL);

    ymc_ctx *ctx = NULL;
    ymc_problem *prob = NULL;
    ymc_run *run = NULL;
    CHECK(ymc_ctx_create(0, &ctx));

    ymc_problem_desc pd;
    memset(&pd, 0, sizeof pd);
    pd.method = YMC_METHOD_MRW;
    pd.state_eq = YMC_EQ_EXACT;
    pd.target.kind = YMC_DENSITY_GAUSS_WHITENED;
    pd.target.dim = 2;
    pd.target.mean = mean;
    pd.target.whiten = whiten;
    pd.target.log_norm = log_norm;
    double sigma[2] = {1.0, 1.0};
    pd.proposal_sigma = sigma;
    CHECK(ymc_problem_create(ctx, &pd, &prob));

    double x0[2] = {-8.0, -7.0};
    ymc_run_desc rd;
    memset(&rd, 0, sizeof rd);
    rd.seed = 20250103;
    rd.n_chains = K;
    rd.initial_state = x0;
    rd.x0_rows = 1;
    rd.record_capacity = T;
    rd.record_flags = YMC_RECORD_TRAJECTORY | YMC_RECORD_ACCEPT | YMC_RECORD_MOMENTS;
    CHECK(ymc_run_create(ctx, prob, &rd, &run));
    CHECK(ymc_run_advance(run, T / 2));
    CHECK(ymc_run_advance(run, T - T / 2));
    CHECK(ymc_run_sync(run));

    const size_t tb = (size_t)T * K * 2 * sizeof(double), ab = (size_t)T * K;
    double *traj = (double *)malloc(tb);
    uint8_t *acc = (uint8_t *)malloc(ab);
    uint64_t *accepted = (uint64_t *)malloc(K * sizeof(uint64_t));
    if (!traj || !acc || !accepted) return 3;
    CHECK(ymc_run_fetch_trajectory(run, 0, T, traj, tb));
    CHECK(ymc_run_fetch_accept(run, 0, T, acc, ab));
    CHECK(ymc_run_fetch_summary(run, accepted, NULL, NULL, NULL, NULL));
    uint64_t total = 0;
    for (uint64_t k = 0; k < K; ++k) total += accepted[k];
    uint64_t h = fnv1a(traj, tb, 14695981039346656037ull);
    h = fnv1a(acc, ab, h);
    printf("%016llx %.17g\n", (unsigned long long)h, (double)total / (double)(K * T));

    /* a record that is full must be refused, not overrun */
    if (ymc_run_advance(run, 1) != YMC_ERR_CAPACITY) {
        fprintf(stderr, "advance past the record capacity was not refused\n");
        return 4;
    }
    free(traj);
    free(acc);
    free(accepted);
    CHECK(ymc_run_destroy(run));
    CHECK(ymc_problem_destroy(prob));
    CHECK(ymc_ctx_destroy(ctx));
    return 0;
}
