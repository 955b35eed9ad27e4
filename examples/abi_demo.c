/* abi_demo.c -- the C ABI of librunlmc_hip.so used from plain C (no Python, no
 * torch): the README workload of the reference (D = 2 outputs, one RBF kernel,
 * m = 100 grid points) with the data sitting on the grid points.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude \
 *       examples/abi_demo.c -o abi_demo \
 *       -Lrunlmc_amd/csrc -lrunlmc_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/runlmc_amd/csrc -Wl,-rpath,/opt/rocm/lib
 *
 * Checks, against a dense O(m^2) Toeplitz product computed right here:
 *   1. rl_gridop_mvm:   K_UU x,  K_UU = B (x) T   (kronecker.py:39-46, bttb.py:144-148)
 *   2. rl_ski_mvm:      (W K_UU W^T + diag(eps)) x   (ski.py:13-16, grid_kernel.py:70-74)
 *   3. rl_solve_batch:  MINRES solve, residual recomputed on the host
 *                       (iterative.py:23-62)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include "runlmc_hip.h"

#define D 2
#define M 100
#define N (D * M)

#define CHECK_RL(call)                                                        \
    do {                                                                      \
        int rc_ = (call);                                                     \
        if (rc_ != RL_OK) {                                                   \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rl_last_error()); \
            return 1;                                                         \
        }                                                                     \
    } while (0)
#define CHECK_HIP(call)                                                       \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));        \
            return 1;                                                         \
        }                                                                     \
    } while (0)

/* y = (B (x) T) x with a dense Toeplitz T (first row `top`) */
static void dense_kron_toeplitz(const double B[D][D], const double* top, const double* x,
                                double* y) {
    for (int a = 0; a < D; ++a)
        for (int i = 0; i < M; ++i) {
            double acc = 0.0;
            for (int b = 0; b < D; ++b)
                for (int j = 0; j < M; ++j)
                    acc += B[a][b] * top[abs(i - j)] * x[b * M + j];
            y[a * M + i] = acc;
        }
}

static double max_rel_err(const double* got, const double* ref, int n) {
    double e = 0.0, s = 0.0;
    for (int i = 0; i < n; ++i) {
        if (fabs(got[i] - ref[i]) > e) e = fabs(got[i] - ref[i]);
        if (fabs(ref[i]) > s) s = fabs(ref[i]);
    }
    return e / s;
}

int main(void) {
    int ndev = 0;
    if (rl_abi_version() != RL_ABI_VERSION) {
        fprintf(stderr, "library implements ABI %d, header declares %d\n", rl_abi_version(),
                RL_ABI_VERSION);
        return 2;
    }
    CHECK_RL(rl_device_count(&ndev));
    printf("backend %s, %d device(s)\n", rl_backend(), ndev);

    /* one RBF kernel on a unit-spaced grid, B = a a^T + diag(kappa) */
    double top[M], a[D] = {1.0, -0.5}, kappa[D] = {0.3, 0.2}, noise[D] = {0.5, 0.4};
    for (int i = 0; i < M; ++i) top[i] = exp(-0.5 * (0.15 * i) * (0.15 * i));
    double B[D][D];
    for (int p = 0; p < D; ++p)
        for (int q = 0; q < D; ++q) B[p][q] = a[p] * a[q] + (p == q ? kappa[p] : 0.0);
    int ranks[1] = {1}, lens[D] = {M, M};

    rl_gridop* g = NULL;
    CHECK_RL(rl_gridop_create(0, D, M, 1, &g));
    CHECK_RL(rl_gridop_set_lmc(g, 1, top, ranks, a, kappa));
    /* which form the top row's products take (0 transform kernels, 1 polynomial, 2 filter;
     * on a grid this short every batch stays on the transform kernels) */
    int forms[1] = {-1}, structured = -1;
    CHECK_RL(rl_gridop_top_forms(g, forms, &structured));
    printf("top-row form %d, structured %d\n", forms[0], structured);

    /* data on the grid points: W = identity (CSR), so K~ = K_UU + diag(eps) */
    int indptr[N + 1], indices[N];
    double ones[N];
    for (int i = 0; i < N; ++i) { indptr[i] = i; indices[i] = i; ones[i] = 1.0; }
    indptr[N] = N;
    rl_ski* s = NULL;
    CHECK_RL(rl_ski_create(g, N, indptr, indices, ones, indptr, indices, ones, &s));
    CHECK_RL(rl_ski_set_noise(s, noise, lens));

    double x[N], ref[N], got[N];
    srand(7);
    for (int i = 0; i < N; ++i) x[i] = rand() / (double)RAND_MAX - 0.5;
    double *dx = NULL, *dy = NULL;
    CHECK_HIP(hipMalloc((void**)&dx, sizeof(x)));
    CHECK_HIP(hipMalloc((void**)&dy, sizeof(x)));
    CHECK_HIP(hipMemcpy(dx, x, sizeof(x), hipMemcpyHostToDevice));

    /* 1. grid operator */
    CHECK_RL(rl_gridop_mvm(g, dx, dy, 1, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(got, dy, sizeof(x), hipMemcpyDeviceToHost));
    dense_kron_toeplitz(B, top, x, ref);
    const double e1 = max_rel_err(got, ref, N);

    /* 2. SKI operator */
    CHECK_RL(rl_ski_mvm(s, dx, dy, 1, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(got, dy, sizeof(x), hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i) ref[i] += noise[i / M] * x[i];
    const double e2 = max_rel_err(got, ref, N);

    /* 3. solve K~ z = x, residual on the host */
    int iters = 0, istop = 0;
    double resid = 0.0, z[N];
    CHECK_RL(rl_solve_batch(s, dx, dy, 1, 0 /* MINRES */, 1e-8, 100, 0, &iters, &resid, &istop,
                            NULL));
    CHECK_HIP(hipMemcpy(z, dy, sizeof(x), hipMemcpyDeviceToHost));
    dense_kron_toeplitz(B, top, z, ref);
    double r2 = 0.0;
    for (int i = 0; i < N; ++i) {
        const double r = x[i] - ref[i] - noise[i / M] * z[i];
        r2 += r * r;
    }
    printf("grid mvm rel err %.2e | ski mvm rel err %.2e | minres %d iterations, istop %d, "
           "reported residual %.2e, host residual %.2e\n", e1, e2, iters, istop, resid, sqrt(r2));

    CHECK_HIP(hipFree(dx));
    CHECK_HIP(hipFree(dy));
    CHECK_RL(rl_ski_destroy(s));
    CHECK_RL(rl_gridop_destroy(g));
    if (e1 > 1e-11 || e2 > 1e-11 || sqrt(r2) > 1e-7 || fabs(sqrt(r2) - resid) > 1e-9) {
        fprintf(stderr, "abi_demo FAILED\n");
        return 1;
    }
    printf("abi_demo ok\n");
    return 0;
}
