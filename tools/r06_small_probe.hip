// Where do k_lr_small's microseconds go?  The real kernel (rl_lowrank.h) at the C2 launch shape
// with phase stamps (RL_TIMING), timed with events over back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRL_TIMING -Irunlmc_amd/csrc -o /tmp/small_probe tools/r06_small_probe.hip
#define RL_TIMING 1
#include "rl_kernels.h"
#include "rl_lowrank.h"
#include <cstdio>
#include <vector>
#include <algorithm>

template <int R>
static void run(int D, int m, int nvec) {
    double *X, *Y, *beta, *Mf;
    const size_t ve = (size_t)nvec * D * m;
    hipMalloc(&X, ve * 8);
    hipMalloc(&Y, ve * 8);
    std::vector<double> hx(ve);
    for (size_t i = 0; i < ve; ++i) hx[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    hipMemcpy(X, hx.data(), ve * 8, hipMemcpyHostToDevice);
    std::vector<double> hb(64, 0.25), hm((size_t)D * R * D * R, 1e-3);
    hipMalloc(&beta, 64 * 8);
    hipMemcpy(beta, hb.data(), 64 * 8, hipMemcpyHostToDevice);
    hipMalloc(&Mf, hm.size() * 8);
    hipMemcpy(Mf, hm.data(), hm.size() * 8, hipMemcpyHostToDevice);
    double* part;
    const int nseg = lr_small_nseg(m);
    hipMalloc(&part, (size_t)nvec * D * nseg * R * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() {
        hipLaunchKernelGGL((k_lr_small_project<R>), dim3(D * nseg, nvec), dim3(RL_LR_SMALL_WG),
                           lr_small_project_lds(R), 0, (const double*)X, D, m, nseg, (const double*)beta, part);
        hipLaunchKernelGGL((k_lr_small_expand<R>), dim3(D * nseg, nvec), dim3(RL_LR_SMALL_WG),
                           lr_small_expand_lds(D, R), 0, (const double*)part, D, m, nseg, (const double*)beta,
                           (const double*)Mf, Y);
    };
    for (int w = 0; w < 5; ++w) launch();
    hipDeviceSynchronize();
    const int reps = 200;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // each kernel alone
    float msp, mse;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_lr_small_project<R>), dim3(D * nseg, nvec), dim3(RL_LR_SMALL_WG),
                           lr_small_project_lds(R), 0, (const double*)X, D, m, nseg, (const double*)beta, part);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&msp, e0, e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_lr_small_expand<R>), dim3(D * nseg, nvec), dim3(RL_LR_SMALL_WG),
                           lr_small_expand_lds(D, R), 0, (const double*)part, D, m, nseg, (const double*)beta,
                           (const double*)Mf, Y);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&mse, e0, e1);
    printf("D %d m %d nvec %d rank %d nseg %d (%d workgroups): %.2f us per product (two launches) | projection "
           "alone %.2f us per launch | expansion alone %.2f us per launch\n",
           D, m, nvec, R, nseg, D * nseg * nvec, ms / reps * 1e3, msp / reps * 1e3, mse / reps * 1e3);
    hipFree(part);
    hipFree(X); hipFree(Y); hipFree(beta); hipFree(Mf);
}

int main() {
    run<24>(4, 5004, 17);
    run<24>(4, 5004, 1);
    run<24>(4, 5004, 64);
    run<24>(10, 1000, 8);
    run<36>(4, 5004, 17);
    return 0;
}
