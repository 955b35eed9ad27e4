// Scratch micro-benchmarks: kernel boundary cost, dependent-load latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty() {}
__global__ void k_touch(double* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0; }
__global__ void k_chain(const int* __restrict__ next, int steps, int* out) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    for (int s = 0; s < steps; ++s) i = next[i];
    if (i == -1) out[0] = i;
}
__global__ void k_stream(const double* __restrict__ a, double* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i] * 2.0;
}
static float run(hipStream_t st, int reps, void (*f)(hipStream_t)) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f(st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f(st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / reps;
}
static double* g_p; static int* g_next; static int* g_out; static double *g_a, *g_b; static int g_steps; static int g_blocks;
int main() {
    hipStream_t st; hipStreamCreate(&st);
    hipMalloc(&g_p, 8); hipMalloc(&g_out, 4);
    const int N = 1 << 22;
    std::vector<int> h(N); for (int i = 0; i < N; ++i) h[i] = (int)(((long)i * 1664525L + 1013904223L) % N);
    hipMalloc(&g_next, N * 4); hipMemcpy(g_next, h.data(), N * 4, hipMemcpyHostToDevice);
    hipMalloc(&g_a, (size_t)N * 8 * 8); hipMalloc(&g_b, (size_t)N * 8 * 8);
    printf("empty kernel, back-to-back: %.2f us\n", run(st, 2000, [](hipStream_t s){ k_empty<<<1, 64, 0, s>>>(); }));
    printf("empty kernel 288x256:       %.2f us\n", run(st, 2000, [](hipStream_t s){ k_empty<<<288, 256, 0, s>>>(); }));
    printf("dependent touch kernels:    %.2f us\n", run(st, 2000, [](hipStream_t s){ k_touch<<<1, 64, 0, s>>>(g_p); }));
    for (int steps : {1, 2, 4, 8, 16}) { g_steps = steps;
        for (int blocks : {1, 288}) { g_blocks = blocks;
            printf("chain steps=%2d blocks=%3d:   %.2f us\n", steps, blocks, run(st, 500, [](hipStream_t s){ k_chain<<<g_blocks, 256, 0, s>>>(g_next, g_steps, g_out); })); } }
    for (size_t mb : {1, 8, 64, 256}) { static size_t n; n = mb * 1024 * 1024 / 8;
        float us = run(st, 50, [](hipStream_t s){ k_stream<<<2048, 256, 0, s>>>(g_a, g_b, n); });
        printf("stream copy %4zu MB: %.2f us  %.1f GB/s\n", mb, us, 2.0 * mb * 1.048576e6 / us / 1e3); }
    // graph of 10 dependent tiny kernels
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 10; ++i) k_touch<<<1, 64, 0, st>>>(g_p);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    static hipGraphExec_t sge; sge = ge;
    printf("graph of 10 touch kernels:  %.2f us per kernel\n", run(st, 300, [](hipStream_t s){ hipGraphLaunch(sge, s); }) / 10);
    return 0;
}
