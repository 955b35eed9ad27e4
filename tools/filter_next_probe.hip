// Scratch micro-benchmarks for the NEXT design of the recursive-filter product (DESIGN.md section 10b:
// the tile kept in registers across the look-back wait, phase A on the matrix cores).  Three
// questions the design rests on, measured instead of guessed (gfx950):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/filter_next_probe tools/filter_next_probe.hip && /tmp/filter_next_probe
//  1. Do fp64 matrix instructions (v_mfma_f64_16x16x4) and fp64 vector instructions overlap on one SIMD,
//     and at what combined rate?  (k_mix: per round M matrix instructions on independent accumulators
//     interleaved with V vector multiply-adds on independent chains; 1 or 2 waves per SIMD)
//  2. At what rate does a wave read x in the matrix-core B layout straight from global memory
//     (lane (segment n = l % 16, k = l / 16) reads the 16 bytes at points 8 j + 2 k of segment n:
//     16 x 64-byte pieces per instruction) against the plain coalesced 16-byte stream?
//  3. What does one workgroup-to-workgroup hand-off cost inside a launch under streaming load
//     (producer: sc1 payload of 220 doubles + flag; consumer: poll, acquire, read) -- the look-back
//     hop of a one-pass scan?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---- 1. matrix || vector ---------------------------------------------------------------
template <int M, int V>
__global__ void __launch_bounds__(512) k_mix(double* out, int rounds, double a, double b) {
    double4_t acc[4];
    double v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{a, b, a, b};
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    const double x = a + threadIdx.x * 1e-9, y = b;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (M > 0) {
#pragma unroll
                for (int i = 0; i < M; ++i) acc[(u * M + i) & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[(u * M + i) & 3], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < V; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(a), "v"(b));
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

template <int M, int V>
static void run_mix(int waves_per_simd) {
    double* out; hipMalloc(&out, 8);
    const int rounds = 4000, threads = 256 * waves_per_simd, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<M, V>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<M, V>), dim3(blocks), dim3(threads), 0, 0, out, rounds, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_round_ns = ms * 1e6 / ((double)rounds * 4 * waves_per_simd);   // per wave and group of (M, V)
    printf("matrix %d + vector %2d per group, %d wave(s)/SIMD: %7.1f ns per group and wave = %6.2f ns per matrix instr, %5.2f ns per vector instr (if alone)\n",
           M, V, waves_per_simd, per_round_ns, M ? per_round_ns / M : 0.0, V ? per_round_ns / V : 0.0);
    hipFree(out);
}

// the same question with the roles on DIFFERENT waves of one SIMD: a 512-thread workgroup puts waves
// w and w + 4 on the same SIMD; waves 0-3 issue matrix instructions only, waves 4-7 vector only
// (mode 0: both, 1: the matrix waves alone (the others exit), 2: the vector waves alone)
__global__ void __launch_bounds__(512) k_split(double* out, int rounds, double a, double b, int mode) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave < 4) {
        if (mode == 2) return;
        double4_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = double4_t{a, b, a, b};
        const double x = a + threadIdx.x * 1e-9, y = b;
        for (int r = 0; r < rounds; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        if (mode == 1) return;
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
        for (int r = 0; r < rounds; ++r)
#pragma unroll
            for (int u = 0; u < 7; ++u)          // 4 x 32 ns of matrix work ~ 56 vector instructions
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    if (s == 12345.678) out[0] = s;
}
static void run_split() {
    double* out; hipMalloc(&out, 8);
    const int rounds = 4000;
    const char* what[3] = {"matrix waves and vector waves together", "matrix waves alone", "vector waves alone"};
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_split, dim3(256), dim3(512), 0, 0, out, 10, 1.0000001, 1e-9, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_split, dim3(256), dim3(512), 0, 0, out, rounds, 1.0000001, 1e-9, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("roles on different waves of a SIMD, %-40s %8.1f ns per round (4 matrix instructions | 56 vector instructions)\n",
               what[mode], ms * 1e6 / rounds);
    }
    hipFree(out);
}

// ---- 2. B-layout reads -------------------------------------------------------------------
struct __attribute__((packed, aligned(8))) Pair { double a, b; };
// rows of 512 points; a wave reads 4 rows per tile; mode 0: coalesced (lane l: pair 2 l + 128 j),
// mode 1: B layout (lane (n, k): pair at 32 n + 8 j + 2 k, j < 4)
template <int MODE>
__global__ void __launch_bounds__(256) k_read(const double* __restrict__ X, size_t nrows, double* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, k = lane >> 4;
    double s = 0.0;
    for (size_t r0 = ((size_t)blockIdx.x * 4 + wave) * 4; r0 + 4 <= nrows; r0 += (size_t)gridDim.x * 16) {
        Pair p[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int off = MODE == 0 ? 2 * lane + 128 * j : 32 * n + 8 * j + 2 * k;
                p[r][j] = *reinterpret_cast<const Pair*>(X + (r0 + r) * 512 + off);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += p[r][j].a + p[r][j].b;
    }
    if (s == 12345.678) out[0] = s;
}
template <int MODE>
static void run_read(const double* X, size_t nrows, double* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_read<MODE>), dim3(2048), dim3(256), 0, 0, X, nrows, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_read<MODE>), dim3(2048), dim3(256), 0, 0, X, nrows, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("read of %.2f GB, %s: %.1f us = %.2f TB/s\n", nrows * 4096.0 / 1e9,
           MODE == 0 ? "coalesced 16-byte stream      " : "matrix-core B layout (64 B pieces)", ms * 1e3 / 5, nrows * 4096.0 / (ms / 5 * 1e-3) / 1e12);
}

// ---- 3. hand-off latency under load ---------------------------------------------------------
// workgroup 2 i produces for workgroup 2 i + 1, HOPS times in a row (ping only: the consumer stamps the
// time from the producer's stamp to the payload being readable); the other workgroups stream memory
__global__ void __launch_bounds__(256) k_handoff(double* payload, unsigned long long* flags, long long* stamps,
                                                 const double* __restrict__ X, size_t nx, int hops, int pairs, double* out) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= 2 * pairs) {                      // background load: stream X
        double s = 0.0;
        for (int rep = 0; rep < 4; ++rep)
            for (size_t i = (size_t)(b - 2 * pairs) * 256 + tid; i < nx; i += (size_t)(gridDim.x - 2 * pairs) * 256) s += X[i];
        if (s == 12345.678) out[0] = s;
        return;
    }
    const int pr = b >> 1;
    double* pay = payload + (size_t)pr * 256;
    unsigned long long* flag = flags + pr * 16;
    if ((b & 1) == 0) {                        // producer
        for (int h = 1; h <= hops; ++h) {
            // pace: wait until the consumer has acknowledged the previous hop
            // (every spin is bounded: a protocol error ends the kernel, it does not hang the GPU)
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(flag + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)(h - 1) && ++spins < 2000000) __builtin_amdgcn_s_sleep(2);
            }
            __syncthreads();
            const long long t0 = wall_clock64();
            if (tid < 220) pay[tid] = (double)h + tid;
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                stamps[(size_t)pr * hops * 2 + 2 * (h - 1)] = t0;
                __hip_atomic_store(flag, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {                                   // consumer
        double s = 0.0;
        for (int h = 1; h <= hops; ++h) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)h && ++spins < 2000000) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            if (tid < 220) s += pay[tid];
            __syncthreads();
            if (tid == 0) {
                stamps[(size_t)pr * hops * 2 + 2 * (h - 1) + 1] = wall_clock64();
                __hip_atomic_store(flag + 8, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (s == 12345.678) out[0] = s;
    }
}

int main() {
    printf("== 1. fp64 matrix instructions next to fp64 vector instructions (256 workgroups)\n");
    run_mix<0, 16>(1); run_mix<0, 16>(2);
    run_mix<1, 0>(1); run_mix<1, 0>(2);
    run_mix<1, 8>(1); run_mix<1, 8>(2);
    run_mix<1, 16>(1); run_mix<1, 16>(2);
    run_mix<1, 24>(2); run_mix<2, 16>(2);
    run_split();
    printf("== 2. reading x: plain stream against the matrix-core B layout\n");
    const size_t nrows = 129 * 10 * 196;                    // C5: 129 vectors x 10 rows x 196 chunks of 512
    double *X, *out; hipMalloc(&X, nrows * 4096); hipMalloc(&out, 8);
    hipMemset(X, 0, nrows * 4096);
    run_read<0>(X, nrows, out); run_read<1>(X, nrows, out);
    printf("== 3. workgroup-to-workgroup hand-off of 220 doubles (plain stores + release fence + flag; poll + acquire), 64 pairs, the rest of 512 workgroups streaming\n");
    const int pairs = 64, hops = 50;
    double* payload; unsigned long long* flags; long long* stamps;
    hipMalloc(&payload, pairs * 256 * 8); hipMalloc(&flags, pairs * 16 * 8); hipMalloc(&stamps, (size_t)pairs * hops * 2 * 8);
    hipMemset(flags, 0, pairs * 16 * 8); hipMemset(stamps, 0, (size_t)pairs * hops * 2 * 8);
    hipLaunchKernelGGL(k_handoff, dim3(512), dim3(256), 0, 0, payload, flags, stamps, X, nrows * 512, hops, pairs, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("hand-off kernel failed\n"); return 1; }
    std::vector<long long> hs((size_t)pairs * hops * 2);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> lat;
    for (int p = 0; p < pairs; ++p)
        for (int h = 5; h < hops; ++h) lat.push_back((hs[((size_t)p * hops + h) * 2 + 1] - hs[((size_t)p * hops + h) * 2]) * 0.01);   // 100 MHz ticks -> us
    std::sort(lat.begin(), lat.end());
    printf("hand-off (producer's first store -> consumer has read the payload): median %.2f us, 10 %% %.2f, 90 %% %.2f, max %.2f (wall_clock64 at 100 MHz)\n",
           lat[lat.size() / 2], lat[lat.size() / 10], lat[lat.size() * 9 / 10], lat.back());
    return 0;
}
