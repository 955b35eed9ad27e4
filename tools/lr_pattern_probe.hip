// Where do the polynomial form's two streaming kernels lose their bandwidth?  The REAL kernels
// (rl_lowrank.h) at the C5 launch shape, with the arithmetic scaled down by the rank template
// parameter (rank 2: the same loads / stores, a twelfth of the multiply-adds) and with row
// lengths that are / are not multiples of a 128-byte line (m = 100004: rows start 32 bytes
// past a line boundary three times out of four):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Irunlmc_amd/csrc -o /tmp/lr_pattern_probe tools/lr_pattern_probe.hip
#include "rl_kernels.h"
#include "rl_lowrank.h"
#include <cstdio>
#include <vector>
#include <algorithm>

template <int R>
static void run(int m, int nrows, int steps) {
    double *X, *part, *beta, *zhat;
    hipMalloc(&X, (size_t)nrows * m * 8);
    hipMemset(X, 0, (size_t)nrows * m * 8);
    const int slots = (m + 1) / 2, chunks = (slots + 64 * steps - 1) / (64 * steps);
    hipMalloc(&part, (size_t)chunks * nrows * R * 8);
    hipMalloc(&beta, 64 * 8);
    hipMemset(beta, 0, 64 * 8);
    hipMalloc(&zhat, (size_t)nrows * R * 8);
    hipMemset(zhat, 0, (size_t)nrows * R * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> tp, te;
    const int nbx = (slots + 255) / 256, rpb = 16;
    for (int rep = 0; rep < 12; ++rep) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_lr_project<R>), dim3(chunks, (nrows + RL_LR_ROWS(R) - 1) / RL_LR_ROWS(R)),
                           dim3(64 * RL_LR_WAVES), (size_t)RL_LR_WAVES * R * 65 * 8, 0, (const double*)X, nrows, m,
                           (const double*)beta, steps, part);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) tp.push_back(ms);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_lr_expand<R, false>), dim3(nbx, (nrows + rpb - 1) / rpb), dim3(256), 0, 0,
                           (const double*)zhat, nrows, m, (const double*)beta, rpb, X);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) te.push_back(ms);
    }
    // ... and each kernel alone, back to back (no write of the other kernel draining meanwhile)
    std::vector<float> tpa, tea;
    for (int rep = 0; rep < 12; ++rep) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_lr_project<R>), dim3(chunks, (nrows + RL_LR_ROWS(R) - 1) / RL_LR_ROWS(R)),
                           dim3(64 * RL_LR_WAVES), (size_t)RL_LR_WAVES * R * 65 * 8, 0, (const double*)X, nrows, m,
                           (const double*)beta, steps, part);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) tpa.push_back(ms);
    }
    for (int rep = 0; rep < 12; ++rep) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_lr_expand<R, false>), dim3(nbx, (nrows + rpb - 1) / rpb), dim3(256), 0, 0,
                           (const double*)zhat, nrows, m, (const double*)beta, rpb, X);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) tea.push_back(ms);
    }
    std::sort(tpa.begin(), tpa.end());
    std::sort(tea.begin(), tea.end());
    printf("   alone: project %6.1f us | expand %6.1f us\n", 1e3 * tpa[tpa.size() / 2], 1e3 * tea[tea.size() / 2]);
    std::sort(tp.begin(), tp.end());
    std::sort(te.begin(), te.end());
    const double gb = (double)nrows * m * 8 / 1e9;
    printf("rank %2d  m %6d  steps %3d (chunks %3d): project %6.1f us %5.2f TB/s | expand %6.1f us %5.2f TB/s\n", R, m,
           steps, chunks, 1e3 * tp[tp.size() / 2], gb / tp[tp.size() / 2], 1e3 * te[te.size() / 2], gb / te[te.size() / 2]);
    hipFree(X); hipFree(part); hipFree(beta); hipFree(zhat);
}

// ---- part 2: a bare reader with the projection's access pattern, one ingredient changed at a time
// wave: RB rows; MIRROR: a point and its mirror per lane-step (ascending + descending stream per
// row) or the whole row ascending; WAVES waves per workgroup; OCC waves per SIMD (attribute);
// rowfast: workgroups numbered with the row block fastest instead of the chunk
template <int RB, bool MIRROR, int WAVES, int OCC, int G>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(OCC)))
k_read(const double* __restrict__ X, int nrows, int m, int steps, int rowfast, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int pbx = blockIdx.x, pby = blockIdx.y;
    if (rowfast) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        pbx = lin / gridDim.y;
        pby = lin - pbx * gridDim.y;
    }
    const int row0 = (pby * WAVES + wave) * RB;
    const int n_begin = pbx * (64 * steps);
    const int span = MIRROR ? (m + 1) / 2 : m;
    const double* xrow[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) xrow[r] = X + (size_t)(row0 + r < nrows ? row0 + r : nrows - 1) * m;
    const unsigned rowbytes = (unsigned)m * 8u;
    double xr[G][RB], xm[G][RB], acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.0;
#pragma unroll
    for (int k = 0; k < G; ++k)
#pragma unroll
        for (int r = 0; r < RB; ++r) xr[k][r] = xm[k][r] = 0.0;
    for (int t = -G; t < steps; t += G) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] += xr[k][r] + (MIRROR ? xm[k][r] : 0.0);
            const int st = t + k + G < steps ? t + k + G : steps - 1;
            const int n = n_begin + lane + 64 * st;
            const int nc = n < span ? n : span - 1;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                xr[k][r] = rl_row_load(xrow[r], rowbytes, (unsigned)nc * 8u);
                if (MIRROR) xm[k][r] = rl_row_load(xrow[r], rowbytes, (unsigned)(m - 1 - nc) * 8u);
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < RB; ++r) s += acc[r];
    if (s == 1.2345) out[0] = s;
}
template <int RB, bool MIRROR, int WAVES, int OCC, int G>
static void read_run(const char* what, int m, int nrows, int steps, int rowfast) {
    double *X, *out;
    hipMalloc(&X, (size_t)nrows * m * 8);
    hipMemset(X, 0, (size_t)nrows * m * 8);
    hipMalloc(&out, 8);
    const int span = MIRROR ? (m + 1) / 2 : m, chunks = (span + 64 * steps - 1) / (64 * steps);
    const int rows_wg = RB * WAVES;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> tp;
    for (int rep = 0; rep < 12; ++rep) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_read<RB, MIRROR, WAVES, OCC, G>), dim3(chunks, (nrows + rows_wg - 1) / rows_wg),
                           dim3(64 * WAVES), 0, 0, (const double*)X, nrows, m, steps, rowfast, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) tp.push_back(ms);
    }
    std::sort(tp.begin(), tp.end());
    const double gb = (double)nrows * m * 8 / 1e9;
    printf("%-58s steps %3d chunks %3d wgs %5d: %6.1f us %5.2f TB/s\n", what, steps, chunks,
           chunks * ((nrows + rows_wg - 1) / rows_wg), 1e3 * tp[tp.size() / 2], gb / tp[tp.size() / 2]);
    hipFree(X); hipFree(out);
}

// ---- part 3: the plain ceilings of this box: read-only, write-only and copy of the same 1.03 GB
__global__ void __launch_bounds__(256) k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n2, int mode) {
    double2 s = {0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        if (mode == 0) { const double2 v = a[i]; s.x += v.x; s.y += v.y; }       // read
        else if (mode == 1) b[i] = double2{1.0, 2.0};                            // write
        else b[i] = a[i];                                                        // copy
    }
    if (mode == 0 && s.x == 1.2345) b[0] = s;
}
static void ceilings() {
    const size_t n2 = (size_t)1290 * 100004 / 2;
    double2 *a, *b;
    hipMalloc(&a, n2 * 16);
    hipMalloc(&b, n2 * 16);
    hipMemset(a, 0, n2 * 16);
    hipMemset(b, 0, n2 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[3] = {"read 1.03 GB", "write 1.03 GB", "copy 1.03 -> 1.03 GB"};
    for (int blocks : {2048, 8192})
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<float> t;
            for (int rep = 0; rep < 12; ++rep) {
                float ms;
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const double2*)a, b, n2, mode);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2) t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            const double gb = (mode == 2 ? 2.0 : 1.0) * n2 * 16 / 1e9;
            printf("%-22s blocks %5d: %6.1f us  %5.2f TB/s (bytes moved / time)\n", names[mode], blocks, 1e3 * t[t.size() / 2],
                   gb / t[t.size() / 2]);
        }
    hipFree(a); hipFree(b);
}

int main() {
    ceilings();
    {
        const int m = 100004, nr = 1290;
        read_run<4, true, 4, 2, 2>("as the projection: 4 rows/wave, mirror, 4 waves, occ 2, ring 2", m, nr, 32, 0);
        read_run<4, true, 4, 2, 2>("same again (another allocation)", m, nr, 32, 0);
        read_run<4, true, 4, 8, 2>("occupancy 8", m, nr, 32, 0);
        read_run<4, true, 4, 8, 4>("occupancy 8, ring 4", m, nr, 32, 0);
        read_run<4, false, 4, 2, 2>("no mirror (rows ascending)", m, nr, 32, 0);
        read_run<4, false, 4, 8, 4>("no mirror, occupancy 8, ring 4", m, nr, 32, 0);
        read_run<1, true, 4, 8, 4>("1 row/wave, mirror, occupancy 8, ring 4", m, nr, 32, 0);
        read_run<1, false, 4, 8, 4>("1 row/wave, no mirror, occupancy 8, ring 4", m, nr, 32, 0);
        read_run<2, true, 4, 4, 4>("2 rows/wave, mirror, occ 4, ring 4", m, nr, 32, 0);
        read_run<4, true, 4, 2, 2>("row block fastest", m, nr, 32, 1);
        read_run<4, true, 4, 2, 2>("steps 8", m, nr, 8, 0);
        read_run<4, true, 4, 2, 2>("steps 16", m, nr, 16, 0);
        read_run<4, true, 2, 2, 2>("2 waves per workgroup", m, nr, 32, 0);
        read_run<4, true, 8, 2, 2>("8 waves per workgroup", m, nr, 32, 0);
        read_run<8, true, 4, 2, 2>("8 rows/wave", m, nr, 32, 0);
    }

    const int nrows = 1290;
    for (int m : {100004, 100000}) {
        run<2>(m, nrows, 32);
        run<24>(m, nrows, 32);
    }
    run<24>(100004, nrows, 64);
    return 0;
}
