// Polynomial-subspace form of the grid product for SMOOTH kernels.
//
// A Toeplitz block T_q[i][j] = k_q(|x_i - x_j|) of a kernel that is smooth over
// the whole grid (an RBF or periodic kernel whose length scale is not small
// against the grid's extent -- the reference's synthetic benchmarks draw
// inverse length scales from logspace(0, 1, Q) on the unit interval) is
// numerically of very low rank on that grid: with Phi (m x r) the orthonormal
// polynomials of degree < r on the grid points,
//
//     T_q = Phi C_q Phi^T   to roundoff,   C_q = Phi^T T_q Phi   (r x r),
//
// r = 24 for every block of BASELINE's C2 and C5 (measured: 1e-14 relative).
// Then  K_UU X = Phi [ sum_q B_q (x) C_q ] Phi^T X :  one pass over X, a tiny
// dense map on r coefficients per output, one pass over Y -- the ALGORITHMIC
// traffic of the product (read x, write y) and r multiply-adds per element each
// way, instead of four passes over zero-padded complex intermediates.
//
// The form is used only when it is as exact as the transform path: for every
// top row the host verifies, on random vectors and against the FFT kernels of
// this same handle, that the two products agree to RL_LR_TOL of the result's
// largest entry; otherwise -- Matern kernels, short length scales, 2-D grids --
// the handle stays on the FFT path (rl_kernels2/3.h).  Reference semantics
// either way: runlmc/linalg/bttb.py:144-148, kronecker.py:39-46.
#pragma once
#include "rl_device.h"

#define RL_LR_RMAX 48          // basis functions generated per handle
#define RL_LR_T 32             // lane-steps of a projection chunk (64 points each) are a multiple of this
#define RL_LR_WAVES 4          // waves per projection workgroup (2: equal, 8: 213 vs 205 us at C5)
#define RL_LR_CUS 256          // compute units of the one target (MI355X, 8 XCDs x 32)
// rows per wave of the projection (RB x R running sums in registers: 4 x 24 or
// 2 x 32 / 2 x 48 doubles) and how many lane-steps ahead the x values are requested
// (a ring of G x RB x 2 doubles in registers: a point and its mirror)
#define RL_LR_RB(R) ((R) <= 24 ? 4 : 2)
#define RL_LR_G(R) ((R) <= 24 ? 2 : 4)
#define RL_LR_ROWS(R) (RL_LR_RB(R) * RL_LR_WAVES)   // rows per projection workgroup
// two projection waves per SIMD (256 registers each); the emulator has no such attribute
// (and no scalar registers: RL_LR_UNIFORM marks a value that is the same in every
// lane of a wave, so that row pointers and offsets live in SGPRs)
#if defined(RL_EMU)
#define RL_LR_PROJECT_ATTR
#define RL_LR_UNIFORM(x) (x)
#else
#define RL_LR_PROJECT_ATTR __attribute__((amdgpu_waves_per_eu(2)))
#define RL_LR_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
#define RL_LR_TOL 2e-13        // accepted |y_fft - y_lr| / max|y_fft| at set time

// The basis is never read from memory by the two streaming kernels: a lane
// evaluates the UNNORMALISED polynomials of its grid point by the three-term
// recurrence  q_0 = 1, q_{j+1} = s q_j - beta_j q_{j-1}  (Phi_j = nu_j q_j; beta,
// nu from the host's long-double recurrence; the normalisation nu is applied to
// the r coefficients in k_lr_mix).  Evaluated in fp64 the recurrence reproduces
// the long-double basis to 1e-14 of its largest entry (degree 48, m = 5e3 .. 1e5).
// The grid is symmetric about its centre and q_j(-s) = (-1)^j q_j(s): a lane owns
// a grid point n of the first half TOGETHER with its mirror m-1-n.  One recurrence
// serves both, the projection feeds the even degrees with x(n) + x(mirror) and the
// odd ones with x(n) - x(mirror), the expansion forms the even and the odd part of
// y once and stores their sum and their difference: half the multiply-adds and
// half the recurrences per element (the centre point of an odd grid is its own
// mirror and counts once).  "Slots" below are the (m + 1) / 2 points of that
// first half.
__device__ __forceinline__ int lr_slots(int m) { return (m + 1) / 2; }
__device__ __forceinline__ double lr_point(int n, int m) {
    return m > 1 ? fma(2.0 / (double)(m - 1), (double)n, -1.0) : 0.0;
}

// ---------------------------------------------------------------------------
// k_lr_project<R>: part[chunk][row][j] = sum_{n or its mirror in chunk} q_j(n) X[row][n],
// rows = the nrows contiguous length-m blocks of X (vector-major, output-minor);
// a chunk is 64 * steps slots.
//   grid (nchunks, ceil(nrows / RL_LR_ROWS(R)))   block 64 * RL_LR_WAVES
// Lanes run along the grid (every load is 512 contiguous bytes of one row --
// ascending for the slots, descending for their mirrors), a
// wave owns RB = RL_LR_RB(R) rows and keeps their RB x R running sums in registers over
// the `steps` points of each lane (a multiple of RL_LR_T, chosen by the host: long
// chunks amortise the reduction); the 64 lanes are summed once per chunk
// through LDS.  x values are requested RL_LR_G(R) lane-steps before their use.
// (Round 3: non-temporal loads for x, A/B in ONE job, three alternations: 0.45 / 0.45 /
// 0.42 ms per C5 product with plain loads, 0.47 / 0.49 / 0.43 with non-temporal ones -- no
// gain, and the run-to-run spread on one box is as large as the box-to-box one.)
// (Round 3, same protocol: two rows per wave and a ring of four lane-steps at rank 24 --
// 148 registers, three waves per SIMD, half as much again in flight per CU: 0.497 / 0.498 /
// 0.498 ms against 0.494 / 0.492 / 0.492.  Neither occupancy nor bytes in flight bound it.)
// (Round 5: three rows per wave and a ring of two at rank 36 -- 256 registers, 8 spilled:
// 336 against 300-315 us per C5 projection.)
// (Measured and dropped, round 3: the projection as a tall-skinny product on the fp64
// matrix cores -- v_mfma_f64_16x16x4_f64, 16 rows x 16 functions per tile, basis tile
// in LDS, sums never reduced across lanes, 124 VGPRs.  The instruction's A layout puts
// the 16 ROWS on adjacent lanes and the 4 slots on lane groups, so fragments loaded
// straight from global memory are 64 separate 8-byte requests per instruction: 0.31 ms
// against 0.19 ms per C5 projection.  Feeding it needs the LDS transposition of
// rl_filter.h's tiles, which costs what the matrix cores save here.)
// ---------------------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(64 * RL_LR_WAVES) RL_LR_PROJECT_ATTR
k_lr_project(const double* __restrict__ X, int nrows, int m, const double* __restrict__ beta,
             int steps, double* __restrict__ part) {
    constexpr int RB = RL_LR_RB(R), G = RL_LR_G(R), ROWS = RL_LR_ROWS(R);
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);       // [WAVES][R][65]
    const int tid = threadIdx.x, lane = tid & 63, wave = RL_LR_UNIFORM(tid >> 6);
    // (chunk fastest: consecutive workgroups read neighbouring chunks of the same
    // rows; the row-fastest order that pays in the expansion measured 206 vs 190 us
    // here, and the four waves side by side on the same four rows -- 2 KB instead of
    // 512 B contiguous per row and lane-step -- 214 vs 201)
    const int pbx = blockIdx.x, pby = blockIdx.y;
    const int row0 = pby * ROWS + wave * RB;
    const int n_begin = pbx * (64 * steps);
    const double* xrow[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r)
        xrow[r] = X + (size_t)(row0 + r < nrows ? row0 + r : nrows - 1) * m;
    double acc[RB][R];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[r][j] = 0.0;
    // x values travel through a ring of G lane-steps: the slot of the step just
    // consumed is requested again for the step G ahead (unconditional loads from
    // clamped points; points past the end count as zero)
    static_assert(RL_LR_T % G == 0, "ring length divides the chunk");
    const int slots = lr_slots(m);
    double xr[G][RB], xm[G][RB];
    // (buffer accesses, rl_device.h: row base in scalar registers + a byte offset.  As plain
    // loads from the const __restrict__ X the compiler folded "value carried from the previous
    // iteration's load" into "load at the point of use" -- sixteen loads at the top of every
    // iteration, each waited for on the spot: the ring existed in the source only.)
    constexpr bool RING = R <= 24;
    const unsigned rowbytes = (unsigned)m * 8u;
    auto request = [&](int slot, int step) {
        const int n = n_begin + lane + 64 * step;
        const int nc = n < slots ? n : slots - 1;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if constexpr (RING) {
                xr[slot][r] = rl_row_load(xrow[r], rowbytes, (unsigned)nc * 8u);
                xm[slot][r] = rl_row_load(xrow[r], rowbytes, (unsigned)(m - 1 - nc) * 8u);
            } else {
                xr[slot][r] = xrow[r][nc];
                xm[slot][r] = xrow[r][m - 1 - nc];
            }
        }
    };
    // The first requests are issued by the loop itself, in an iteration of their own that
    // consumes zeros (t = -G: one iteration of arithmetic more per chunk).  Issued before the
    // loop they were pending on the way in, row by row instead of slot by slot, and the wait
    // counts of the loop -- derived from the worse of the two ways into it -- let every
    // iteration wait for all but two of its sixteen loads.
    // (Ranks above 24 -- two rows per wave, a ring of four lane-steps -- keep the plain loads
    // and the requests ahead of the loop: there the iteration of zeros is an eighth of a
    // 32-step chunk's arithmetic and the pinned order exposes the recurrence's latency --
    // C5 periodic, rank 36: 0.61 against 0.56 ms per product.)
#pragma unroll
    for (int k = 0; k < G; ++k) {
        if constexpr (RING) {
#pragma unroll
            for (int r = 0; r < RB; ++r) xr[k][r] = xm[k][r] = 0.0;
        } else {
            request(k, k);
        }
    }
#pragma unroll 1
    for (int t = RING ? -G : 0; t < steps; t += G) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
#if !defined(RL_EMU)
            // (the lane-steps stay in source order: scheduled freely, every step's values were
            // consumed -- and waited for -- at the top of the iteration)
            if constexpr (RING) __builtin_amdgcn_sched_barrier(0);
#endif
            const int n = n_begin + lane + 64 * (t + k);
            const int nc = n < slots ? (n > 0 ? n : 0) : slots - 1;     // (n < 0: the iteration of zeros)
            const double s = lr_point(nc, m);
            const double live = n < slots ? 1.0 : 0.0;
            const double pair = m - 1 - nc != nc ? live : 0.0;     // (the centre has no mirror)
            double xe[RB], xo[RB];          // what the even / the odd degrees see
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const double a = xr[k][r] * live, b = xm[k][r] * pair;
                xe[r] = a + b;
                xo[r] = a - b;
            }
            // (the last G requests repeat the chunk's last step: no branch, a cache hit)
            request(k, t + k + G < steps ? t + k + G : steps - 1);
            double qm = 0.0, q = 1.0;
#pragma unroll
            for (int j = 0; j < R; ++j) {
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    acc[r][j] = fma(q, (j & 1) ? xo[r] : xe[r], acc[r][j]);
                const double qn = fma(s, q, -beta[j] * qm);
                qm = q;
                q = qn;
            }
        }
    }
    // sum over the 64 lanes, one row at a time: [wave][j][lane] in LDS, then one
    // thread per (wave, j)
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int j = 0; j < R; ++j) red[(wave * R + j) * 65 + lane] = acc[r][j];
        __syncthreads();
        for (int e = tid; e < RL_LR_WAVES * R; e += 64 * RL_LR_WAVES) {
            const int w = e / R, j = e - w * R;
            const double* src = red + (size_t)e * 65;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 2
            for (int l = 0; l < 64; l += 4) {
                s0 += src[l];
                s1 += src[l + 1];
                s2 += src[l + 2];
                s3 += src[l + 3];
            }
            const int row = pby * ROWS + w * RB + r;
            if (row < nrows)
                part[((size_t)pbx * nrows + row) * R + j] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// k_lr_mix: per vector, sum the projection's partial results over the chunks and
// apply the dense coefficient map
//     Zhat[a][i] = nu_i sum_q sum_b B_q[a][b] sum_j C_q[i][j] (nu_j Z[b][j])
// (nu: the basis normalisation, see lr_point above).
//   grid (nvec)   block RL_LR_MIXT = 1024 (four groups of 256)
//   LDS: Z [D][r] + W [Q][D][r] + S [max(4, Q)][D][r]   (lr_mix_lds below; operators whose
//   tables would pass 64 KB -- sixteen outputs at rank 48 with more than four terms -- run with
//   S inside W's space and the last step as one chain per output: split3 = 0)
//   Cq [Q][r][r], Bq [Q][D][D] (single-top products pass Q = 1, B = identity)
// The kernel is three dependent steps of a few thousand multiply-adds: latency, not work.
// With 256 threads each step was a chain (25 chunk loads in four batches; five outputs of
// 24 loads each; 50 multiply-adds per output): 15-19 us between the two streaming kernels
// of every product and every solver round.  Four groups of threads split the chunks
// (group k: chunks k, k + 4, ...; the four sums meet in LDS in a fixed order), the second
// step's Q D r outputs and the third step's sum over q go one per thread.
// ---------------------------------------------------------------------------
#define RL_LR_MIXT 1024
// LDS bytes of k_lr_mix and whether its last step runs split over q (host side)
static inline size_t lr_mix_lds(int D, int r, int Q, int* split3) {
    const size_t dr = (size_t)D * r, q4 = (size_t)(Q > 4 ? Q : 4);
    const size_t full = (1 + (size_t)Q + q4) * dr * sizeof(double);
    *split3 = full <= 64 * 1024 ? 1 : 0;
    return *split3 ? full : (1 + q4) * dr * sizeof(double);
}
static __global__ void __launch_bounds__(RL_LR_MIXT)
k_lr_mix(const double* __restrict__ part, int nchunks, int nvec, int D, int r, int Q,
         const double* __restrict__ Cq, const double* __restrict__ Bq,
         const double* __restrict__ nu, double* __restrict__ Zhat,
         const int* __restrict__ run_ptr, int split3) {
    RL_SMEM(smem);
    double* Z = reinterpret_cast<double*>(smem);          // [D][r]
    double* W = Z + D * r;                                 // [Q][D][r]  (split3 = 0: [max(4, Q)][D][r])
    double* S = split3 ? W + Q * D * r : W;                // [max(4, Q)][D][r]
    const int v = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int grp = tid >> 8, t = tid & 255, ngrp = nthr >> 8;        // (nthr a multiple of 256)
    const int nrows = nvec * D, Dr = D * r;
    // step 1: group grp sums its share of the chunks (or, for k_rp_project's partial sums
    // part[run][v][j], of the runs run_ptr[b] .. run_ptr[b + 1] of output b) in ascending order
    for (int e = t; e < Dr; e += 256) {
        int c0 = 0, c1 = nchunks;
        size_t stride = (size_t)nrows * r;
        const double* src = part + (size_t)v * Dr + e;
        if (run_ptr != nullptr) {
            const int b = e / r, j = e - b * r;
            c0 = run_ptr[b];
            c1 = run_ptr[b + 1];
            stride = (size_t)nvec * r;
            src = part + (size_t)v * r + j;
        }
        double s = 0.0;
        for (int c = c0 + grp; c < c1; c += ngrp) s += src[(size_t)c * stride];
        S[grp * Dr + e] = s;
    }
    __syncthreads();
    for (int e = tid; e < Dr; e += nthr) {
        double s = 0.0;
        if (ngrp == 4) s = (S[e] + S[Dr + e]) + (S[2 * Dr + e] + S[3 * Dr + e]);
        else for (int k = 0; k < ngrp; ++k) s += S[k * Dr + e];
        Z[e] = nu[e % r] * s;
    }
    __syncthreads();
    // step 2: W[q][b][i] = sum_j C_q[i][j] Z[b][j]
    for (int e = tid; e < Q * Dr; e += nthr) {
        const int q = e / Dr, rem = e - q * Dr;
        const int b = rem / r, i = rem - b * r;
        const double* c = Cq + ((size_t)q * r + i) * r;
        const double* z = Z + b * r;
        double s = 0.0;
        for (int j = 0; j < r; ++j) s = fma(c[j], z[j], s);
        W[e] = s;
    }
    __syncthreads();
    if (!split3) {
        for (int e = tid; e < Dr; e += nthr) {
            const int a = e / r, i = e - a * r;
            double s = 0.0;
            for (int q = 0; q < Q; ++q) {
                const double* bq = Bq + ((size_t)q * D + a) * D;
                for (int b = 0; b < D; ++b) s = fma(bq[b], W[(q * D + b) * r + i], s);
            }
            Zhat[(size_t)v * Dr + e] = nu[i] * s;
        }
        return;
    }
    // step 3: T[q][a][i] = sum_b B_q[a][b] W[q][b][i], then the sum over q in ascending order
    for (int e = tid; e < Q * Dr; e += nthr) {
        const int q = e / Dr, rem = e - q * Dr;
        const int a = rem / r, i = rem - a * r;
        const double* bq = Bq + ((size_t)q * D + a) * D;
        double s = 0.0;
        for (int b = 0; b < D; ++b) s = fma(bq[b], W[(q * D + b) * r + i], s);
        S[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < Dr; e += nthr) {
        double s = 0.0;
        for (int q = 0; q < Q; ++q) s += S[q * Dr + e];
        Zhat[(size_t)v * Dr + e] = nu[e % r] * s;
    }
}

// ---------------------------------------------------------------------------
// k_lr_expand<R>: Y[row][n] (+)= sum_j q_j(n) Zhat[row][j]  (Zhat carries nu).
//   grid (ceil(slots / 256), ceil(nrows / rows_per_block))   block 256
// A thread owns one slot (a grid point and its mirror): the R basis values of
// the point (recurrence, once) stay in registers for all the rows of the block;
// the coefficients of a row are the same for every lane (scalar loads); the even
// and the odd degrees are summed separately, y(n) is their sum and y(mirror)
// their difference.  (Measured and dropped: two slots per thread 318 vs 250 us
// at C5, two rows of coefficients in flight 324 -- the scalar registers run
// out --, non-temporal stores 271 vs 243.  Round 3: the expansion of the larger
// ranks on the fp64 matrix cores -- Zhat fragments in registers, the chunk's basis in
// LDS, 16 rows x 16 slots per tile, 128-byte stores: 555 vs 330-370 us at rank 48.)
// ---------------------------------------------------------------------------
// (ACC is a template parameter, not a run-time flag: with the read of Y behind a run-time
// condition the compiler waited for EVERY outstanding access before each of the two stores
// of a row -- one store in flight per wave, which is what bounded the kernel.  The plain
// kernel has no load at all, the accumulating one loads unconditionally from the clamped
// positions.)
template <int R, bool ACC>
__global__ void __launch_bounds__(256)
k_lr_expand(const double* __restrict__ Zhat, int nrows, int m, const double* __restrict__ beta,
            int rows_per_block, double* __restrict__ Y) {
    const int slots = lr_slots(m);
    // workgroups are numbered with the ROW block fastest: consecutive workgroups
    // write the same columns of different rows (measured at C5: 204-215 us against
    // 252-266 with the column block fastest, where the whole chip writes the same
    // few rows in lockstep)
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int bx = lin / gridDim.y, by = lin - bx * gridDim.y;
    const int n = bx * 256 + threadIdx.x;
    const int nc = n < slots ? n : slots - 1;
    const int mir = m - 1 - nc;
    const bool live = n < slots, pair = live && mir != nc;
    const int row0 = by * rows_per_block;
    const int row1 = row0 + rows_per_block < nrows ? row0 + rows_per_block : nrows;
    const double s = lr_point(nc, m);
    double p[R];
    {
        double qm = 0.0, q = 1.0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            p[j] = q;
            const double qn = fma(s, q, -beta[j] * qm);
            qm = q;
            q = qn;
        }
    }
    for (int row = row0; row < row1; ++row) {
        const double* z = Zhat + (size_t)row * R;
        // (the row's coefficients are fetched in ONE batch of scalar loads, pinned ahead of the
        // multiply-adds: left to itself the scheduler fetched 16 at a time, each batch waited
        // for before the next was issued -- five scalar-memory round trips per row at rank 36,
        // 300 us per C5 expansion instead of 240)
        double zz[R];
#pragma unroll
        for (int j = 0; j < R; ++j) zz[j] = z[j];
#if !defined(RL_EMU)
        __builtin_amdgcn_sched_barrier(0);
#endif
        double ev = 0.0, od = 0.0;
#pragma unroll
        for (int j = 0; j + 1 < R; j += 2) {
            ev = fma(zz[j], p[j], ev);
            od = fma(zz[j + 1], p[j + 1], od);
        }
        // (accumulate: the operator's filter part has written Y already, rl_filter.h)
        double* y0 = Y + (size_t)row * m + nc;
        double* y1 = Y + (size_t)row * m + mir;
        double o0 = 0.0, o1 = 0.0;
        if constexpr (ACC) {
            o0 = *y0;
            o1 = *y1;
        }
        if (live) *y0 = (ev + od) + o0;
        if (pair) *y1 = (ev - od) + o1;
    }
}

// ---------------------------------------------------------------------------
// Small batches (round 6): k_lr_small_project<R> + k_lr_small_expand<R>, TWO launches.
// Below the batch gate the three kernels above are three dependent launches of a few
// workgroups each, and their projection walks a whole row per wave: BASELINE's C2 batch (17
// vectors, D = 4, m = 5004) took 25 us that way -- the projection alone 17 -- and 18 us on the
// three transform kernels.  What such a batch needs is its few hundred thousand multiply-adds
// spread over the WHOLE chip and as few dependent launches as the data flow allows:
//   k_lr_small_project  grid (D nseg, nvec): workgroup (b, sg, v) projects segment sg of row b
//                       of vector v (a quarter row, say) -> part[v][b][sg][R]
//   k_lr_small_expand   grid (D nseg, nvec): workgroup (a, sg, v) sums the vector's partials,
//                       applies ROW BLOCK a of the coefficient map
//                         Mf[a][i][b][j] = nu_i nu_j sum_q B_q[a][b] C_q[i][j]
//                       (host-built per parameter update, forms_setup; the R x D R block is
//                       re-applied by each of the row's nseg workgroups: 2 304 multiply-adds at
//                       C2, nothing against a hand-off) and expands its segment of row a.
// (First built as ONE launch, a workgroup per (row, vector) that projected all D rows itself:
// 15.6 us whatever the batch -- 720 k multiply-adds through one compute unit's fp64 pipe are
// 5 us, the expansion 1.3, and the 64-lane sums waited for the slowest wave; stamps in
// profiles/r06/small_batch_probe.txt.  One launch with an in-kernel hand-off between the
// projection and the map would save one launch gap and risk a spin-wait; two launches do not.)
// Same arithmetic as k_lr_project / k_lr_mix / k_lr_expand (point + mirror parity, the monic
// recurrence), another summation order.  Used only for operators ALREADY verified to be wholly
// in the polynomial form, D m <= RL_LR_SMALL_MAX elements per vector.
// Four slots run their recurrences in lockstep (the loop over degrees outermost): one slot at
// a time is a chain of 24 dependent multiply-adds, 28 cycles a step on this pipe.
// ---------------------------------------------------------------------------
#define RL_LR_SMALL_MAX 32768
#define RL_LR_SMALL_WG 256      // threads of both kernels
#define RL_LR_SU 4              // slots in lockstep
#define RL_LR_SMP 32            // map entries a thread requests ahead (D r <= 4 * this: its whole share)
// segments per row: one pass of a workgroup's 256 threads x 4 lockstep slots each (at most 16)
static inline int lr_small_nseg(int m) {
    const int slots = (m + 1) / 2;
    int nseg = (slots + RL_LR_SMALL_WG * RL_LR_SU - 1) / (RL_LR_SMALL_WG * RL_LR_SU);
    return nseg < 1 ? 1 : (nseg > 16 ? 16 : nseg);
}
template <int R>
__global__ void __launch_bounds__(RL_LR_SMALL_WG)
k_lr_small_project(const double* __restrict__ X, int D, int m, int nseg, const double* __restrict__ beta,
                   double* __restrict__ part) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);               // [4][R][65]
    double* wsum = red + (size_t)4 * R * 65;                      // [4][R]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / nseg, sg = blockIdx.x - b * nseg, v = blockIdx.y;
    const int slots = lr_slots(m), seglen = (slots + nseg - 1) / nseg;
    const int n0 = sg * seglen, n1 = n0 + seglen < slots ? n0 + seglen : slots;
    const double* xb = X + ((size_t)v * D + b) * m;
    double acc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j] = 0.0;
    for (int nb = n0 + tid; nb < n1; nb += RL_LR_SMALL_WG * RL_LR_SU) {
        double s[RL_LR_SU], q[RL_LR_SU], qm[RL_LR_SU], xe[RL_LR_SU], xo[RL_LR_SU];
#pragma unroll
        for (int u = 0; u < RL_LR_SU; ++u) {
            const int n = nb + RL_LR_SMALL_WG * u;
            const bool live = n < n1;
            const int nc = live ? n : n1 - 1, mir = m - 1 - nc;
            const double x0 = live ? xb[nc] : 0.0, x1 = live && mir != nc ? xb[mir] : 0.0;
            xe[u] = x0 + x1;
            xo[u] = x0 - x1;
            s[u] = lr_point(nc, m);
            q[u] = 1.0;
            qm[u] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const double nbj = -beta[j];
#pragma unroll
            for (int u = 0; u < RL_LR_SU; ++u) {
                acc[j] = fma(q[u], (j & 1) ? xo[u] : xe[u], acc[j]);
                const double qn = fma(s[u], q[u], nbj * qm[u]);
                qm[u] = q[u];
                q[u] = qn;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) red[((size_t)wave * R + j) * 65 + lane] = acc[j];
    __syncthreads();
    // thread (w, j) sums the 64 lanes of wave w, then the four waves meet
    if (tid < 4 * R) {
        const double* src = red + (size_t)tid * 65;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
        for (int l = 0; l < 64; l += 4) {
            s0 += src[l];
            s1 += src[l + 1];
            s2 += src[l + 2];
            s3 += src[l + 3];
        }
        wsum[tid] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (tid < R)
        part[(((size_t)v * D + b) * nseg + sg) * R + tid] =
            (wsum[tid] + wsum[R + tid]) + (wsum[2 * R + tid] + wsum[3 * R + tid]);
}
static inline size_t lr_small_project_lds(int R) { return ((size_t)4 * R * 65 + 4 * R) * sizeof(double); }
static inline size_t lr_small_expand_lds(int D, int R) { return ((size_t)D * R + 256 + R) * sizeof(double); }

template <int R>
__global__ void __launch_bounds__(RL_LR_SMALL_WG)
k_lr_small_expand(const double* __restrict__ part, int D, int m, int nseg, const double* __restrict__ beta,
                  const double* __restrict__ Mf, double* __restrict__ Y) {
    RL_SMEM(smem);
    double* Z = reinterpret_cast<double*>(smem);                  // [D][R]
    double* P = Z + (size_t)D * R;                                // [256]
    double* Zh = P + 256;                                         // [R]
    const int tid = threadIdx.x;
    const int a = blockIdx.x / nseg, sg = blockIdx.x - a * nseg, v = blockIdx.y;
    const int DR = D * R;
    // this thread's share of row block a of the map -- coefficient i = tid / 4, entries pt,
    // pt + 4, ... -- requested FIRST (it does not depend on the partial sums): the first
    // RL_LR_SMP entries here, the rest (D r > 4 RL_LR_SMP) in batches below
    const int i = tid >> 2, pt = tid & 3;
    const double* mrow = Mf + ((size_t)a * R + (i < R ? i : 0)) * DR;
    double mreg[RL_LR_SMP];
#pragma unroll
    for (int k = 0; k < RL_LR_SMP; ++k) {
        const int e = pt + 4 * k;
        mreg[k] = mrow[e < DR ? e : pt];
    }
    // the vector's coefficients: segments summed in ascending order
    for (int e = tid; e < DR; e += RL_LR_SMALL_WG) {
        const double* src = part + ((size_t)v * DR + e - (e % R)) * nseg + (e % R);
        double z = 0.0;
        for (int g = 0; g < nseg; ++g) z += src[(size_t)g * R];
        Z[e] = z;
    }
    __syncthreads();
    // Zh[i] = sum_e Mf[a][i][e] Z[e]: four partial sums per coefficient (R <= 64, 256 threads)
    {
        double sacc = 0.0;
        if (i < R) {
#pragma unroll
            for (int k = 0; k < RL_LR_SMP; ++k) {
                const int e = pt + 4 * k;
                if (e < DR) sacc = fma(mreg[k], Z[e], sacc);
            }
            for (int e0 = pt + 4 * RL_LR_SMP; e0 < DR; e0 += 4 * RL_LR_SMP) {
                double mm[RL_LR_SMP];
#pragma unroll
                for (int k = 0; k < RL_LR_SMP; ++k) {
                    const int e = e0 + 4 * k;
                    mm[k] = mrow[e < DR ? e : pt];
                }
#pragma unroll
                for (int k = 0; k < RL_LR_SMP; ++k) {
                    const int e = e0 + 4 * k;
                    if (e < DR) sacc = fma(mm[k], Z[e], sacc);
                }
            }
        }
        P[tid] = sacc;
        __syncthreads();
        if (tid < R) Zh[tid] = (P[4 * tid] + P[4 * tid + 1]) + (P[4 * tid + 2] + P[4 * tid + 3]);
        __syncthreads();
    }
    double zz[R];
#pragma unroll
    for (int j = 0; j < R; ++j) zz[j] = Zh[j];
    const int slots = lr_slots(m), seglen = (slots + nseg - 1) / nseg;
    const int n0 = sg * seglen, n1 = n0 + seglen < slots ? n0 + seglen : slots;
    double* ya = Y + ((size_t)v * D + a) * m;
    for (int nb = n0 + tid; nb < n1; nb += RL_LR_SMALL_WG * RL_LR_SU) {
        double s[RL_LR_SU], q[RL_LR_SU], qm[RL_LR_SU], ev[RL_LR_SU], od[RL_LR_SU];
#pragma unroll
        for (int u = 0; u < RL_LR_SU; ++u) {
            const int n = nb + RL_LR_SMALL_WG * u;
            s[u] = lr_point(n < n1 ? n : n1 - 1, m);
            q[u] = 1.0;
            qm[u] = 0.0;
            ev[u] = od[u] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const double nbj = -beta[j];
#pragma unroll
            for (int u = 0; u < RL_LR_SU; ++u) {
                if (j & 1) od[u] = fma(zz[j], q[u], od[u]);
                else ev[u] = fma(zz[j], q[u], ev[u]);
                const double qn = fma(s[u], q[u], nbj * qm[u]);
                qm[u] = q[u];
                q[u] = qn;
            }
        }
#pragma unroll
        for (int u = 0; u < RL_LR_SU; ++u) {
            const int n = nb + RL_LR_SMALL_WG * u;
            if (n < n1) {
                const int mir = m - 1 - n;
                ya[n] = ev[u] + od[u];
                if (mir != n) ya[mir] = ev[u] - od[u];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Row-wise pieces for SMALL systems (rl_solver.h, Minres2Bufs::poly_part): with
// K_UU = Phi M Phi^T a solver round needs no grid vector at all.  A data row i of
// output d interpolates the four grid points n_i .. n_i + 3 with weights w_i:
//   (W Phi)[i, j] = sum_e w_i[e] Phi_j(n_i + e)
// so the projection of W^T y is accumulated row by row (lr_row_accumulate) and the
// four grid values of a row come straight from the r mixed coefficients of its
// output (lr_row_values) -- four recurrences per row each way, rank RL_LR_RS.
// Points past the end of the grid only ever meet zero weights (the polynomial is
// finite there).
// ---------------------------------------------------------------------------
#define RL_LR_RS 24
// g[e] = sum_j z[j] q_j(n + e),  e < 4   (z carries the normalisation; pass it in
// REGISTERS: read from LDS inside the recurrence it costs a round trip per degree)
// (lr_point_values: the same at four arbitrary points)
__device__ __forceinline__ void lr_point_values(const double* z, const double* __restrict__ beta,
                                                const int n[4], int m, double g[4]) {
    double s[4], qm[4], q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s[e] = lr_point(n[e], m);
        qm[e] = 0.0;
        q[e] = 1.0;
        g[e] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < RL_LR_RS; ++j) {
        const double zj = z[j], bj = beta[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            g[e] = fma(zj, q[e], g[e]);
            const double qn = fma(s[e], q[e], -bj * qm[e]);
            qm[e] = q[e];
            q[e] = qn;
        }
    }
}
__device__ __forceinline__ void lr_row_values(const double* z, const double* __restrict__ beta,
                                              int n, int m, double g[4]) {
    double s[4], qm[4], q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s[e] = lr_point(n + e, m);
        qm[e] = 0.0;
        q[e] = 1.0;
        g[e] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < RL_LR_RS; ++j) {
        const double zj = z[j], bj = beta[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            g[e] = fma(zj, q[e], g[e]);
            const double qn = fma(s[e], q[e], -bj * qm[e]);
            qm[e] = q[e];
            q[e] = qn;
        }
    }
}
// acc[j] += y * sum_e w[e] q_j(n + e)
__device__ __forceinline__ void lr_row_accumulate(double acc[RL_LR_RS],
                                                  const double* __restrict__ beta, int n, int m,
                                                  const double w[4], double y) {
    double s[4], qm[4], q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s[e] = lr_point(n + e, m);
        qm[e] = 0.0;
        q[e] = 1.0;
    }
#pragma unroll
    for (int j = 0; j < RL_LR_RS; ++j) {
        const double bj = beta[j];
        double t = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t = fma(w[e], q[e], t);
            const double qn = fma(s[e], q[e], -bj * qm[e]);
            qm[e] = q[e];
            q[e] = qn;
        }
        acc[j] = fma(y, t, acc[j]);
    }
}

// ---------------------------------------------------------------------------
// k_spmv_w_poly<VB, XPT, R>: Y = W (Phi Zhat) (+ diag (.) X2) -- the interpolation
// product of k_spmv_w_staged (rl_kernels.h) with the polynomial form's EXPANSION
// inside it: the contiguous grid range of a workgroup's RL_THREADS data rows is not
// read from a grid vector but evaluated from the mixed coefficients,
//     g[v][c] = sum_j q_j(n) Zhat[v D + d][j],   c = d m + n,
// once per group of VB vectors into the same LDS tile the rows then gather from.  A
// thread keeps the R basis values of its XPT points in registers for all the groups
// it walks (recurrence once per workgroup); the coefficients of a group -- VB vectors
// x the at most two outputs a range of <= 4 RL_THREADS < m points meets -- are staged
// in LDS and read as broadcasts.  R multiply-adds per (grid point, vector) on the
// vector pipe (0.11 ms of instructions per C5 round at full lanes) against the write
// of the grid vector by k_lr_expand and its read by k_spmv_w_staged (0.4 ms).
// (The standalone expansion sums even and odd degrees apart, for a point and its
// mirror at once; here a point's R terms are summed in order: the two agree to
// roundoff, not bit for bit.)
// (Round 4, measured and dropped -- the counters put the LDS pipe at 0.32 of this kernel's
// 0.48 ms per C5 round, SQ_LDS_IDX_ACTIVE, one 8-byte broadcast read per multiply-add:
// (a) the coefficients through SCALAR loads instead -- a wave's 64 points lie in one output
// except at an output border, three s_load_dwordx16 per (point slot, vector): 554 us against
// 517 on the same box; (b) one coefficient read serving both of a thread's points: 198
// registers, two waves per SIMD instead of three, 558 us; held to 168 registers it spills,
// 639 us.  What overlaps the LDS cycles today is worth more than removing half of them.)
//   grid (ceil(nrows / RL_THREADS), ceil(ceil(nvec / VB) / vgroups))
//   LDS: VB xcap doubles (grid values) + VB 2 R (coefficients)
// ---------------------------------------------------------------------------
template <int VB, int XPT, int R>
__global__ void __launch_bounds__(RL_THREADS)
k_spmv_w_poly(const int* __restrict__ base, const double* __restrict__ w4, int nrows, int ncols,
              int nvec, const double* __restrict__ Zhat, const double* __restrict__ beta, int D,
              int m, double* __restrict__ Y, const double* __restrict__ diag,
              const double* __restrict__ X2, int xcap, int vgroups) {
    RL_SMEM(smem);
    double* xs = reinterpret_cast<double*>(smem);          // [VB][xcap]
    double* zs = xs + (size_t)VB * xcap;                   // [VB][2][R]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int bx = lin / gridDim.y, by = lin - bx * gridDim.y;
    const int r0 = bx * nthr;
    const int rl = (r0 + nthr < nrows ? r0 + nthr : nrows) - 1;
    const int groups = (nvec + VB - 1) / VB;
    const int g0 = by * vgroups;
    const int ng = groups - g0 < vgroups ? groups - g0 : vgroups;
    const int c0 = base[r0], c1 = base[rl] + 4;
    const int len = c1 - c0;
    const int row = r0 + tid;
    const int rowc = row <= rl ? row : rl;
    const int b = base[rowc];
    double w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = w4[(size_t)4 * rowc + e];
    const double dg = diag != nullptr ? diag[rowc] : 0.0;
    // the basis at this thread's points of the range (points past the grid -- a base
    // column near the end of the last output plus its four entries -- meet zero weights)
    const int d0 = c0 / m;
    double p[XPT][R];
    int dsel[XPT];
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
        int c = c0 + tid + u * nthr;
        c = c < ncols ? c : ncols - 1;
        const int d = c / m, n = c - d * m;
        dsel[u] = d - d0 < 1 ? 0 : 1;
        const double s = lr_point(n, m);
        double qm = 0.0, q = 1.0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            p[u][j] = q;
            const double qn = fma(s, q, -beta[j] * qm);
            qm = q;
            q = qn;
        }
    }
    const double* xrow = xs + (b - c0);
    for (int grp = 0; grp < ng; ++grp) {
        const int v0 = (g0 + grp) * VB;
        for (int e = tid; e < VB * 2 * R; e += nthr) {
            const int j = e / (2 * R), rem = e - j * 2 * R, dd = rem / R, k = rem - dd * R;
            const int v = v0 + j < nvec ? v0 + j : nvec - 1;
            const int dc = d0 + dd < D ? d0 + dd : D - 1;
            zs[e] = Zhat[((size_t)v * D + dc) * R + k];
        }
        double x2[VB];
#pragma unroll
        for (int j = 0; j < VB; ++j)
            x2[j] = diag != nullptr ? X2[(size_t)(v0 + j < nvec ? v0 + j : nvec - 1) * nrows + rowc] : 0.0;
        __syncthreads();
        // (a ragged last group skips the evaluation of its empty slots: uniform branch)
        const int nvg = nvec - v0 < VB ? nvec - v0 : VB;
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int i = tid + u * nthr;
            if (i < len) {
#pragma unroll
                for (int j = 0; j < VB; ++j)
                    if (j < nvg) {
                        const double* z = zs + (j * 2 + dsel[u]) * R;
                        double ev = 0.0, od = 0.0;
#pragma unroll
                        for (int k = 0; k + 1 < R; k += 2) {
                            ev = fma(z[k], p[u][k], ev);
                            od = fma(z[k + 1], p[u][k + 1], od);
                        }
                        xs[(size_t)j * xcap + i] = ev + od;
                    }
            }
        }
        __syncthreads();
        if (row <= rl) {
#pragma unroll
            for (int j = 0; j < VB; ++j)
                if (j < nvg) {
                    double acc = 0.0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = fma(w[e], xrow[(size_t)j * xcap + e], acc);
                    if (diag != nullptr) acc = fma(dg, x2[j], acc);
                    Y[(size_t)(v0 + j) * nrows + row] = acc;
                }
        }
        __syncthreads();
    }
}

// out[row][j] = nu_j sum_c part[c][row][j]: the coefficients Phi^T x of grid rows on the NORMALISED basis from
// k_lr_project's partial sums (rl_gridop_project).   grid (ceil(nrows r / 256))   block 256
static __global__ void __launch_bounds__(256)
k_lr_coeffs(const double* __restrict__ part, int nchunks, int nrows, int r,
            const double* __restrict__ nu, double* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x, tot = (size_t)nrows * r;
    if (e >= tot) return;
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += part[(size_t)c * tot + e];
    out[e] = nu[e % r] * s;
}
