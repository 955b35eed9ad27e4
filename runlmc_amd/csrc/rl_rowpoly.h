// Row-polynomial form of a polynomial-form SKI operator (round 4): LARGE solver rounds.
//
// When every top row of the grid operator is in the polynomial-subspace form
// (rl_lowrank.h: K_UU = Phi M Phi^T on the grid's orthonormal polynomials), the SKI
// operator of reference approx/ski.py:13-16
//
//     K~ x = W K_UU W^T x + eps (.) x = F M F^T x + eps (.) x,      F = W Phi  (n x r per output)
//
// needs neither the interpolation products nor a grid vector: F[i][j] = sum_e w_i[e] q_j(n_i + e)
// -- the r polynomials interpolated at data row i -- depends on the inputs and the grid only, is
// built ONCE per (SKI handle, rank) and kept (8 r n bytes: 192 MB at C5, rank 24), and a
// solver round's operator is
//     k_rp_project   part = F^T Y   (a tall-skinny product: fp64 matrix cores, operands
//                                    staged through LDS)
//     k_lr_mix       Zhat = nu (.) sum_q B_q C_q (nu (.) Z)          (rl_lowrank.h)
//     k_rp_expand    Q = F Zhat (+ eps (.) Y)                         (scalar-loaded coefficients)
// i.e. one read of Y, one write of Q and one read of F (the expansion computes its rows of F
// from the interpolation entries, FLY below), against W^T (read Y, write g), projection
// (read g), mix, W with expansion (write Q) before: 1.18 -> 0.60 ms per C5 round, DESIGN.md
// section 6.
// Same operator, another summation order (roundoff-level agreement, as k_spmv_w_poly).
// Rows are in the handle's SORTED order (by output, then by grid position), so the rows of
// an output are contiguous; F is degree-major, F[j * n + i].
// Round 5: MINRES's two vector kernels ride inside these two -- B in the projection (FB,
// RpFuse; round 4), P in the expansion (FP, RpPFuse) -- and, for operators that are NOT
// wholly in the polynomial form, P rides inside the staged W product (k_spmv_w_staged_p at
// the end of this file: it shares RpPFuse and the scalar head k_minres2_ph).
#pragma once
#include "rl_device.h"
#include "rl_lowrank.h"

#define RL_RP_TILE 128                   // data rows per LDS tile
#define RL_RP_LD (RL_RP_TILE + 1)        // padded LDS row, ODD: the compiler reads the matrix-core operands with ds_read2_b64 (16-lane groups, 32 dword banks): lanes i = 0..15 land on banks 2 i (+1)
#define RL_RP_VG 16                      // vectors per matrix-core block (the instruction's N)
// vector blocks a workgroup walks per tile (accumulators: NG x degree tiles x 4 doubles per lane):
// 5 (80 vectors: nine, for the 129 vectors of C5 in one block, spill 83 vector registers at rank 24)
#define RL_RP_NG(R) ((R) <= 32 ? 5 : 3)
// (with the solver's update inside -- FB: a second operand, the Gram accumulators -- fewer)
#define RL_RP_NG_FB(R) ((R) <= 32 ? 3 : 2)
#define RL_RP_RMAX 48

// ---------------------------------------------------------------------------
// k_rp_build: F[j][i] = sum_e w_i[e] q_j(n_i + e)   (unnormalised q, as everywhere: the
// normalisation nu rides in k_lr_mix).  One thread per data row, four recurrences.
//   grid (ceil(n / 256))   block 256
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
k_rp_build(const int* __restrict__ base, const double* __restrict__ w4, int n, int m, int R,
           const double* __restrict__ beta, double* __restrict__ F) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = base[i], d = b / m, n0 = b - d * m;
    double w[4], s[4], qm[4], q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // (a base column near the end of an output plus its entries: zero weights there)
        const int ne = n0 + e < m ? n0 + e : m - 1;
        w[e] = n0 + e < m ? w4[(size_t)4 * i + e] : 0.0;
        s[e] = lr_point(ne, m);
        qm[e] = 0.0;
        q[e] = 1.0;
    }
    for (int j = 0; j < R; ++j) {
        double f = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) f = fma(w[e], q[e], f);
        F[(size_t)j * n + i] = f;
        const double bj = beta[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double qn = fma(s[e], q[e], -bj * qm[e]);
            qm[e] = q[e];
            q[e] = qn;
        }
    }
}

// A row's values of F straight from its interpolation entry (base column, four weights):
// f(j) called for j = 0 .. R - 1 in order.  (FLY variants of the kernels below: no table --
// F costs a 17-vector batch as much HBM traffic as its vectors do.)
struct RpRow {
    double w[4], s[4], qm[4], q[4];
    __device__ __forceinline__ void start(int b, int m, const double* __restrict__ w4row) {
        const int d = b / m, n0 = b - d * m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ne = n0 + e < m ? n0 + e : m - 1;
            w[e] = n0 + e < m ? w4row[e] : 0.0;
            s[e] = lr_point(ne, m);
            qm[e] = 0.0;
            q[e] = 1.0;
        }
    }
    __device__ __forceinline__ double next(double bj) {
        double f = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) f = fma(w[e], q[e], f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double qn = fma(s[e], q[e], -bj * qm[e]);
            qm[e] = q[e];
            q[e] = qn;
        }
        return f;
    }
};

// scatter of the interpolation entries into the caller's row order (rl_ski_mvm)
static __global__ void __launch_bounds__(256)
k_rp_permute_entries(const int* __restrict__ base, const double* __restrict__ w4,
                     const int* __restrict__ perm, int n, int* __restrict__ base_c,
                     double* __restrict__ w4_c) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int dst = perm[i];
    base_c[dst] = base[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) w4_c[(size_t)4 * dst + e] = w4[(size_t)4 * i + e];
}

// ---------------------------------------------------------------------------
// k_rp_project<R>: part[run][v][j] = sum_{i in run} F[j][i] Y[v][i].
//   grid (8 ceil(nruns / 8) x ceil(nvec / (RL_RP_NG(R) * RL_RP_VG)))   block 256 (four waves)
//   runs: [nruns][3] = first row, end row, output -- contiguous rows of ONE output
//   LDS: F tile [32][LD] + two Y tiles [16][LD]   (2 workgroups per CU)
// A run is walked in tiles of 128 rows.  Per tile the F values go to LDS once and serve all
// the workgroup's vector blocks; a block of 16 vectors is staged (coalesced: 64 consecutive
// rows of one vector per wave-load; the next block's loads in flight meanwhile) and
// multiplied on the matrix cores, v_mfma_f64_16x16x4_f64: D(16 x 16) += A(16 x 4) B(4 x 16) with
// A = F (degrees x rows), B = Y (rows x vectors); wave w owns rows 32 w .. 32 w + 31 of the
// tile (8 instructions per degree tile), accumulators stay in registers over the whole run
// (2 degree tiles x 4 doubles per vector block).  Operand layouts (tools/mfma_f64_layout.hip,
// measured on gfx950): A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k, D[row][col] in
// register row / 4 of lane col + 16 (row % 4) (RL_RP_DROW).  The four waves' results meet in LDS at the end.
// (emulator: no matrix instruction -- a thread owns two entries of the 32 x 16 result and
// sums the tile's rows itself; staging, masks and indexing are shared)
// ---------------------------------------------------------------------------
#if !defined(RL_EMU)
typedef double rp_double4 __attribute__((ext_vector_type(4)));
// D layout: register r of lane l holds row RL_RP_DROW(l, r), column l & 15
#define RL_RP_DROW(l, r) (((l) >> 4) + 4 * (r))
#define RL_RP_PROJECT_ATTR __attribute__((amdgpu_waves_per_eu(2)))
#else
#define RL_RP_PROJECT_ATTR
#endif

// The solver's vector update inside the projection (FB; rl_solver.h, k_minres2_b): the batch is
// MINRES's y' and the projection is wanted of  y_r = y' - coef[v] y_{r-1}.  The tile loader
// forms y_r from the two operands, stores it over y' (every element of the batch is staged by
// exactly one thread of one workgroup) and the block's squared norms come out of the matrix
// cores as the diagonal of one more product of the staged block with itself:
//   nrm[v][run] = sum_{i in run} y_r[v][i]^2   (deterministic: fixed order of rows and waves).
// A frozen system has coef 0: its values are rewritten unchanged, its norm is not read.
struct RpFuse {
    const double* r2;       // y_{r-1}, [nvec][n]
    const double* coef;     // [nvec]
    double* nrm;            // [nvec][nruns]
};
// sum of v over the workgroup's 256 threads in a fixed order (LDS scratch of 256 doubles)
__device__ __forceinline__ double rp_block_sum(double v, double* scr, int tid) {
    scr[tid] = v;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if (tid < h) scr[tid] += scr[tid + h];
        __syncthreads();
    }
    const double r = scr[0];
    __syncthreads();
    return r;
}

// MINRES's P inside the EXPANSION (round 5, RpPFuse).  P's element work per row and system --
// finish iteration r - 1 (w_k, x_k, partial ||x_k||^2), start iteration r (y' = q' / beta_r -
// (beta_r / beta_{r-1}) y_{r-2}, partial alfa_r) -- needs q' = K~ y_{r-1}, which the expansion
// holds in a register: it reads y_{r-1} for the noise term anyway, so the fused kernel reads
// r2, r1, w1, w2, x and writes w1, x, y' -- eight streams -- where expansion + P moved
// 2 + 9 (q' written, then read back with six more operands).  P's scalar chain (sums of the
// partial dot products, plane rotation) runs ONCE per system in a head kernel,
// k_minres2_ph (rl_solver.h), which leaves the coefficients of the element work in pc:
//   pc[v][0] go (0: frozen system, nothing of it is touched), [1] fin (round >= 2),
//   [2] oinv = 1 / beta_{r-1}, [3] oldeps, [4] delta, [5] denom, [6] phi,
//   [7] sinv = 1 / beta_r, [8] coef = beta_r / beta_{r-1}
// The partial sums of alfa_r and ||x_k||^2 leave per workgroup (fixed order: lanes, then
// the four waves): partA / partC [v][gridDim.x].  Same statements as k_minres2_p, same bits.
#define RL_RP_PCW 10
struct RpPFuse {
    const double* pc;       // [nvec][RL_RP_PCW]; nullptr: plain expansion
    double* r1;             // y_{r-2}: read, then receives y'
    double* w1;             // w_{k-2}: read, receives w_k
    const double* w2;       // w_{k-1}
    double* x;
    double* partA;          // [nvec][gridDim.x]
    double* partC;
};
// sums of a and b over the 64 lanes of a wave, valid in lane 0 (fixed order)
__device__ __forceinline__ void rp_wave_sum2(double& a, double& b) {
#if !defined(RL_EMU)
    // lanes >= 32 carry on with b, the others with a: one exchange serves both sums
    const int lane = threadIdx.x & 63;
    const bool hi = (lane & 32) != 0;
    const double mine = hi ? b : a, send = hi ? a : b;
    double v = mine + __shfl_xor(send, 32, 64);
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    a = v;                                  // lanes < 32: sum of a; lanes >= 32: sum of b
    b = __shfl(v, 32, 64);
#endif
}

template <int R, bool FLY, bool FB = false>
__global__ void __launch_bounds__(256) RL_RP_PROJECT_ATTR
k_rp_project(const double* Y, int n, int nvec, const double* __restrict__ F,
             const int* __restrict__ runs, int nruns, double* __restrict__ part,
             int* __restrict__ bump, const int* __restrict__ base, const double* __restrict__ w4,
             int m, const double* __restrict__ beta, RpFuse fz = RpFuse{nullptr, nullptr, nullptr}) {
    static_assert(R <= RL_RP_RMAX, "rank");
    double* Yw = const_cast<double*>(Y);     // (FB only: y_r stored over y')
    // (the solver's round counter: bumped by the first kernel of a round)
    if (bump != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *bump += 1;
    constexpr int TILE = RL_RP_TILE, LD = RL_RP_LD, VG = RL_RP_VG, NG = FB ? RL_RP_NG_FB(R) : RL_RP_NG(R);
    constexpr int NT = (R + 15) / 16;                 // degree tiles of 16
    RL_SMEM(smem);
    double* Fs = reinterpret_cast<double*>(smem);      // [16 NT][LD]
    double* Ys = Fs + (size_t)16 * NT * LD;            // [2][VG][LD]
    double* Yt = Ys + (size_t)2 * VG * LD;             // [TILE]: the lone last vector (below)
    const int tid = threadIdx.x;
    // 1-D launch of 8 ceil(nruns / 8) x nyb workgroups.  The vector blocks of a run read the
    // SAME tiles of F: workgroups b and b + 8 -- the same XCD under the dispatcher's round
    // robin, started together -- take the same run and neighbouring vector blocks, so the
    // second read of a tile finds it in that XCD's L2.  (Only speed depends on the placement.)
    const int nyb = (nvec + NG * VG - 1) / (NG * VG);
    const int bb = blockIdx.x, grp = bb / (8 * nyb), rem = bb - grp * (8 * nyb);
    const int run = grp * 8 + (rem & 7), yb = rem >> 3;
    if (run >= nruns) return;
    const int r0 = runs[3 * run], r1 = runs[3 * run + 1];
    const int vbase = yb * (NG * VG);
    const int nvb = nvec - vbase < NG * VG ? nvec - vbase : NG * VG;
    // A probe batch is N + 1 vectors, N a multiple of 16: its last vector would cost a whole
    // block of 16 (staging, a barrier, 8 NT matrix instructions) for one column.  A LONE last
    // vector takes the vector pipe instead: thread (degree slot t & 63, row part t >> 6)
    // keeps one running sum over the run, 32 multiply-adds per tile out of LDS.
    const bool lone = nvb > 1 && (nvb % VG) == 1;
    const int vlone = vbase + nvb - 1;
    const int ng = (nvb - (lone ? 1 : 0) + VG - 1) / VG;
    double tacc = 0.0, ytr = 0.0, ytr2 = 0.0, tn = 0.0;
#if !defined(RL_EMU)
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    rp_double4 C[NG][NT];
    rp_double4 G[FB ? NG : 1];                         // FB: Gram blocks of the staged vectors
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < NT; ++t) C[g][t] = rp_double4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < (FB ? NG : 1); ++g) G[g] = rp_double4{0.0, 0.0, 0.0, 0.0};
    // (vector of staging value u: (tid >> 7) + 2 u -- wave-uniform: coefficients by scalar loads)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 7);
#else
    double C[NG][2 * NT];                              // entries tid, tid + 256 (, ...) of [16 NT][16]
    double G[NG];                                      // FB: thread jv < 16 sums its vector's squares
    for (int g = 0; g < NG; ++g) {
        G[g] = 0.0;
        for (int t = 0; t < 2 * NT; ++t) C[g][t] = 0.0;
    }
    const int wv = tid >> 7;
#endif
    const double cfl = FB && lone ? fz.coef[vlone] : 0.0;
    // staging registers of one vector block: value idx = tid + 256 u -> vector idx / 128, row idx % 128
    // (one block ahead; two blocks ahead measured the same: 471 vs 457 us)
    constexpr int NU = VG * TILE / 256;
    double yr[1][NU], yr2[1][FB ? NU : 1], cfr[FB ? NU : 1];
    // (staging value u of a thread: vector wv + 2 u of the block -- uniform over the wave, so a
    // vector's base address is a scalar -- and row tid % 128 of the tile for every u: ONE
    // per-lane offset serves all the loads)
    static_assert(TILE == 128, "staging index: idx = tid + 256 u -> vector idx / 128, row idx % 128");
    auto request = [&](int t0, int g, int slot) {
        int row = t0 + (tid & (TILE - 1));
        row = row < r1 ? row : r1 - 1;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            int vu = vbase + g * VG + wv + 2 * u;
            vu = vu < nvec ? vu : nvec - 1;
            yr[slot][u] = (Y + (size_t)vu * n)[row];
            if constexpr (FB) {
                yr2[slot][u] = (fz.r2 + (size_t)vu * n)[row];
                cfr[u] = fz.coef[vu];
            }
        }
    };
    request(r0, 0, 0);
    for (int t0 = r0; t0 < r1; t0 += TILE) {
        // (block 0 of this tile was requested behind the previous tile's last barrier)
        if (lone && tid < TILE) {
            const int row = t0 + tid < r1 ? t0 + tid : r1 - 1;
            ytr = Y[(size_t)vlone * n + row];
            if constexpr (FB) ytr2 = fz.r2[(size_t)vlone * n + row];
        }
        // the tile's F values: degree-major in memory, 128 consecutive rows per degree.
        // (All loads first, unconditional from clamped positions, masked afterwards: a
        // conditional load is a branch with a full memory wait behind it -- the first version
        // paid a round trip per value: 457 us per C5 round)
        if (FLY) {
            // the tile's F values computed from the rows' interpolation entries: waves 0-1 run
            // a row each (four recurrences), waves 2-3 clear the padding degrees
            if (tid < TILE) {
                int row = t0 + tid;
                const bool live = row < r1;
                row = live ? row : r1 - 1;
                RpRow rw;
                rw.start(base[row], m, w4 + (size_t)4 * row);
#pragma unroll 4
                for (int j = 0; j < R; ++j) {
                    const double f = rw.next(beta[j]);
                    Fs[j * LD + tid] = live ? f : 0.0;
                }
            } else {
                for (int idx = R * TILE + tid - TILE; idx < 16 * NT * TILE; idx += 256 - TILE) {
                    const int deg = idx / TILE, rr = idx - deg * TILE;
                    Fs[deg * LD + rr] = 0.0;
                }
            }
        } else {
            constexpr int NF = R * TILE / 256;         // R even: exact
            static_assert((R * TILE) % 256 == 0, "tile of F divides over the threads");
            constexpr int NB = NF <= 12 ? NF : (NF + 1) / 2;      // values per batch (registers)
#pragma unroll
            for (int u0 = 0; u0 < NF; u0 += NB) {
                double fr[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int uu = u0 + u < NF ? u0 + u : NF - 1;
                    const int idx = tid + 256 * uu, deg = idx / TILE, rr = idx - deg * TILE;
                    int row = t0 + rr;
                    row = row < r1 ? row : r1 - 1;
                    fr[u] = F[(size_t)deg * n + row];
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    if (u0 + u < NF) {
                        const int idx = tid + 256 * (u0 + u), deg = idx / TILE, rr = idx - deg * TILE;
                        Fs[deg * LD + rr] = t0 + rr < r1 ? fr[u] : 0.0;
                    }
                }
            }
            // degrees R .. 16 NT - 1 of the last degree tile: zero (written once would do; the
            // tile is small)
            for (int idx = R * TILE + tid; idx < 16 * NT * TILE; idx += 256) {
                const int deg = idx / TILE, rr = idx - deg * TILE;
                Fs[deg * LD + rr] = 0.0;
            }
        }
        if (lone && tid < TILE) {
            const bool live = t0 + tid < r1;
            double yl = ytr;
            if constexpr (FB) {
                yl = ytr - cfl * ytr2;
                if (live) {
                    Yw[(size_t)vlone * n + t0 + tid] = yl;
                    tn = fma(yl, yl, tn);
                }
            }
            Yt[tid] = live ? yl : 0.0;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g < ng) {
                double* yb = Ys + (size_t)(g & 1) * VG * LD;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int jv = wv + 2 * u, rr = tid & (TILE - 1);
                    const bool live = vbase + g * VG + jv < nvec && t0 + rr < r1;
                    double yv_ = yr[0][u];
                    if constexpr (FB) {
                        yv_ = yr[0][u] - cfr[u] * yr2[0][u];
                        if (live) (Yw + (size_t)(vbase + g * VG + jv) * n)[t0 + rr] = yv_;
                    }
                    yb[jv * LD + rr] = live ? yv_ : 0.0;
                }
                __syncthreads();
                if (g + 1 < ng) request(t0, g + 1, 0);
                else if (t0 + TILE < r1) request(t0 + TILE, 0, 0);    // next tile's first block
                if (g == 0 && lone) {
                    const int deg = tid & 63, p0 = (tid >> 6) * (TILE / 4);
                    if (deg < R) {
                        const double* fr_ = Fs + deg * LD + p0;
#pragma unroll 2
                        for (int rr = 0; rr < TILE / 4; ++rr) tacc = fma(fr_[rr], Yt[p0 + rr], tacc);
                    }
                }
#if !defined(RL_EMU)
                const double* fa = Fs + li * LD + 32 * wave + lk;
                const double* yv = yb + li * LD + 32 * wave + lk;
#pragma unroll
                for (int s = 0; s < TILE / 16; ++s) {          // 8 steps of 4 rows per wave
                    const double b = yv[4 * s];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        C[g][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[(size_t)16 * t * LD + 4 * s],
                                                                       b, C[g][t], 0, 0, 0);
                    // (the operand register of B is also that of A for the same (vector, row))
                    if constexpr (FB) G[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, G[g], 0, 0, 0);
                }
#else
                for (int t = 0; t < 2 * NT; ++t) {
                    const int e = tid + 256 * t;
                    if (e < 16 * NT * 16) {
                        const int deg = e / 16, jv = e - deg * 16;
                        double sum = C[g][t];
                        for (int rr = 0; rr < TILE; ++rr) sum = fma(Fs[deg * LD + rr], yb[jv * LD + rr], sum);
                        C[g][t] = sum;
                    }
                }
                if (FB && tid < VG) {
                    double sum = G[g];
                    for (int rr = 0; rr < TILE; ++rr) sum = fma(yb[tid * LD + rr], yb[tid * LD + rr], sum);
                    G[g] = sum;
                }
#endif
            }
        }
        __syncthreads();        // every read of Fs and of both Y tiles is done before the next tile
    }
    // the lone last vector: its four row parts summed through LDS
    if (lone) {
        double* sl = Ys;                                // [4][64]
        sl[tid] = tacc;
        __syncthreads();
        if (tid < R)
            part[((size_t)run * nvec + vlone) * R + tid] =
                (sl[tid] + sl[64 + tid]) + (sl[128 + tid] + sl[192 + tid]);
        __syncthreads();
    }
    if constexpr (FB) {
        // squared norms: the lone vector's by a fixed-order sum over the workgroup, a block's
        // from the diagonal of its Gram block (lane (i, k) holds entry (k + 4 reg, i): the
        // diagonal of vector i sits in lane i + 16 (i % 4), register i / 4), the four waves'
        // row parts summed through LDS
        if (lone) {
            const double t = rp_block_sum(tn, Ys, tid);
            if (tid == 0) fz.nrm[(size_t)vlone * nruns + run] = t;
        }
#if !defined(RL_EMU)
        double* gs = Ys;                                // [NG][4 waves][16]
#pragma unroll
        for (int g = 0; g < NG; ++g)
            if (g < ng && (li & 3) == lk) gs[(g * 4 + wave) * 16 + li] = G[g][li >> 2];
        __syncthreads();
        if (tid < 16 * NG) {
            const int g = tid >> 4, jv = tid & 15, v = vbase + g * VG + jv;
            if (g < ng && v < nvec)
                fz.nrm[(size_t)v * nruns + run] = (gs[(g * 4 + 0) * 16 + jv] + gs[(g * 4 + 1) * 16 + jv]) +
                                                  (gs[(g * 4 + 2) * 16 + jv] + gs[(g * 4 + 3) * 16 + jv]);
        }
        __syncthreads();
#else
        for (int g = 0; g < ng; ++g)
            if (tid < VG && vbase + g * VG + tid < nvec)
                fz.nrm[(size_t)(vbase + g * VG + tid) * nruns + run] = G[g];
#endif
    }
    // results: the four waves' blocks summed through LDS (the Y tiles' space), then written
#if !defined(RL_EMU)
    double* scr = Ys;                                   // [4 waves][16 NT][16]
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g < ng) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    scr[((size_t)wave * 16 * NT + 16 * t + RL_RP_DROW(lane, r)) * 16 + li] = C[g][t][r];
            __syncthreads();
            for (int e = tid; e < 16 * NT * 16; e += 256) {
                const int deg = e / 16, jv = e - deg * 16, v = vbase + g * VG + jv;
                if (deg < R && v < nvec) {
                    const double s01 = scr[e] + scr[(size_t)16 * NT * 16 + e];
                    const double s23 = scr[(size_t)2 * 16 * NT * 16 + e] + scr[(size_t)3 * 16 * NT * 16 + e];
                    part[((size_t)run * nvec + v) * R + deg] = s01 + s23;
                }
            }
            __syncthreads();
        }
    }
#else
    for (int g = 0; g < ng; ++g)
        for (int t = 0; t < 2 * NT; ++t) {
            const int e = tid + 256 * t;
            if (e < 16 * NT * 16) {
                const int deg = e / 16, jv = e - deg * 16, v = vbase + g * VG + jv;
                if (deg < R && v < nvec) part[((size_t)run * nvec + v) * R + deg] = C[g][t];
            }
        }
#endif
}

// ---------------------------------------------------------------------------
// k_rp_project1<R>: the same for ONE block of at most 16 vectors plus a lone last one -- the
// 17 vectors of one rank's share of an 8-way probe split.  With a single block per tile the
// big kernel's tile is all latency (loads of F and Y, a barrier, 16 matrix instructions):
// here the NEXT tile's loads are in flight while the current tile is multiplied, and the
// smaller footprint (one Y tile, 8 accumulator registers) admits three workgroups per CU.
// C5, 17 vectors: 72 us against the general kernel's 84 (same box); with F computed from the
// interpolation entries (FLY) 74 -- the table stays.
//   grid (nruns)   block 256   LDS: F tile [16 NT][LD] + Y tile [16][LD] + lone [TILE]
// ---------------------------------------------------------------------------
#if !defined(RL_EMU)
// (three workgroups per CU also with the solver's update inside, while the rank allows)
#define RL_RP_PROJECT1_ATTR(R, FB) __attribute__((amdgpu_waves_per_eu(((FB) && (R) <= 24) ? 3 : 1)))
#else
#define RL_RP_PROJECT1_ATTR(R, FB)
#endif
template <int R, bool FLY, bool FB = false>
__global__ void __launch_bounds__(256) RL_RP_PROJECT1_ATTR(R, FB)
k_rp_project1(const double* Y, int n, int nvec, const double* __restrict__ F,
              const int* __restrict__ runs, double* __restrict__ part, int* __restrict__ bump,
              const int* __restrict__ base, const double* __restrict__ w4, int m,
              const double* __restrict__ beta, RpFuse fz = RpFuse{nullptr, nullptr, nullptr}) {
    static_assert(R <= RL_RP_RMAX, "rank");
    double* Yw = const_cast<double*>(Y);     // (FB: y_r stored over y', as in k_rp_project)
    const int nruns = gridDim.x;
    if (bump != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *bump += 1;
    constexpr int TILE = RL_RP_TILE, LD = RL_RP_LD, VG = RL_RP_VG;
    constexpr int NT = (R + 15) / 16, NF = R * TILE / 256, NU = VG * TILE / 256;
    RL_SMEM(smem);
    double* Fs = reinterpret_cast<double*>(smem);      // [16 NT][LD]
    double* Ys = Fs + (size_t)16 * NT * LD;            // [VG][LD]
    double* Yt = Ys + (size_t)VG * LD;                 // [TILE]
    const int tid = threadIdx.x;
    const int run = blockIdx.x, r0 = runs[3 * run], r1 = runs[3 * run + 1];
    const bool lone = nvec > 1 && (nvec % VG) == 1;
    const int vlone = nvec - 1;
    double tacc = 0.0, tn = 0.0;
#if !defined(RL_EMU)
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    rp_double4 C[NT];
    rp_double4 G = rp_double4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NT; ++t) C[t] = rp_double4{0.0, 0.0, 0.0, 0.0};
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 7);
#else
    double C[2 * NT], G = 0.0;
    for (int t = 0; t < 2 * NT; ++t) C[t] = 0.0;
    const int wv = tid >> 7;
#endif
    // (FB: the vectors' coefficients do not change over the run)
    double cfr[FB ? VG * TILE / 256 : 1];
    if constexpr (FB) {
#pragma unroll
        for (int u = 0; u < VG * TILE / 256; ++u) {
            const int vu = wv + 2 * u;
            cfr[u] = fz.coef[vu < nvec ? vu : nvec - 1];
        }
    }
    const double cfl = FB && lone ? fz.coef[vlone] : 0.0;
    // the padding degrees of the last degree tile never change
    for (int idx = R * TILE + tid; idx < 16 * NT * TILE; idx += 256) {
        const int deg = idx / TILE, rr = idx - deg * TILE;
        Fs[deg * LD + rr] = 0.0;
    }
    double fr[FLY ? 4 : NF], yr[NU], yr2[FB ? NU : 1], ytr = 0.0, ytr2 = 0.0;
    int fb = 0;
    auto load_tile = [&](int t0) {
        if (FLY) {
            // (the row's interpolation entry instead of its R values of F)
            const int row = t0 + (tid & (TILE - 1)) < r1 ? t0 + (tid & (TILE - 1)) : r1 - 1;
            fb = base[row];
#pragma unroll
            for (int e = 0; e < 4; ++e) fr[e] = w4[(size_t)4 * row + e];
        } else {
#pragma unroll
            for (int u = 0; u < (FLY ? 4 : NF); ++u) {
                const int idx = tid + 256 * u, deg = idx / TILE, rr = idx - deg * TILE;
                int row = t0 + rr;
                row = row < r1 ? row : r1 - 1;
                fr[u] = F[(size_t)deg * n + row];
            }
        }
        int yrow = t0 + (tid & (TILE - 1));
        yrow = yrow < r1 ? yrow : r1 - 1;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int jv = wv + 2 * u;                      // (uniform: scalar base address)
            const int v = jv < nvec ? jv : nvec - 1;
            yr[u] = (Y + (size_t)v * n)[yrow];
            if constexpr (FB) yr2[u] = (fz.r2 + (size_t)v * n)[yrow];
        }
        if (tid < TILE) {
            const int row = t0 + tid < r1 ? t0 + tid : r1 - 1;
            ytr = Y[(size_t)vlone * n + row];
            if constexpr (FB) ytr2 = fz.r2[(size_t)vlone * n + row];
        }
    };
    load_tile(r0);
    const int nfull = lone ? nvec - 1 : nvec;           // vectors of the matrix-core block
    for (int t0 = r0; t0 < r1; t0 += TILE) {
        if (FLY) {
            // waves 0-1: a row each, four recurrences; degrees j = 0 .. R - 1 in order
            if (tid < TILE) {
                RpRow rw;
                rw.start(fb, m, fr);
                const bool live = t0 + tid < r1;
#pragma unroll 4
                for (int j = 0; j < R; ++j) {
                    const double f = rw.next(beta[j]);
                    Fs[j * LD + tid] = live ? f : 0.0;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < (FLY ? 4 : NF); ++u) {
                const int idx = tid + 256 * u, deg = idx / TILE, rr = idx - deg * TILE;
                Fs[deg * LD + rr] = t0 + rr < r1 ? fr[u] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int jv = wv + 2 * u, rr = tid & (TILE - 1);
            const bool live = jv < nfull && t0 + rr < r1;
            double yv_ = yr[u];
            if constexpr (FB) {
                yv_ = yr[u] - cfr[u] * yr2[u];
                if (live) (Yw + (size_t)jv * n)[t0 + rr] = yv_;
            }
            Ys[jv * LD + rr] = live ? yv_ : 0.0;
        }
        if (tid < TILE) {
            const bool live = lone && t0 + tid < r1;
            double yl = ytr;
            if constexpr (FB) {
                yl = ytr - cfl * ytr2;
                if (live) {
                    Yw[(size_t)vlone * n + t0 + tid] = yl;
                    tn = fma(yl, yl, tn);
                }
            }
            Yt[tid] = live ? yl : 0.0;
        }
        __syncthreads();
        // (FB: the last tile's request must not repeat -- its rows have just been rewritten;
        // a request past the run would read another workgroup's rows: harmless, never used --
        // but stay inside the run: repeat the tile only when FB is off)
        if (!FB || t0 + TILE < r1) load_tile(t0 + TILE < r1 ? t0 + TILE : t0);
        if (lone) {
            const int deg = tid & 63, p0 = (tid >> 6) * (TILE / 4);
            if (deg < R) {
                const double* fq = Fs + deg * LD + p0;
#pragma unroll 2
                for (int rr = 0; rr < TILE / 4; ++rr) tacc = fma(fq[rr], Yt[p0 + rr], tacc);
            }
        }
#if !defined(RL_EMU)
        {
            const double* fa = Fs + li * LD + 32 * wave + lk;
            const double* yv = Ys + li * LD + 32 * wave + lk;
#pragma unroll
            for (int s = 0; s < TILE / 16; ++s) {
                const double b = yv[4 * s];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    C[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[(size_t)16 * t * LD + 4 * s], b,
                                                                C[t], 0, 0, 0);
                if constexpr (FB) G = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, G, 0, 0, 0);
            }
        }
#else
        for (int t = 0; t < 2 * NT; ++t) {
            const int e = tid + 256 * t;
            if (e < 16 * NT * 16) {
                const int deg = e / 16, jv = e - deg * 16;
                double sum = C[t];
                for (int rr = 0; rr < TILE; ++rr) sum = fma(Fs[deg * LD + rr], Ys[jv * LD + rr], sum);
                C[t] = sum;
            }
        }
        if (FB && tid < VG) {
            double sum = G;
            for (int rr = 0; rr < TILE; ++rr) sum = fma(Ys[tid * LD + rr], Ys[tid * LD + rr], sum);
            G = sum;
        }
#endif
        __syncthreads();
    }
    if (lone) {
        double* sl = Ys;
        sl[tid] = tacc;
        __syncthreads();
        if (tid < R)
            part[((size_t)run * nvec + vlone) * R + tid] =
                (sl[tid] + sl[64 + tid]) + (sl[128 + tid] + sl[192 + tid]);
        __syncthreads();
    }
    if constexpr (FB) {
        if (lone) {
            const double t = rp_block_sum(tn, Ys, tid);
            if (tid == 0) fz.nrm[(size_t)vlone * nruns + run] = t;
        }
#if !defined(RL_EMU)
        double* gs = Ys;                                 // [4 waves][16]
        if ((li & 3) == lk) gs[wave * 16 + li] = G[li >> 2];
        __syncthreads();
        if (tid < nfull)
            fz.nrm[(size_t)tid * nruns + run] = (gs[tid] + gs[16 + tid]) + (gs[32 + tid] + gs[48 + tid]);
        __syncthreads();
#else
        if (tid < nfull) fz.nrm[(size_t)tid * nruns + run] = G;
#endif
    }
#if !defined(RL_EMU)
    {
        double* scr = Fs;                                // [4 waves][16 NT][16]: 16 KB at NT = 2
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                scr[((size_t)wave * 16 * NT + 16 * t + RL_RP_DROW(lane, r)) * 16 + li] = C[t][r];
        __syncthreads();
        for (int e = tid; e < 16 * NT * 16; e += 256) {
            const int deg = e / 16, jv = e - deg * 16;
            if (deg < R && jv < nfull) {
                const double s01 = scr[e] + scr[(size_t)16 * NT * 16 + e];
                const double s23 = scr[(size_t)2 * 16 * NT * 16 + e] + scr[(size_t)3 * 16 * NT * 16 + e];
                part[((size_t)run * nvec + jv) * R + deg] = s01 + s23;
            }
        }
    }
#else
    for (int t = 0; t < 2 * NT; ++t) {
        const int e = tid + 256 * t;
        if (e < 16 * NT * 16) {
            const int deg = e / 16, jv = e - deg * 16;
            if (deg < R && jv < nfull) part[((size_t)run * nvec + jv) * R + deg] = C[t];
        }
    }
#endif
}

// ---------------------------------------------------------------------------
// k_rp_expand<R>: Q[v][i] = sum_j F[j][i] Zhat[v D + d(i)][j]  (+ diag[i] X2[v][i]).
//   grid (ceil(n / 256))   block 256
// A thread owns a data row and keeps its R values of F in registers for all the vectors;
// a wave's 64 rows lie in one output except at an output border, so a vector's coefficient
// row is wave-uniform and comes through scalar loads (the structure of k_lr_expand; the wave
// at a border loads per lane).  Workgroups start at different vectors (v0 = 7 block mod nvec)
// so that the chip does not write the same few vectors in lockstep (k_lr_expand's finding).
//   out_end [D]: end row (sorted order) of each output
// ---------------------------------------------------------------------------
__device__ __forceinline__ int rp_output_of(const int* __restrict__ out_end, int D, int i) {
    int lo = 0, hi = D - 1;                 // first d with i < out_end[d]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (i < out_end[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

template <int R, bool FLY, bool FP = false, bool NOISY = true>
__global__ void __launch_bounds__(256)
k_rp_expand(const double* __restrict__ Zhat, const double* __restrict__ F, int n, int nvec, int D,
            const int* __restrict__ out_end, double* __restrict__ Q,
            const double* __restrict__ diag, const double* __restrict__ X2, int stagger,
            const int* __restrict__ base, const double* __restrict__ w4, int m,
            const double* __restrict__ beta, RpPFuse pf = RpPFuse{nullptr, nullptr, nullptr, nullptr,
                                                                  nullptr, nullptr, nullptr}) {
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    const int ic = i < n ? i : n - 1;
    double p[R];
    if (FLY) {
        RpRow rw;
        rw.start(base[ic], m, w4 + (size_t)4 * ic);
#pragma unroll
        for (int j = 0; j < R; ++j) p[j] = rw.next(beta[j]);
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) p[j] = F[(size_t)j * n + ic];
    }
    const double dg = diag != nullptr ? diag[ic] : 0.0;
    const int wf = blockIdx.x * 256 + (tid & ~63);
    const int wl = wf + 63 < n ? wf + 63 : n - 1;
    const int dfirst = RL_LR_UNIFORM(rp_output_of(out_end, D, wf < n ? wf : n - 1));
    const int dlast = RL_LR_UNIFORM(rp_output_of(out_end, D, wl));
    const int dmine = dfirst == dlast ? dfirst : rp_output_of(out_end, D, ic);
    // (Measured and dropped, round 5: a second grid dimension over groups of 4 .. 64 vectors, so
    // that the chip works inside a few vectors' pages at a time -- 2.87-3.20 ms per C5 round
    // against 2.83 plain, 3.64-4.18 against 3.76 with P inside: placement was not the limit.)
    const int cnt = nvec;
    const int voff = (int)(((unsigned)stagger * blockIdx.x) % (unsigned)cnt);
    if constexpr (FP) {
        // MINRES's P inside (RpPFuse above): q' stays in a register
        RL_SMEM(smem);
        double* ws = reinterpret_cast<double*>(smem);        // [nvec][4 waves][2]  (+ 256: emulator)
#if !defined(RL_EMU)
        const int lane = tid & 63, wave = tid >> 6;
#endif
        const bool live = i < n;
        const bool uni = dfirst == dlast;
        // The operands of a system (y_{r-1}, y_{r-2}, w_{k-2}, w_{k-1}, x: five loads per thread)
        // are requested TWO systems ahead.  A frozen system moves nothing: its request goes to
        // the addresses of the last live one (cache hits).
        // (What the first versions of this loop lost, 2.40-2.96 ms per C5 round against 1.63 +
        // 0.24 for P and the plain expansion, was code generation, not the order of the walk:
        // this kernel stores through pointers the compiler cannot tell from pc and Zhat, so it
        // fetched the wave-uniform coefficients with a vector load per lane and waited for
        // EVERY outstanding access around them; rotating the register sets by copies made it
        // wait for the newest request each system; per-lane 64-bit addresses reused the
        // requests' registers -- another full wait.  With the coefficients in the constant
        // address space (scalar loads), a ring of three register sets in a loop unrolled by
        // three and buffer accesses (scalar base + one offset register) no wait of the loop
        // drains the queue: 1.52 ms, the round 2.30 against 2.62.)
        struct Ops {
            double r2, r1, w1, w2, x;
        };
        auto vat = [&](int it) {                 // the it-th system this workgroup visits
            const int vv = voff + it;
            return vv < cnt ? vv : vv - cnt;
        };
        // (a system's base address is wave-uniform and the row's byte offset fits 32 bits -- the
        // host takes this path for n < 2^28 only: buffer accesses, rl_device.h -- scalar base,
        // one offset register for the whole loop, rows past n read 0 and store nothing)
        const unsigned off8 = (unsigned)i * 8u, nb8 = (unsigned)n * 8u;
        int vlive = 0;                           // (some system's rows: any valid address)
        auto request = [&](int it) {
            int vv = it < cnt ? vat(it) : vlive;
            if (RL_KCONST(pf.pc)[(size_t)vv * RL_RP_PCW] == 0.0) vv = vlive;
            else vlive = vv;
            const size_t at = (size_t)vv * n;
            return Ops{rl_row_load(X2 + at, nb8, off8), rl_row_load(pf.r1 + at, nb8, off8),
                       rl_row_load(pf.w1 + at, nb8, off8), rl_row_load(pf.w2 + at, nb8, off8),
                       rl_row_load(pf.x + at, nb8, off8)};
        };
        // (a ring of three register sets, the loop unrolled by three: rotating two sets by
        // copies made the compiler wait for the loads just issued before every copy)
        Ops ring[3];
        ring[0] = request(0);
        ring[1] = request(1);
        auto step = [&](int it, const Ops& cur) {
            const int vv = vat(it);
            rl_kconst cp = RL_KCONST(pf.pc) + (size_t)vv * RL_RP_PCW;       // wave-uniform: scalar loads
            // (the system's nine coefficients in one batch, not four dependent round trips)
            double c[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) c[j] = cp[j];
#if !defined(RL_EMU)
            __builtin_amdgcn_sched_barrier(0);
#endif
            double accA = 0.0, accC = 0.0;
            if (c[0] != 0.0) {
                const size_t at = (size_t)vv * n;
                double ev = 0.0, od = 0.0;
                if (uni) {
                    // (ONE batch of scalar loads, pinned ahead of the multiply-adds: k_lr_expand)
                    rl_kconst z = RL_KCONST(Zhat) + ((size_t)vv * D + dfirst) * R;
                    double zz[R];
#pragma unroll
                    for (int j = 0; j < R; ++j) zz[j] = z[j];
#if !defined(RL_EMU)
                    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                    for (int j = 0; j + 1 < R; j += 2) {
                        ev = fma(zz[j], p[j], ev);
                        od = fma(zz[j + 1], p[j + 1], od);
                    }
                } else {
                    const double* z = Zhat + ((size_t)vv * D + dmine) * R;
#pragma unroll
                    for (int j = 0; j + 1 < R; j += 2) {
                        ev = fma(z[j], p[j], ev);
                        od = fma(z[j + 1], p[j + 1], od);
                    }
                }
                double q = ev + od;
                if (diag != nullptr) q = fma(dg, cur.r2, q);
                // the element work of k_minres2_p, statement by statement
                if (c[1] != 0.0) {
                    const double wn = (cur.r1 * c[2] - c[3] * cur.w1 - c[4] * cur.w2) * c[5];
                    const double xi = cur.x + c[6] * wn;
                    rl_row_store(pf.w1 + at, nb8, off8, wn);
                    rl_row_store(pf.x + at, nb8, off8, xi);
                    accC = live ? xi * xi : 0.0;
                }
                const double yi = q * c[7] - c[8] * cur.r1;
                rl_row_store(pf.r1 + at, nb8, off8, yi);
                accA = live ? (cur.r2 * c[7]) * yi : 0.0;
            }
#if defined(RL_EMU)
            {
                double* red = ws + (size_t)nvec * 8;
                accA = rp_block_sum(accA, red, tid);
                accC = rp_block_sum(accC, red, tid);
                if (tid == 0) {
                    ws[(size_t)it * 8] = accA;
                    ws[(size_t)it * 8 + 1] = accC;
                }
            }
#else
            rp_wave_sum2(accA, accC);
            if (lane == 0) {
                ws[((size_t)it * 4 + wave) * 2] = accA;
                ws[((size_t)it * 4 + wave) * 2 + 1] = accC;
            }
#endif
        };
        for (int it0 = 0; it0 < cnt; it0 += 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (it0 + k < cnt) {
                    ring[(k + 2) % 3] = request(it0 + k + 2);
                    step(it0 + k, ring[k]);
                }
            }
        }
        __syncthreads();
        for (int it = tid; it < cnt; it += 256) {
            const int vv = vat(it);
#if defined(RL_EMU)
            const double a = ws[(size_t)it * 8], cc = ws[(size_t)it * 8 + 1];
#else
            const double* w = ws + (size_t)it * 8;
            const double a = (w[0] + w[2]) + (w[4] + w[6]), cc = (w[1] + w[3]) + (w[5] + w[7]);
#endif
            pf.partA[(size_t)vv * gridDim.x + blockIdx.x] = a;
            pf.partC[(size_t)vv * gridDim.x + blockIdx.x] = cc;
        }
        return;
    }
    // Plain expansion.  The noise term's operand X2 is requested two vectors ahead through a
    // ring of three registers (loop unrolled by three, as above) and every access is a buffer
    // access: before, a vector's load was waited for with everything else outstanding -- the
    // previous vector's store included, one store in flight per wave.
    // Two instantiations of the walk (not one loop with a select on the pointer: the compiler
    // then loads the coefficients per lane in both cases -- twelve dependent 16-byte vector
    // loads per vector, 800 us per C5 round instead of the scalar loads' 2xx).
    const unsigned off8 = (unsigned)i * 8u, nb8 = (unsigned)n * 8u;
    constexpr bool noisy = NOISY;     // (diag != nullptr, known at compile time: a load behind a
                                      // run-time condition costs the exact wait counts)
    auto vat = [&](int it) {                     // the it-th vector this workgroup visits
        const int vv = voff + (it < cnt ? it : cnt - 1);
        return vv < cnt ? vv : vv - cnt;
    };
    auto walk = [&](auto zrow) {
        double xr[3] = {0.0, 0.0, 0.0};
        if constexpr (noisy) {
            xr[0] = rl_row_load(X2 + (size_t)vat(0) * n, nb8, off8);
            xr[1] = rl_row_load(X2 + (size_t)vat(1) * n, nb8, off8);
        }
        for (int it0 = 0; it0 < cnt; it0 += 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (it0 + k < cnt) {
                    const int vv = vat(it0 + k);
                    if constexpr (noisy) xr[(k + 2) % 3] = rl_row_load(X2 + (size_t)vat(it0 + k + 2) * n, nb8, off8);
                    const auto z = zrow(vv);
                    double zz[R];
#pragma unroll
                    for (int j = 0; j < R; ++j) zz[j] = z[j];
#if !defined(RL_EMU)
                    __builtin_amdgcn_sched_barrier(0);       // (one batch of loads: k_lr_expand)
#endif
                    double ev = 0.0, od = 0.0;
#pragma unroll
                    for (int j = 0; j + 1 < R; j += 2) {
                        ev = fma(zz[j], p[j], ev);
                        od = fma(zz[j + 1], p[j + 1], od);
                    }
                    double acc = ev + od;
                    if constexpr (noisy) acc = fma(dg, xr[k], acc);
                    rl_row_store(Q + (size_t)vv * n, nb8, off8, acc);
                }
            }
        }
    };
    if (dfirst == dlast)        // wave-uniform coefficient rows: scalar loads
        walk([&](int vv) { return RL_KCONST(Zhat) + ((size_t)vv * D + dfirst) * R; });
    else
        walk([&](int vv) { return Zhat + ((size_t)vv * D + dmine) * R; });
}

// ---------------------------------------------------------------------------
// k_spmv_w_staged_p<VB, XPT>: MINRES's P inside the interpolation product W g (+ eps (.) y) of
// a round whose operator is NOT in the row-polynomial form (filter and transform forms:
// W^T, grid product, W).  k_spmv_w_staged (rl_kernels.h) with the element work of k_minres2_p on
// the q' = (W g)[row] + eps y_{r-1}[row] a thread holds in a register for each of a group's VB
// systems -- the kernel reads g (staged through LDS as before), y_{r-1}, y_{r-2}, w_{k-2},
// w_{k-1}, x and writes w_k, x, y': nine streams where W and P moved 2 + 9.  The scalar chain
// is k_minres2_ph's (pc: RpPFuse above), the partial sums of alfa and ||x||^2 leave per
// (system, row block): partA / partC [v][row blocks].  A frozen system's descriptors are
// EMPTY (zero bytes): its loads return 0 without a fetch and nothing of it is stored.
//   grid / block / LDS as k_spmv_w_staged + vgroups VB 8 doubles (+ 256: emulator)
// ---------------------------------------------------------------------------
template <int VB, int XPT>
__global__ void __launch_bounds__(RL_THREADS)
k_spmv_w_staged_p(const int* __restrict__ base, const double* __restrict__ w4, int nrows, int ncols,
                  int nvec, const double* __restrict__ G, const double* __restrict__ diag,
                  const double* X2, int xcap, int vgroups, RpPFuse pf) {
    RL_SMEM(smem);
    double* xs = reinterpret_cast<double*>(smem);          // [VB][xcap]
    double* ws = xs + (size_t)VB * xcap;                   // [vgroups VB][4 waves][2]
    const int tid = threadIdx.x, nthr = blockDim.x;
#if !defined(RL_EMU)
    const int lane = tid & 63, wave = tid >> 6;
#endif
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
    const bool live = row <= rl;
    const int b = base[rowc];
    double w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = w4[(size_t)4 * rowc + e];
    const double dg = diag != nullptr ? diag[rowc] : 0.0;
    // (rows past the workgroup's last one lie past the vectors' end or in the next workgroup:
    // their accesses get an offset past every descriptor)
    const unsigned nb8 = (unsigned)nrows * 8u, off8 = live ? (unsigned)row * 8u : nb8;
    double xr[VB][XPT];
    auto request = [&](int grp) {
        const int v0 = (g0 + grp) * VB;
        const int nv = nvec - v0 < VB ? nvec - v0 : VB;
#pragma unroll
        for (int j = 0; j < VB; ++j) {
            const double* g = G + (size_t)(v0 + (j < nv ? j : 0)) * ncols;
#pragma unroll
            for (int u = 0; u < XPT; ++u) {
                // (positions past the range -- up to xcap - len of them -- repeat its last
                // element: a hit in the same line instead of a fetch of the neighbour's range;
                // the PMC counters showed the staged kernels fetching 1.2-2.3x their operand)
                int c = c0 + tid + u * nthr;
                c = c < c1 ? c : c1 - 1;
                c = c < ncols ? c : ncols - 1;
                c = c > 0 ? c : 0;
                xr[j][u] = g[c];
            }
        }
    };
    request(0);
    const double* xrow = xs + (b - c0);
    for (int grp = 0; grp < ng; ++grp) {
        const int v0 = (g0 + grp) * VB;
        const int nv = nvec - v0 < VB ? nvec - v0 : VB;
        // the operands of the group's systems: requested here, used after the sums out of LDS
        double o2[VB], o1[VB], ow1[VB], ow2[VB], ox[VB];
#pragma unroll
        for (int j = 0; j < VB; ++j) {
            const int v = v0 + (j < nv ? j : 0);
            const bool go = j < nv && RL_KCONST(pf.pc)[(size_t)v * RL_RP_PCW] != 0.0;
            const unsigned nb = go ? nb8 : 0u;
            const size_t at = (size_t)v * nrows;
            o2[j] = rl_row_load(X2 + at, nb, off8);
            o1[j] = rl_row_load(pf.r1 + at, nb, off8);
            ow1[j] = rl_row_load(pf.w1 + at, nb, off8);
            ow2[j] = rl_row_load(pf.w2 + at, nb, off8);
            ox[j] = rl_row_load(pf.x + at, nb, off8);
        }
#pragma unroll
        for (int j = 0; j < VB; ++j)
#pragma unroll
            for (int u = 0; u < XPT; ++u) {
                const int i = tid + u * nthr;
                if (i < len) xs[(size_t)j * xcap + i] = xr[j][u];
            }
        __syncthreads();
        if (grp + 1 < ng) request(grp + 1);
#pragma unroll
        for (int j = 0; j < VB; ++j) {
            const int v = v0 + (j < nv ? j : 0);
            rl_kconst cp = RL_KCONST(pf.pc) + (size_t)v * RL_RP_PCW;
            double c[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) c[k] = cp[k];
            double accA = 0.0, accC = 0.0;
            if (j < nv && c[0] != 0.0) {
                const size_t at = (size_t)v * nrows;
                double q = 0.0;
#pragma unroll
                for (int e = 0; e < 4; ++e) q = fma(w[e], xrow[(size_t)j * xcap + e], q);
                if (diag != nullptr) q = fma(dg, o2[j], q);
                // the element work of k_minres2_p, statement by statement
                if (c[1] != 0.0) {
                    const double wn = (o1[j] * c[2] - c[3] * ow1[j] - c[4] * ow2[j]) * c[5];
                    const double xi = ox[j] + c[6] * wn;
                    rl_row_store(pf.w1 + at, nb8, off8, wn);
                    rl_row_store(pf.x + at, nb8, off8, xi);
                    accC = live ? xi * xi : 0.0;
                }
                const double yi = q * c[7] - c[8] * o1[j];
                rl_row_store(pf.r1 + at, nb8, off8, yi);
                accA = live ? (o2[j] * c[7]) * yi : 0.0;
            }
            const int slot = grp * VB + j;
#if defined(RL_EMU)
            {
                double* red = ws + (size_t)vgroups * VB * 8;
                accA = rp_block_sum(accA, red, tid);
                accC = rp_block_sum(accC, red, tid);
                if (tid == 0) {
                    ws[(size_t)slot * 8] = accA;
                    ws[(size_t)slot * 8 + 1] = accC;
                }
            }
#else
            rp_wave_sum2(accA, accC);
            if (lane == 0) {
                ws[((size_t)slot * 4 + wave) * 2] = accA;
                ws[((size_t)slot * 4 + wave) * 2 + 1] = accC;
            }
#endif
        }
        __syncthreads();
    }
    const int nbx = gridDim.x;
    for (int it = tid; it < ng * VB; it += nthr) {
        const int v = g0 * VB + it;
        if (v < nvec) {
#if defined(RL_EMU)
            const double a = ws[(size_t)it * 8], cc = ws[(size_t)it * 8 + 1];
#else
            const double* p = ws + (size_t)it * 8;
            const double a = (p[0] + p[2]) + (p[4] + p[6]), cc = (p[1] + p[3]) + (p[5] + p[7]);
#endif
            pf.partA[(size_t)v * nbx + bx] = a;
            pf.partC[(size_t)v * nbx + bx] = cc;
        }
    }
}
