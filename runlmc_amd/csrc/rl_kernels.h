// Kernels of the grid operator  K_UU = sum_q B_q (x) T_q  and of the SKI
// wrapper  K~ = W K_UU W^T + diag(eps).
//
// Reference behaviour being replaced (vlad17/runlmc):
//   runlmc/linalg/bttb.py:106-120,144-148   circulant embed + rfftn / irfftn
//   runlmc/linalg/kronecker.py:39-46         (B (x) T) x
//   runlmc/linalg/sum_matrix.py:31-32        sum over q
//   runlmc/approx/ski.py:13-16               W . , W^T .
//   runlmc/linalg/diag.py:24-25              eps * x
//
// Device formulation:
//   * two real right-hand sides ride one complex transform (x1 + i x2): the
//     circulant spectra and the coregionalisation mix are real, so the
//     operator commutes with that packing and no real-FFT untangling exists;
//   * the length-L complex FFT is split L = N1 x N2 (four-step): k_cols_fwd
//     does the N1-point column transforms for a tile of columns in LDS and the
//     inter-step twiddle, k_rows_mix does the N2-point row transforms for ALL
//     D outputs of a few rows, applies the real D x D mix at every frequency in
//     place, and runs the adjoint row transforms, k_cols_inv does the adjoint
//     column transforms and the crop;
//   * frequencies stay in the scrambled order the in-place passes produce; the
//     spectra are made by the same forward graph (k_rows_spec).
#pragma once
#include "rl_fft.h"

#define RL_THREADS 256

// W_L^e from two short tables: e = hi * 2^shift + lo
struct TwiddleL {
    const cplx* lo;
    const cplx* hi;
    int shift;
    int mask;
};
__device__ __forceinline__ cplx twiddle_L(const TwiddleL& t, int e) {
    return c_mul(t.lo[e & t.mask], t.hi[e >> t.shift]);
}

// Grid geometry.  1-D (m1 == 0): m points embedded in one length-L transform
// that the kernels split N1 x N2 with an inter-step twiddle.  2-D (a BTTB of
// Toeplitz blocks on an m1 x m2 grid, reference bttb.py:110-148 with two sizes):
// the embedding is a genuine N1 x N2 two-dimensional transform -- same kernels,
// zero padding / mirroring / cropping per axis, and NO inter-step twiddle
// (TwiddleL.lo == NULL).
struct Geom {
    int m;        // points per output block (m1 * m2 in 2-D)
    int m1, m2;   // 2-D grid sizes; m1 == 0 means 1-D
};

// index into the m-point block of padded position (n1, n2), or -1 for padding.
// mode 1: symmetric circulant column of a top row (mirror per axis).
__device__ __forceinline__ int padded_source(const Geom& g, int n1, int n2, int N1, int N2,
                                             int mode) {
    if (g.m1 == 0) {
        const int n = n1 * N2 + n2;
        if (n < g.m) return n;
        if (mode == 1 && n > N1 * N2 - g.m) return N1 * N2 - n;
        return -1;
    }
    int i1 = -1, i2 = -1;
    if (n1 < g.m1) i1 = n1; else if (mode == 1 && n1 > N1 - g.m1) i1 = N1 - n1;
    if (n2 < g.m2) i2 = n2; else if (mode == 1 && n2 > N2 - g.m2) i2 = N2 - n2;
    return (i1 < 0 || i2 < 0) ? -1 : i1 * g.m2 + i2;
}

__device__ __forceinline__ void load_table(cplx* dst, const cplx* src, int n, int tid, int nthr) {
    for (int i = tid; i < n; i += nthr) dst[i] = src[i];
}

// Optional fused W^T product in front of the column transforms (batched
// solves of small systems, where W^T x as a kernel of its own costs a launch
// and a round trip): the grid vector is never materialised, the column kernels
// gather  g[row] = sum_k WT[row, k] v[col_k]  while it loads.
struct Gather {
    const int* indptr;      // NULL: plain load from X
    const int* indices;
    const double* vals;
    const double* src;      // [nvec][n] data-space vectors
    int n;                  // entries per data-space vector
    int nnz;                // entries of the CSR (for clamping)
    const int* lo;          // non-NULL: the columns of row r are lo[r], lo[r] + 1, ...
                            // (one level of dependent loads less: SkiTerm)
};

// one grid value for the two vectors of a pair (unconditional loads from
// clamped indices, masked afterwards; the first four entries of the row
// together, longer rows finish in a loop)
__device__ __forceinline__ void gather_row(const Gather& gs, int row, const double* d0,
                                           const double* d1, double* re, double* im) {
    constexpr int NZ = 4;
    const int last = gs.nnz > 0 ? gs.nnz - 1 : 0;
    const int kb = gs.indptr[row], ke = gs.indptr[row + 1];
    const int l0 = gs.lo != nullptr ? gs.lo[row] : 0;
    double wa[NZ], g0[NZ], g1[NZ];
#pragma unroll
    for (int e = 0; e < NZ; ++e) {
        const int k = kb + e < last ? kb + e : last;
        const double a = gs.vals[k];
        int col = gs.lo != nullptr ? l0 + e : gs.indices[k];
        col = col < gs.n ? col : gs.n - 1;
        g0[e] = d0[col];
        g1[e] = d1[col];
        wa[e] = kb + e < ke ? a : 0.0;
    }
    double r = 0.0, i = 0.0;
#pragma unroll
    for (int e = 0; e < NZ; ++e) {
        r = fma(wa[e], g0[e], r);
        i = fma(wa[e], g1[e], i);
    }
    for (int k = kb + NZ; k < ke; ++k) {
        const double a = gs.vals[k];
        const int col = gs.indices[k];
        r = fma(a, d0[col], r);
        i = fma(a, d1[col], i);
    }
    *re = r;
    *im = i;
}

// ---------------------------------------------------------------------------
// k_cols_fwd: pad + pack two real vectors -> N1-point column FFTs -> twiddle.
//   grid (N2 / C, D, npairs)   block RL_THREADS
//   X     [nvec][D][m] real (mode 0)  or  tops [ntop][m] with D == 1 (mode 1:
//         symmetric circulant column  c[n] = t[n] (n < m), t[L-n] (n > L-m))
//   T     [npairs][D][N1][N2] complex, position r of a column holds frequency
//         k1 = freq1[r]
// LDS: tile [N1][C] + twiddle table [N1]
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(RL_THREADS)
k_cols_fwd(const double* __restrict__ X, int nvec, int D, Geom geo, int mode, cplx* __restrict__ T,
           int N1, int N2, int C, FftPlan plan1, const cplx* __restrict__ tw1,
           const int* __restrict__ freq1, TwiddleL twl, Gather gs, int* __restrict__ bump) {
    // (the solver's round counter, see rl_solver.h: advanced by the first kernel
    // of a round's operator product)
    if (bump != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 &&
        threadIdx.x == 0)
        *bump += 1;
    RL_SMEM(smem);
    cplx* tile = reinterpret_cast<cplx*>(smem);
    cplx* tw = tile + (size_t)N1 * C;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int c0 = blockIdx.x * C, b = blockIdx.y, pair = blockIdx.z;
    const int L = N1 * N2;
    const int m = geo.m;
    const int v0 = 2 * pair, v1 = 2 * pair + 1;
    const double* x0 = X + ((size_t)v0 * D + b) * m;
    const double* x1 = X + ((size_t)v1 * D + b) * m;
    const bool has1 = v1 < nvec;

    load_table(tw, tw1, N1, tid, nthr);
    if (gs.indptr == nullptr) {
        for (int idx = tid; idx < N1 * C; idx += nthr) {
            const int c = idx % C, n1 = idx / C;
            double re = 0.0, im = 0.0;
            const int src = padded_source(geo, n1, c0 + c, N1, N2, mode);
            if (src >= 0) {
                re = x0[src];
                if (has1) im = x1[src];
            }
            tile[idx] = c_make(re, im);
        }
    } else {
        // fused W^T (see Gather)
        const double* d0 = gs.src + (size_t)v0 * gs.n;
        const double* d1 = has1 ? gs.src + (size_t)v1 * gs.n : d0;
        for (int idx = tid; idx < N1 * C; idx += nthr) {
            const int c = idx % C, n1 = idx / C;
            double re = 0.0, im = 0.0;
            const int src = padded_source(geo, n1, c0 + c, N1, N2, 0);
            if (src >= 0) gather_row(gs, b * m + src, d0, d1, &re, &im);
            tile[idx] = c_make(re, has1 ? im : 0.0);
        }
    }
    __syncthreads();
    fft_tile_forward(tile, plan1, C, C, tw, tid, nthr);

    cplx* out = T + ((size_t)pair * D + b) * L;
    for (int idx = tid; idx < N1 * C; idx += nthr) {
        const int c = idx % C, r = idx / C;
        const int n2 = c0 + c;
        cplx z = tile[idx];
        if (twl.lo != nullptr) z = c_mul(z, twiddle_L(twl, freq1[r] * n2));
        out[(size_t)r * N2 + n2] = z;
    }
}

// ---------------------------------------------------------------------------
// k_rows_spec: N2-point row FFTs of the packed circulant columns; real part ->
// spectrum 2*pair, imaginary part -> spectrum 2*pair+1, both scaled by 1/L.
//   grid (N1 / R, npairs)   T [npairs][1][N1][N2]   spec [ntop][N1*N2]
// LDS: tile [N2][CB] (CB = R | 1) + twiddle table [N2]
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(RL_THREADS)
k_rows_spec(const cplx* __restrict__ T, double* __restrict__ spec, int ntop, int N1, int N2,
            int R, FftPlan plan2, const cplx* __restrict__ tw2) {
    RL_SMEM(smem);
    const int CB = R | 1;
    cplx* tile = reinterpret_cast<cplx*>(smem);
    cplx* tw = tile + (size_t)N2 * CB;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int r0 = blockIdx.x * R, pair = blockIdx.y;
    const size_t L = (size_t)N1 * N2;
    const cplx* in = T + (size_t)pair * L;

    load_table(tw, tw2, N2, tid, nthr);
    for (int idx = tid; idx < R * N2; idx += nthr) {
        const int n2 = idx % N2, rr = idx / N2;
        tile[(size_t)n2 * CB + rr] = in[(size_t)(r0 + rr) * N2 + n2];
    }
    __syncthreads();
    fft_tile_forward(tile, plan2, R, CB, tw, tid, nthr);
    const double scale = 1.0 / (double)L;
    const int q0 = 2 * pair, q1 = 2 * pair + 1;
    for (int idx = tid; idx < R * N2; idx += nthr) {
        const int pos = idx % N2, rr = idx / N2;
        const cplx z = tile[(size_t)pos * CB + rr];
        const size_t o = (size_t)(r0 + rr) * N2 + pos;
        spec[(size_t)q0 * L + o] = z.x * scale;
        if (q1 < ntop) spec[(size_t)q1 * L + o] = z.y * scale;
    }
}

// ---------------------------------------------------------------------------
// Frequency-domain coregionalisation mix, factored form
//   B_q = sum_{f : facQ[f] == q} facW[f] a_f a_f^T + diag(kappa_q)
//   Yhat_a = (sum_q kappa_q[a] s_q) Z_a + sum_f a_f[a] (facW[f] s_{q(f)}) (a_f . Z)
// with s_q the (real, 1/L-scaled) circulant spectrum at this frequency.
// ---------------------------------------------------------------------------
struct MixParams {
    int Q;
    int nfac;
    const double* spec;   // [Q][L]
    const double* facA;   // [nfac][D]
    const double* facW;   // [nfac]
    const int* facQ;      // [nfac]
    const double* kappa;  // [Q][D]
    // optional tables of the third-generation row kernel (rl_kernels3.h):
    // dc [D][L], gs [nfac][L]; NULL = mix from spec / kappa / facW / facQ
    const double* dc;
    const double* gs;
};

// ---------------------------------------------------------------------------
// k_rows_mix<D>: for R rows and all D outputs: N2-point row FFTs, mix, adjoint
// row FFTs, conjugate inter-step twiddle; in place on T.
//   grid (N1 / R, npairs)
// LDS: tile [N2][CB] (CB = (R*D) | 1) + twiddle table [N2]
// ---------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(RL_THREADS)
k_rows_mix(cplx* __restrict__ T, int N1, int N2, int R, FftPlan plan2,
           const cplx* __restrict__ tw2, const int* __restrict__ freq1, TwiddleL twl,
           MixParams mp) {
    RL_SMEM(smem);
    const int cols = R * D;
    const int CB = cols | 1;
    cplx* tile = reinterpret_cast<cplx*>(smem);
    cplx* tw = tile + (size_t)N2 * CB;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int r0 = blockIdx.x * R, pair = blockIdx.y;
    const size_t L = (size_t)N1 * N2;
    cplx* base = T + (size_t)pair * D * L;

    load_table(tw, tw2, N2, tid, nthr);
    for (int idx = tid; idx < cols * N2; idx += nthr) {
        const int n2 = idx % N2, rb = idx / N2;
        const int rr = rb / D, b = rb % D;
        tile[(size_t)n2 * CB + rb] = base[(size_t)b * L + (size_t)(r0 + rr) * N2 + n2];
    }
    __syncthreads();
    fft_tile_forward(tile, plan2, cols, CB, tw, tid, nthr);

    for (int idx = tid; idx < R * N2; idx += nthr) {
        const int pos = idx % N2, rr = idx / N2;
        cplx* zp = tile + (size_t)pos * CB + rr * D;
        const size_t o = (size_t)(r0 + rr) * N2 + pos;
        cplx z[D], y[D];
#pragma unroll
        for (int b = 0; b < D; ++b) z[b] = zp[b];
        double dc[D];
#pragma unroll
        for (int a = 0; a < D; ++a) dc[a] = 0.0;
        for (int q = 0; q < mp.Q; ++q) {
            const double s = mp.spec[(size_t)q * L + o];
#pragma unroll
            for (int a = 0; a < D; ++a) dc[a] = fma(mp.kappa[q * D + a], s, dc[a]);
        }
#pragma unroll
        for (int a = 0; a < D; ++a) y[a] = c_scale(z[a], dc[a]);
        for (int f = 0; f < mp.nfac; ++f) {
            const double* af = mp.facA + (size_t)f * D;
            double sx = 0.0, sy = 0.0;
#pragma unroll
            for (int b = 0; b < D; ++b) {
                sx = fma(af[b], z[b].x, sx);
                sy = fma(af[b], z[b].y, sy);
            }
            const double g = mp.facW[f] * mp.spec[(size_t)mp.facQ[f] * L + o];
            sx *= g;
            sy *= g;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                y[a].x = fma(af[a], sx, y[a].x);
                y[a].y = fma(af[a], sy, y[a].y);
            }
        }
#pragma unroll
        for (int a = 0; a < D; ++a) zp[a] = y[a];
    }
    __syncthreads();
    fft_tile_adjoint(tile, plan2, cols, CB, tw, tid, nthr);

    for (int idx = tid; idx < cols * N2; idx += nthr) {
        const int n2 = idx % N2, rb = idx / N2;
        const int rr = rb / D, b = rb % D;
        cplx z = tile[(size_t)n2 * CB + rb];
        if (twl.lo != nullptr) z = c_mulc(z, twiddle_L(twl, freq1[r0 + rr] * n2));
        base[(size_t)b * L + (size_t)(r0 + rr) * N2 + n2] = z;
    }
}

// ---------------------------------------------------------------------------
// k_cols_inv: adjoint N1-point column FFTs, crop to m, unpack the pair.
//   grid (ceil(ncols_needed / C), D, npairs)
//   Y [nvec][D][m];  beta == 0: Y = result, else Y += result
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(RL_THREADS)
k_cols_inv(const cplx* __restrict__ T, double* __restrict__ Y, int nvec, int D, Geom geo, int N1,
           int N2, int C, FftPlan plan1, const cplx* __restrict__ tw1) {
    RL_SMEM(smem);
    cplx* tile = reinterpret_cast<cplx*>(smem);
    cplx* tw = tile + (size_t)N1 * C;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int c0 = blockIdx.x * C, b = blockIdx.y, pair = blockIdx.z;
    const size_t L = (size_t)N1 * N2;
    const cplx* in = T + ((size_t)pair * D + b) * L;

    load_table(tw, tw1, N1, tid, nthr);
    for (int idx = tid; idx < N1 * C; idx += nthr) {
        const int c = idx % C, r = idx / C;
        tile[idx] = in[(size_t)r * N2 + c0 + c];
    }
    __syncthreads();
    fft_tile_adjoint(tile, plan1, C, C, tw, tid, nthr);

    const int m = geo.m;
    const int v0 = 2 * pair, v1 = 2 * pair + 1;
    double* y0 = Y + ((size_t)v0 * D + b) * m;
    double* y1 = Y + ((size_t)v1 * D + b) * m;
    const bool has1 = v1 < nvec;
    for (int idx = tid; idx < N1 * C; idx += nthr) {
        const int c = idx % C, n1 = idx / C;
        const int dst = padded_source(geo, n1, c0 + c, N1, N2, 0);
        if (dst >= 0) {
            const cplx z = tile[idx];
            y0[dst] = z.x;
            if (has1) y1[dst] = z.y;
        }
    }
}

// ---------------------------------------------------------------------------
// CSR gather SpMV over a batch of vectors:  Y[v] = A X[v]  (+ diag * X2[v])
//   grid (ceil(nrows / RL_THREADS), ceil(nvec / VB))
// Each thread owns one row and VB vectors: the row's CSR entries (12 bytes per
// non-zero) are fetched once and reused for every vector, so the matrix
// structure is streamed nvec / VB times instead of nvec times.
// Used for W^T x (rows = D*m grid points) and for W g + eps * x (rows = n);
// `accumulate` adds into Y (further terms of a split-kernel operator).
// ---------------------------------------------------------------------------
template <int RL_SPMV_VB>
__global__ void __launch_bounds__(RL_THREADS)
k_spmv(const int* __restrict__ indptr, const int* __restrict__ indices,
       const double* __restrict__ vals, int nrows, int ncols, int nvec,
       const double* __restrict__ X, double* __restrict__ Y, const double* __restrict__ diag,
       const double* __restrict__ X2, int accumulate, int* __restrict__ bump) {
    // the solver's round counter: advanced by the first kernel of a round's
    // operator product (nothing in this kernel reads it)
    if (bump != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *bump += 1;
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int v0 = blockIdx.y * RL_SPMV_VB;
    if (row >= nrows) return;
    const int nv = nvec - v0 < RL_SPMV_VB ? nvec - v0 : RL_SPMV_VB;
    double acc[RL_SPMV_VB];
#pragma unroll
    for (int j = 0; j < RL_SPMV_VB; ++j) acc[j] = 0.0;
    const double* x = X + (size_t)v0 * ncols;
    const int kb = indptr[row], k1 = indptr[row + 1];
    // the first NZ entries of the row are requested together (interpolation
    // rows have 4): entries -> gathered values is a dependent chain per entry,
    // and small batches are bound by exactly that latency
    constexpr int NZ = 4;
    // (unconditional loads from clamped indices, masked afterwards: a
    // conditional load is a branch with a full memory wait behind it)
    const int nnz = indptr[nrows];
    const int last = nnz > 0 ? nnz - 1 : 0;
    // (rows longer than NZ -- W^T where several data points share a grid cell --
    // go on in further groups of NZ, each requested together like the first)
    int kg = kb;
    do {
        double wa[NZ];
        int wc[NZ];
#pragma unroll
        for (int e = 0; e < NZ; ++e) {
            const int k = kg + e < last ? kg + e : last;
            const double a = vals[k];
            wc[e] = indices[k];
            wa[e] = kg + e < k1 ? a : 0.0;
        }
#pragma unroll
        for (int j = 0; j < RL_SPMV_VB; ++j) {
            const int jj = j < nv ? j : 0;
            double xv[NZ];
#pragma unroll
            for (int e = 0; e < NZ; ++e) xv[e] = x[(size_t)jj * ncols + wc[e]];
#pragma unroll
            for (int e = 0; e < NZ; ++e) acc[j] = fma(wa[e], xv[e], acc[j]);
        }
        kg += NZ;
    } while (kg < k1);
    const double dg = diag != nullptr ? diag[row] : 0.0;
#pragma unroll
    for (int j = 0; j < RL_SPMV_VB; ++j)
        if (j < nv) {
            double r = acc[j];
            if (diag != nullptr) r = fma(dg, X2[(size_t)(v0 + j) * nrows + row], r);
            if (accumulate) r += Y[(size_t)(v0 + j) * nrows + row];
            Y[(size_t)(v0 + j) * nrows + row] = r;
        }
}

// ---------------------------------------------------------------------------
// k_spmv_wide: the same product for LONG rows (W^T when there are many more
// data points than grid points: 30 entries per row on the weather workload,
// which one thread per row walks as a serial chain of gathers).  LPR lanes
// share a row, each takes every LPR-th entry (4 of them requested together),
// the partial sums meet in LDS.
//   grid (ceil(nrows * LPR / RL_THREADS), nvec)   LDS: RL_THREADS doubles
// ---------------------------------------------------------------------------
template <int LPR>
__global__ void __launch_bounds__(RL_THREADS)
k_spmv_wide(const int* __restrict__ indptr, const int* __restrict__ indices,
            const double* __restrict__ vals, int nrows, int ncols, const double* __restrict__ X,
            double* __restrict__ Y, int* __restrict__ bump) {
    if (bump != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *bump += 1;
    RL_SMEM(smem);
    double* part = reinterpret_cast<double*>(smem);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int rowu = t / LPR, lane = t - rowu * LPR;
    const int row = rowu < nrows ? rowu : nrows - 1;
    const double* x = X + (size_t)blockIdx.y * ncols;
    const int nnz = indptr[nrows];
    const int last = nnz > 0 ? nnz - 1 : 0;
    const int kb = indptr[row], k1 = indptr[row + 1];
    double acc = 0.0;
    constexpr int NZ = 4;
    for (int k0 = kb + lane; k0 < k1; k0 += NZ * LPR) {
        double wa[NZ], xv[NZ];
#pragma unroll
        for (int e = 0; e < NZ; ++e) {
            const int ku = k0 + e * LPR;
            const int k = ku < last ? ku : last;
            const double a = vals[k];
            xv[e] = x[indices[k]];
            wa[e] = ku < k1 ? a : 0.0;
        }
#pragma unroll
        for (int e = 0; e < NZ; ++e) acc = fma(wa[e], xv[e], acc);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (lane == 0 && rowu < nrows) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < LPR; ++j) s += part[threadIdx.x + j];
        Y[(size_t)blockIdx.y * nrows + rowu] = s;
    }
}

// ---------------------------------------------------------------------------
// k_spmv_wt_staged<VB, XPT>: W^T x for LARGE batches when W^T has the consecutive-
// range structure (SkiTerm::WT_lo: the entries of grid row r multiply the data
// rows lo[r], lo[r] + 1, ...; lo is non-decreasing).  The data range of a
// workgroup's RL_THREADS grid rows is then one contiguous piece of every
// vector: it is staged into LDS with coalesced loads together with the
// workgroup's weights, and the rows -- whose lengths vary (as many entries as
// data points near the grid point) -- are summed out of LDS.  One phase of
// global loads per wavefront instead of one per group of four entries of its
// longest row (k_spmv), same summation order, same results.
// A workgroup walks `vgroups` groups of VB vectors with the SAME rows: weights,
// row pointers and ranges are read once for all of them, and the pieces of the
// next group travel from memory into registers (XPT values per thread and
// vector, xcap <= XPT * RL_THREADS) while the current group is summed out of
// LDS and stored -- loads, sums and stores of a workgroup overlap instead of
// taking turns (measured at C5, 129 vectors: DESIGN.md).
// (Measured and dropped, round 3: the pieces of a group as one flat list of 16-byte
// pairs, two groups ahead in flight -- fewer, wider loads, none of them past the piece.
// 536 against 467 us per C5 product with pairs at the piece's own parity, 745 with pairs
// aligned to 16 bytes and nine slots a thread: the kernel's registers went from ~100 to
// 165 and its occupancy with them.)
//   grid (ceil(nrows / RL_THREADS), ceil(ceil(nvec / VB) / vgroups))
//   LDS: VB * xcap doubles (vector pieces) + ecap doubles (weights); the host
//   guarantees every workgroup's range <= xcap and entries <= ecap
// ---------------------------------------------------------------------------
template <int VB, int XPT>
__global__ void __launch_bounds__(RL_THREADS)
k_spmv_wt_staged(const int* __restrict__ indptr, const int* __restrict__ lo,
                 const double* __restrict__ vals, int nrows, int ncols, int nvec,
                 const double* __restrict__ X, double* __restrict__ Y, int xcap, int vgroups,
                 int* __restrict__ bump) {
    if (bump != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *bump += 1;
    RL_SMEM(smem);
    double* xs = reinterpret_cast<double*>(smem);          // [VB][xcap]
    double* vs = xs + (size_t)VB * xcap;                   // [ecap]
    const int tid = threadIdx.x, nthr = blockDim.x;
    // (workgroups numbered with the vector-group column fastest: 644 / 472 us against
    // 675 / 490 for W / W^T at C5 -- neighbours do not all write the same vectors)
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int bx = lin / gridDim.y, by = lin - bx * gridDim.y;
    const int r0 = bx * nthr;
    const int rl = (r0 + nthr < nrows ? r0 + nthr : nrows) - 1;      // last row of the workgroup
    const int groups = (nvec + VB - 1) / VB;
    const int g0 = by * vgroups;
    const int ng = groups - g0 < vgroups ? groups - g0 : vgroups;
    const int k0 = indptr[r0], k1 = indptr[rl + 1];
    const int c0 = lo[r0], c1 = lo[rl] + (k1 - indptr[rl]);
    const int len = c1 - c0;
    const int row = r0 + tid;
    const int rowc = row <= rl ? row : rl;
    const int kb = indptr[rowc];
    const int cnt = row <= rl ? indptr[rowc + 1] - kb : 0;
    const int l = lo[rowc];
    for (int i = tid; i < k1 - k0; i += nthr) vs[i] = vals[k0 + i];
    // (unconditional loads from clamped positions: a range may be empty)
    double xr[VB][XPT];
    auto request = [&](int grp) {
        const int v0 = (g0 + grp) * VB;
        const int nv = nvec - v0 < VB ? nvec - v0 : VB;
#pragma unroll
        for (int j = 0; j < VB; ++j) {
            const double* x = X + (size_t)(v0 + (j < nv ? j : 0)) * ncols;
#pragma unroll
            for (int u = 0; u < XPT; ++u) {
                // (positions past the range -- up to xcap - len of them -- repeat its last
                // element: a hit in the same line instead of a fetch of the neighbour's range;
                // the PMC counters showed the staged kernels fetching 1.2-2.3x their operand)
                int c = c0 + tid + u * nthr;
                c = c < c1 ? c : c1 - 1;
                c = c < ncols ? c : ncols - 1;
                c = c > 0 ? c : 0;
                xr[j][u] = x[c];
            }
        }
    };
    request(0);
    const double* a = vs + (kb - k0);
    const double* xrow = xs + (l - c0);
    for (int grp = 0; grp < ng; ++grp) {
        // (round 4, measured and dropped: skipping the LDS traffic and sums of a ragged last
        // group's empty slots -- 17 = 2 x 8 + 1 vectors -- behind a uniform branch: 96 vs 88 us
        // at 17 vectors, 519 vs 485 at 129; the same guard pays in k_spmv_w_poly, whose slots
        // cost 24 multiply-adds each: 78 vs 96 us at 17 vectors)
#pragma unroll
        for (int j = 0; j < VB; ++j)
#pragma unroll
            for (int u = 0; u < XPT; ++u) {
                const int i = tid + u * nthr;
                if (i < len) xs[(size_t)j * xcap + i] = xr[j][u];
            }
        __syncthreads();
        if (grp + 1 < ng) request(grp + 1);
        double acc[VB];
#pragma unroll
        for (int j = 0; j < VB; ++j) acc[j] = 0.0;
        for (int e = 0; e < cnt; ++e) {
            const double w = a[e];
#pragma unroll
            for (int j = 0; j < VB; ++j) acc[j] = fma(w, xrow[(size_t)j * xcap + e], acc[j]);
        }
        const int v0 = (g0 + grp) * VB;
        if (row <= rl) {
#pragma unroll
            for (int j = 0; j < VB; ++j)
                if (v0 + j < nvec) Y[(size_t)(v0 + j) * nrows + row] = acc[j];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// k_spmv_w_staged<VB, XPT>: Y = W g (+ diag (.) X2) for LARGE batches when W is
// held as base column + four weights per row with non-decreasing bases
// (SkiTerm::W4_base / W4_w): the grid range of a workgroup's RL_THREADS data
// rows is one contiguous piece of every grid vector, staged into LDS with
// coalesced loads; base and weights of a row are independent loads.  Two
// dependent levels of global loads (bases of the first and last row -> range)
// instead of three (row pointers -> entries -> gathered values), no index
// array, same summation order as k_spmv.  A workgroup walks `vgroups` groups of
// VB vectors with the same rows (base, weights and diag stay in registers) and
// requests the next group's pieces while it sums the current one, like
// k_spmv_wt_staged.
//   grid (ceil(nrows / RL_THREADS), ceil(ceil(nvec / VB) / vgroups))
//   LDS: VB * xcap doubles, xcap <= XPT * RL_THREADS
// ---------------------------------------------------------------------------
template <int VB, int XPT>
__global__ void __launch_bounds__(RL_THREADS)
k_spmv_w_staged(const int* __restrict__ base, const double* __restrict__ w4, int nrows,
                int ncols, int nvec, const double* __restrict__ G, double* __restrict__ Y,
                const double* __restrict__ diag, const double* __restrict__ X2, int xcap,
                int vgroups) {
    RL_SMEM(smem);
    double* xs = reinterpret_cast<double*>(smem);          // [VB][xcap]
    const int tid = threadIdx.x, nthr = blockDim.x;
    // (workgroups numbered with the vector-group column fastest: 644 / 472 us against
    // 675 / 490 for W / W^T at C5 -- neighbours do not all write the same vectors)
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
    double xr[VB][XPT], x2[VB];
    auto request = [&](int grp) {
        const int v0 = (g0 + grp) * VB;
        const int nv = nvec - v0 < VB ? nvec - v0 : VB;
#pragma unroll
        for (int j = 0; j < VB; ++j) {
            const size_t v = (size_t)(v0 + (j < nv ? j : 0));
            const double* g = G + v * ncols;
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
            x2[j] = diag != nullptr ? X2[v * nrows + rowc] : 0.0;
        }
    };
    request(0);
    const double* xrow = xs + (b - c0);
    for (int grp = 0; grp < ng; ++grp) {
        double d2[VB];
#pragma unroll
        for (int j = 0; j < VB; ++j) {
            d2[j] = x2[j];
#pragma unroll
            for (int u = 0; u < XPT; ++u) {
                const int i = tid + u * nthr;
                if (i < len) xs[(size_t)j * xcap + i] = xr[j][u];
            }
        }
        __syncthreads();
        if (grp + 1 < ng) request(grp + 1);
        const int v0 = (g0 + grp) * VB;
        if (row <= rl) {
#pragma unroll
            for (int j = 0; j < VB; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = fma(w[e], xrow[(size_t)j * xcap + e], acc);
                if (diag != nullptr) acc = fma(dg, d2[j], acc);
                if (v0 + j < nvec) Y[(size_t)(v0 + j) * nrows + row] = acc;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Row permutation of a batch of data-space vectors.
//   gather : Y[v][i]       = X[v][perm[i]]
//   scatter: Y[v][perm[i]] = X[v][i]
// The SKI handle keeps its data points sorted by grid position internally
// (interpolation rows of unsorted inputs gather from random grid positions:
// one cache line per 8 useful bytes); callers keep their own order.
//   grid (ceil(n / RL_THREADS), nvec)
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(RL_THREADS)
k_permute_rows(const double* __restrict__ X, double* __restrict__ Y,
               const int* __restrict__ perm, int n, int scatter) {
    // Sorting keeps every output's points together, so the random side of the
    // copy stays inside one output's slice of the vector (0.8 MB at C5): an
    // L2-sized window -- if the workgroups that share it sit on ONE XCD.  Blocks
    // go to the eight XCDs round robin, so XCD k takes the contiguous eighth
    // [k nb/8, (k+1) nb/8) of the row blocks; with the plain order every XCD
    // pulled every window through its own L2 (8x the reads: 1.2 ms per 129
    // C5 vectors instead of 0.4).  Placement only affects speed.
    const int nb = gridDim.x, b = blockIdx.x;
    const int lb = (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b;
    const int i = lb * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t off = (size_t)blockIdx.y * n;
    if (scatter)
        Y[off + perm[i]] = X[off + i];
    else
        Y[off + i] = X[off + perm[i]];
}
