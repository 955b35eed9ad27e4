// Second-generation grid-product kernels: the FIRST radix pass of every
// transform is done in registers straight from the global load and the LAST
// one in registers straight into the global store (or the mix), so a two-pass
// transform (N <= 256) crosses LDS once and a three-pass one twice, instead of
// once per pass plus once each for staging in and out.  Every thread issues its
// RA (8 or 16) independent global loads before it touches any of them.
//
// Same flow graph, same scrambled orders, same intermediates layout as
// rl_kernels.h (which stays as the fallback for transform lengths without a
// fused instantiation); see tests/flow_model.py.
//
// Plan convention: plan.radix[0] == RA (first pass), plan.radix[npass-1] == RB
// (last pass), whatever sits between runs in LDS (fft_pass_any).
#pragma once
#include "rl_kernels.h"

// upper bound of the workgroup size of the k2_* kernels; the launcher picks 256
// or 512 threads by how many butterflies the first pass of a tile has
#define RL_THREADS2 512

struct Tile2 {
    int N1, N2;
    int C, logC;        // columns per k2_cols_* workgroup (power of two)
    int R;              // rows per k2_rows_mix workgroup (power of two)
    unsigned colsMagic; // fast_div magic of R * D
    int thrC, thrR;     // workgroup sizes of the column / row kernels
    int aff;            // > 0: pair-affine order over this many pairs (affine_tile below)
};

// Pair-affine order (round 4).  The three kernels of a product hand a pair's
// intermediates T[pair] from one to the next through memory; each XCD has its own
// 4 MB L2, and workgroups go to the XCDs round robin by linear block id.  With the
// orders above, the tiles that write T[pair] in one kernel and those that read it in the
// next sit on different XCDs, so every read of T comes from HBM / Infinity Cache.  In
// this order XCD k owns the pairs k, k + 8, ... in ALL three kernels (a 1-D launch of
// 8 * ceil(pairs / 8) * tiles workgroups; those of a pair past the end return at
// once), and the host keeps a chunk's intermediates per XCD inside its L2: a kernel's
// reads of T are then served by the L2 the previous kernel wrote through.  (Placement
// is the dispatcher's observed behaviour: only speed depends on it.)
__device__ __forceinline__ bool affine_tile(const Tile2& tp, int tiles, int* tile, int* pair) {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int pl = slot / tiles;
    *tile = slot - pl * tiles;
    *pair = pl * 8 + xcd;
    return *pair < tp.aff;
}

// Column-tile index of this workgroup.  Workgroups go to the eight XCDs round
// robin by linear block id, so with the plain order the column tiles that share
// a 128-byte line of x / y (a tile of C = 8 real columns is 64 bytes wide) land on
// different XCDs and the line is fetched once per XCD.  When the tile count is a
// multiple of 8, XCD k takes the contiguous tiles [k n/8, (k+1) n/8): neighbours
// hit in that XCD's L2.  (Placement is the dispatcher's observed behaviour; only
// speed depends on it.)
__device__ __forceinline__ int xcd_column_tile() {
    const int n = gridDim.x, b = blockIdx.x;
    return (n & 7) == 0 && n >= 16 ? (b & 7) * (n >> 3) + (b >> 3) : b;
}

// middle passes (everything but first and last), forward / adjoint
__device__ __forceinline__ void middle_forward(cplx* tile, const FftPlan& plan, int cols, int ld,
                                               const cplx* tw, int tid, int nthr,
                                               unsigned magic) {
    int ns = plan.n / plan.radix[0];
    for (int s = 1; s < plan.npass - 1; ++s) {
        fft_pass_any<false>(plan.radix[s], tile, plan.n, ns, cols, ld, tw, tid, nthr, magic);
        ns /= plan.radix[s];
        __syncthreads();
    }
}
__device__ __forceinline__ void middle_adjoint(cplx* tile, const FftPlan& plan, int cols, int ld,
                                               const cplx* tw, int tid, int nthr,
                                               unsigned magic) {
    int ns = plan.radix[plan.npass - 1];
    for (int s = plan.npass - 2; s >= 1; --s) {
        ns *= plan.radix[s];
        fft_pass_any<true>(plan.radix[s], tile, plan.n, ns, cols, ld, tw, tid, nthr, magic);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// k2_cols_fwd<RA, RB, GATHER>: as k_cols_fwd (GATHER: with the W^T product
// fused into the load, see Gather).  grid (N2 / C, D, npairs)
// LDS: tile [N1][C]
// ---------------------------------------------------------------------------
template <int RA, int RB, bool GATHER>
__global__ void __launch_bounds__(RL_THREADS2)
k2_cols_fwd(const double* __restrict__ X, int nvec, int D, Geom geo, int mode,
            cplx* __restrict__ T, Tile2 tp, FftPlan plan1, const cplx* __restrict__ tw1,
            const int* __restrict__ freq1, TwiddleL twl, Gather gs) {
    RL_STAMP(10);
    RL_SMEM(smem);
    cplx* tile = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int N1 = tp.N1, N2 = tp.N2, C = tp.C;
    int ct, b, pair;
    if (tp.aff > 0) {
        int t;
        if (!affine_tile(tp, (N2 / C) * D, &t, &pair)) return;
        ct = t % (N2 / C);
        b = t / (N2 / C);
    } else {
        ct = xcd_column_tile(), b = blockIdx.y, pair = blockIdx.z;
    }
    const int c0 = ct * C;
    const int L = N1 * N2;
    const int m = geo.m;
    const int v0 = 2 * pair, v1 = 2 * pair + 1;
    const double* x0 = X + ((size_t)v0 * D + b) * m;
    const double* x1 = X + ((size_t)v1 * D + b) * m;
    const bool has1 = v1 < nvec;
    const int sub = N1 / RA;

    // fused W^T: the two data-space vectors of this pair
    const double* d0 = nullptr;
    const double* d1 = nullptr;
    if (GATHER) {
        d0 = gs.src + (size_t)v0 * gs.n;
        d1 = has1 ? gs.src + (size_t)v1 * gs.n : d0;     // loads stay unconditional
    }
    for (int w = tid; w < sub * C; w += nthr) {
        const int c = w & (C - 1), j = w >> tp.logC;
        cplx v[RA];
        if (!GATHER) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int src = padded_source(geo, j + sub * i, c0 + c, N1, N2, mode);
                double re = 0.0, im = 0.0;
                if (src >= 0) {
                    re = x0[src];
                    if (has1) im = x1[src];
                }
                v[i] = c_make(re, im);
            }
        } else {
            // three dependent levels (row pointers -> entries -> data values),
            // each requested for all RA legs before the next is touched; the
            // first NZ entries of a row are unrolled, longer rows loop
            // Every load below is UNCONDITIONAL (indices clamped to valid
            // entries, results masked afterwards): a conditional load compiles
            // to a branch with a full memory wait behind it, which serialises
            // the very latencies this code is arranged to overlap.
            constexpr int NZ = 4;
            constexpr int CH = RA == 3 ? 3 : (RA == 5 ? 5 : 4);   // legs in flight together
            const int last = gs.nnz > 0 ? gs.nnz - 1 : 0;
#pragma unroll
            for (int i0 = 0; i0 < RA; i0 += CH) {
                int kb[CH], ke[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    // (a leg past RA in the last chunk behaves like padding)
                    const int src = i0 + i < RA
                        ? padded_source(geo, j + sub * (i0 + i), c0 + c, N1, N2, 0) : -1;
                    const int row = src >= 0 ? b * m + src : 0;
                    const int p0 = gs.indptr[row], p1 = gs.indptr[row + 1];
                    kb[i] = src >= 0 ? p0 : 0;
                    ke[i] = src >= 0 ? p1 : 0;
                }
                double wa[CH][NZ];
                int wc[CH][NZ];
                if (gs.lo != nullptr) {
                    // consecutive columns: first column with the row pointers,
                    // weights and data values together in the next level
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const int src = i0 + i < RA
                            ? padded_source(geo, j + sub * (i0 + i), c0 + c, N1, N2, 0) : -1;
                        const int l0 = gs.lo[src >= 0 ? b * m + src : 0];
#pragma unroll
                        for (int e = 0; e < NZ; ++e) {
                            const int col = l0 + e;
                            wc[i][e] = col < gs.n ? col : gs.n - 1;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < CH; ++i)
#pragma unroll
                        for (int e = 0; e < NZ; ++e) {
                            const int k = kb[i] + e < last ? kb[i] + e : last;
                            const double a = gs.vals[k];
                            wa[i][e] = kb[i] + e < ke[i] ? a : 0.0;
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < CH; ++i)
#pragma unroll
                        for (int e = 0; e < NZ; ++e) {
                            const int k = kb[i] + e < last ? kb[i] + e : last;
                            const double a = gs.vals[k];
                            wc[i][e] = gs.indices[k];
                            wa[i][e] = kb[i] + e < ke[i] ? a : 0.0;
                        }
                }
                // data values of ALL legs of the chunk before any is used (the
                // rare rows longer than NZ finish afterwards)
                double g0[CH][NZ], g1[CH][NZ];
#pragma unroll
                for (int i = 0; i < CH; ++i)
#pragma unroll
                    for (int e = 0; e < NZ; ++e) {
                        g0[i][e] = d0[wc[i][e]];
                        g1[i][e] = d1[wc[i][e]];
                    }
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    double re = 0.0, im = 0.0;
#pragma unroll
                    for (int e = 0; e < NZ; ++e) {
                        re = fma(wa[i][e], g0[i][e], re);
                        im = fma(wa[i][e], g1[i][e], im);
                    }
                    if (i0 + i < RA) v[i0 + i] = c_make(re, has1 ? im : 0.0);
                }
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    if (i0 + i < RA)
                        for (int k = kb[i] + NZ; k < ke[i]; ++k) {
                            const double a = gs.vals[k];
                            const int col = gs.indices[k];
                            v[i0 + i].x = fma(a, d0[col], v[i0 + i].x);
                            if (has1) v[i0 + i].y = fma(a, d1[col], v[i0 + i].y);
                        }
            }
        }
        SmallDft<RA, false>::run(v);
#pragma unroll
        for (int k = 1; k < RA; ++k) v[k] = c_mul(v[k], tw1[j * k]);
#pragma unroll
        for (int k = 0; k < RA; ++k) tile[(size_t)(j + sub * k) * C + c] = v[k];
    }
    RL_STAMP(11);
    __syncthreads();
    middle_forward(tile, plan1, C, C, tw1, tid, nthr, 0);
    RL_STAMP(12);

    cplx* out = T + ((size_t)pair * D + b) * L;
    for (int w = tid; w < (N1 / RB) * C; w += nthr) {
        const int c = w & (C - 1), g = (w >> tp.logC) * RB;
        cplx v[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) v[i] = tile[(size_t)(g + i) * C + c];
        SmallDft<RB, false>::run(v);
        const int n2 = c0 + c;
        if (twl.lo != nullptr) {
            // inter-step twiddle W_L^{freq1[g + k] n2}.  The last pass puts
            // frequency f0 + k (N1 / RB) at position g + k, so the RB factors are
            // W_L^{f0 n2} * (W_L^{(N1 / RB) n2})^k: five table look-ups (base,
            // step^1,2,4,8) and at most three multiplications per power instead
            // of RB look-ups of three dependent loads each
            const int M = N1 / RB;
            const cplx wb = twiddle_L(twl, freq1[g] * n2);
            cplx sk[RB];
            sk[1] = twiddle_L(twl, M * n2);
            sk[2] = twiddle_L(twl, 2 * M * n2);
            sk[4] = twiddle_L(twl, 4 * M * n2);
            sk[3] = c_mul(sk[2], sk[1]);
            sk[5] = c_mul(sk[4], sk[1]);
            sk[6] = c_mul(sk[4], sk[2]);
            sk[7] = c_mul(sk[4], sk[3]);
            if (RB == 16) {
                sk[RB / 2] = twiddle_L(twl, (RB / 2) * M * n2);
#pragma unroll
                for (int k = RB / 2 + 1; k < RB; ++k) sk[k] = c_mul(sk[RB / 2], sk[k - RB / 2]);
            }
            v[0] = c_mul(v[0], wb);
#pragma unroll
            for (int k = 1; k < RB; ++k) v[k] = c_mul(v[k], c_mul(wb, sk[k]));
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) out[(size_t)(g + k) * N2 + n2] = v[k];
    }
    RL_STAMP(13);
}

// ---------------------------------------------------------------------------
// k2_cols_inv<RA, RB>: as k_cols_inv.  grid (tiles, D, npairs)
// ---------------------------------------------------------------------------
template <int RA, int RB>
__global__ void __launch_bounds__(RL_THREADS2)
k2_cols_inv(const cplx* __restrict__ T, double* __restrict__ Y, int nvec, int D, Geom geo,
            Tile2 tp, FftPlan plan1, const cplx* __restrict__ tw1) {
    RL_STAMP(30);
    RL_SMEM(smem);
    cplx* tile = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int N1 = tp.N1, N2 = tp.N2, C = tp.C;
    int ct, b, pair;
    if (tp.aff > 0) {
        const int need = geo.m1 ? geo.m2 : (geo.m < N2 ? geo.m : N2);
        const int nct = (need + C - 1) / C;
        int t;
        if (!affine_tile(tp, nct * D, &t, &pair)) return;
        ct = t % nct;
        b = t / nct;
    } else {
        ct = xcd_column_tile(), b = blockIdx.y, pair = blockIdx.z;
    }
    const int c0 = ct * C;
    const size_t L = (size_t)N1 * N2;
    const cplx* in = T + ((size_t)pair * D + b) * L;

    // adjoint of the last forward pass, straight from global
    for (int w = tid; w < (N1 / RB) * C; w += nthr) {
        const int c = w & (C - 1), g = (w >> tp.logC) * RB;
        cplx v[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) v[k] = in[(size_t)(g + k) * N2 + c0 + c];
        SmallDft<RB, true>::run(v);
#pragma unroll
        for (int i = 0; i < RB; ++i) tile[(size_t)(g + i) * C + c] = v[i];
    }
    RL_STAMP(32);
    __syncthreads();
    middle_adjoint(tile, plan1, C, C, tw1, tid, nthr, 0);
    RL_STAMP(33);

    // adjoint of the first forward pass, straight into the cropped output
    const int sub = N1 / RA;
    const int m = geo.m;
    const int v0 = 2 * pair, v1 = 2 * pair + 1;
    double* y0 = Y + ((size_t)v0 * D + b) * m;
    double* y1 = Y + ((size_t)v1 * D + b) * m;
    const bool has1 = v1 < nvec;
    // rows n1 >= ceil((m - c0) / N2) of this tile are cropped away entirely
    for (int w = tid; w < sub * C; w += nthr) {
        const int c = w & (C - 1), j = w >> tp.logC;
        // even the first leg (the smallest row index) is cropped away
        if (padded_source(geo, j, c0 + c, N1, N2, 0) < 0) continue;
        cplx v[RA];
#pragma unroll
        for (int k = 0; k < RA; ++k) v[k] = tile[(size_t)(j + sub * k) * C + c];
#pragma unroll
        for (int k = 1; k < RA; ++k) v[k] = c_mulc(v[k], tw1[j * k]);
        SmallDft<RA, true>::run(v);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int dst = padded_source(geo, j + sub * i, c0 + c, N1, N2, 0);
            if (dst >= 0) {
                y0[dst] = v[i].x;
                if (has1) y1[dst] = v[i].y;
            }
        }
    }
    RL_STAMP(31);
}

// ---------------------------------------------------------------------------
// the per-frequency mix on D complex values held in registers
// ---------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void mix_point(cplx* z, const MixParams& mp, size_t L, size_t o) {
    cplx y[D];
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
    for (int a = 0; a < D; ++a) z[a] = y[a];
}

// ---------------------------------------------------------------------------
// k2_rows_mix<D, RA, RB>: as k_rows_mix.  grid (N1 / R, npairs)
// LDS: tile [N2][ld], ld = (R*D) | 1
// First pass from global with the transform index fastest across lanes
// (coalesced 16-byte elements); LDS passes with the column fastest.
// ---------------------------------------------------------------------------
template <int D, int RA, int RB>
__global__ void __launch_bounds__(RL_THREADS2)
k2_rows_mix(cplx* __restrict__ T, Tile2 tp, FftPlan plan2, const cplx* __restrict__ tw2,
            const int* __restrict__ freq1, TwiddleL twl, MixParams mp, int* __restrict__ bump) {
    // the solver's round counter when W^T is fused into k2_cols_fwd (which reads
    // it): advanced here, by a kernel that does not
    RL_STAMP(20);
    if (bump != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *bump += 1;
    RL_SMEM(smem);
    cplx* tile = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int N1 = tp.N1, N2 = tp.N2, R = tp.R;
    const int cols = R * D;
    const int ld = cols | 1;
    const int rt = blockIdx.x, pair = blockIdx.y;
    const int r0 = rt * R;
    const size_t L = (size_t)N1 * N2;
    cplx* base = T + (size_t)pair * D * L;
    const int sub = N2 / RA;
    const int logsub = 31 - __builtin_clz(sub);

    for (int w = tid; w < sub * cols; w += nthr) {
        const int j = w & (sub - 1), col = w >> logsub;
        const int rr = col / D, b = col - rr * D;
        const cplx* src = base + (size_t)b * L + (size_t)(r0 + rr) * N2;
        cplx v[RA];
#pragma unroll
        for (int i = 0; i < RA; ++i) v[i] = src[j + sub * i];
        SmallDft<RA, false>::run(v);
#pragma unroll
        for (int k = 1; k < RA; ++k) v[k] = c_mul(v[k], tw2[j * k]);
#pragma unroll
        for (int k = 0; k < RA; ++k) tile[(size_t)(j + sub * k) * ld + col] = v[k];
    }
    RL_STAMP(22);
    __syncthreads();
    middle_forward(tile, plan2, cols, ld, tw2, tid, nthr, tp.colsMagic);
    fft_pass<RB, false>(tile, N2, RB, cols, ld, tw2, tid, nthr, tp.colsMagic);
    __syncthreads();
    RL_STAMP(23);

    for (int idx = tid; idx < R * N2; idx += nthr) {
        const int pos = idx & (N2 - 1), rr = idx / N2;
        cplx* zp = tile + (size_t)pos * ld + rr * D;
        cplx z[D];
#pragma unroll
        for (int b = 0; b < D; ++b) z[b] = zp[b];
        mix_point<D>(z, mp, L, (size_t)(r0 + rr) * N2 + pos);
#pragma unroll
        for (int a = 0; a < D; ++a) zp[a] = z[a];
    }
    RL_STAMP(24);
    __syncthreads();
    fft_pass<RB, true>(tile, N2, RB, cols, ld, tw2, tid, nthr, tp.colsMagic);
    __syncthreads();
    middle_adjoint(tile, plan2, cols, ld, tw2, tid, nthr, tp.colsMagic);
    RL_STAMP(25);

    for (int w = tid; w < sub * cols; w += nthr) {
        const int j = w & (sub - 1), col = w >> logsub;
        const int rr = col / D, b = col - rr * D;
        cplx* dst = base + (size_t)b * L + (size_t)(r0 + rr) * N2;
        const int k1 = freq1[r0 + rr];
        cplx v[RA];
#pragma unroll
        for (int k = 0; k < RA; ++k) v[k] = tile[(size_t)(j + sub * k) * ld + col];
#pragma unroll
        for (int k = 1; k < RA; ++k) v[k] = c_mulc(v[k], tw2[j * k]);
        SmallDft<RA, true>::run(v);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int n2 = j + sub * i;
            cplx z = v[i];
            if (twl.lo != nullptr) z = c_mulc(z, twiddle_L(twl, k1 * n2));
            dst[n2] = z;
        }
    }
    RL_STAMP(21);
}

// ---------------------------------------------------------------------------
// k1_product<D>: the whole product of one PAIR of vectors in one workgroup,
// for grids short enough that all D transforms of the pair fit one LDS tile
// (L * (D | 1) * 16 bytes: the reference's README, FX2007 and weather
// workloads).  One kernel instead of three, no intermediates in global memory,
// and -- unlike k4_product -- pair packing is kept, so nothing is untangled:
//   pad + pack the pair ->
//   single-level in-place transform of all D outputs side by side ->
//   real D x D mix at every position -> adjoint transform -> crop, unpack.
// Spectra for this kernel: [Q][L] in the scrambled order of the single-level
// transform, made by the same kernel (mode 1: X = tops, two per transform).
//   grid (npairs)   block up to RL_THREADS2
// LDS: tile [L][ld], ld = D | 1, then the twiddle table [L]
// ---------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(RL_THREADS2)
k1_product(const double* __restrict__ X, double* __restrict__ Y, int nvec, Geom geo, int mode,
           FftPlan plan, const cplx* __restrict__ twL, MixParams mp,
           double* __restrict__ spec_out) {
    RL_SMEM(smem);
    cplx* tile = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int L = plan.n, m = geo.m;
    constexpr int ld = D | 1;
    cplx* tw = tile + (size_t)L * ld;
    const int pair = blockIdx.x;
    const int v0 = 2 * pair, v1 = 2 * pair + 1;
    const bool has1 = v1 < nvec;

    load_table(tw, twL, L, tid, nthr);
    if (mode == 1) {
        // spectra: circulant columns of tops v0, v1 (one output column)
        const double* t0 = X + (size_t)v0 * m;
        const double* t1 = X + (size_t)v1 * m;
        for (int n = tid; n < L; n += nthr) {
            const int src = padded_source(geo, 0, n, 1, L, 1);
            double re = 0.0, im = 0.0;
            if (src >= 0) {
                re = t0[src];
                if (has1) im = t1[src];
            }
            tile[(size_t)n * ld] = c_make(re, im);
        }
    } else {
        for (int w = tid; w < D * L; w += nthr) {
            const int b = w / L, n = w - b * L;         // n fastest: coalesced reads
            double re = 0.0, im = 0.0;
            if (n < m) {
                re = X[((size_t)v0 * D + b) * m + n];
                if (has1) im = X[((size_t)v1 * D + b) * m + n];
            }
            tile[(size_t)n * ld + b] = c_make(re, im);
        }
    }
    __syncthreads();
    const int cols = mode == 1 ? 1 : D;
    fft_tile_forward(tile, plan, cols, ld, tw, tid, nthr);

    if (mode == 1) {
        const double scale = 1.0 / (double)L;
        for (int pos = tid; pos < L; pos += nthr) {
            const cplx z = tile[(size_t)pos * ld];
            spec_out[(size_t)v0 * L + pos] = z.x * scale;
            if (has1) spec_out[(size_t)v1 * L + pos] = z.y * scale;
        }
        return;
    }
    for (int pos = tid; pos < L; pos += nthr) {
        cplx* zp = tile + (size_t)pos * ld;
        cplx z[D];
#pragma unroll
        for (int b = 0; b < D; ++b) z[b] = zp[b];
        mix_point<D>(z, mp, (size_t)L, (size_t)pos);
#pragma unroll
        for (int a = 0; a < D; ++a) zp[a] = z[a];
    }
    __syncthreads();
    fft_tile_adjoint(tile, plan, D, ld, tw, tid, nthr);

    for (int w = tid; w < D * m; w += nthr) {
        const int b = w / m, n = w - b * m;
        const cplx z = tile[(size_t)n * ld + b];
        Y[((size_t)v0 * D + b) * m + n] = z.x;
        if (has1) Y[((size_t)v1 * D + b) * m + n] = z.y;
    }
}
