// Third-generation grid-product kernels (1-D and 2-D grids, same flow graph and
// same intermediates layout as rl_kernels2.h).  What changed, and why
// (profiles/r02: the k2 kernels sat at 16 % VALU / 53 % wait, with 37 % of
// their LDS cycles lost to bank conflicts and a ds_write_b128 costing 13
// cycles against 4 for the read):
//
//   * FEWER LDS ROUND TRIPS.  A row transform is N2 = RA * RB * 2 with RA, RB in
//     {8, 16}: pass A runs in registers straight from the global load, pass B
//     crosses LDS once, and the final radix-2 pass is done by the MIX threads
//     themselves (a thread owns the two positions 2g, 2g+1 of all D outputs:
//     butterfly, real D x D mix at both frequencies, adjoint butterfly), so a
//     row tile crosses LDS four times instead of six.
//   * NO BANK CONFLICTS.  Transforms sit one after another (tile[col][pos]) and
//     the position is XOR-swizzled, pos ^ ((pos >> log2 RB) & 15) ^
//     ((pos >> (log2 RB - 1)) & 1), which makes every access pattern of the three
//     phases conflict-free for 16-byte accesses (tools/lds_conflicts.py is the
//     bank model and the search that found it; the unswizzled layout costs 8x in
//     pass B).
//   * NO PADDING, so the C5 row tile (D = 10, N2 = 512) is exactly 80 KiB and
//     TWO workgroups share a CU: one computes while the other waits on HBM.
//   * Workgroup sizes follow the work (320 threads for 10 x 32 butterflies), not
//     a fixed 256 / 512 that left 37 % of the lanes idle in every pass.
#pragma once
#include "rl_kernels2.h"

#define RL_THREADS3 512
// Register budget of the row kernel: 128 VGPRs (four waves per SIMD) while D
// leaves room, so that TWO 320-thread workgroups share a CU at C5 (a workgroup's
// waves go to the SIMDs cyclically from the same start: two five-wave
// workgroups need room for four waves on one SIMD; measured with a residency
// census, profiles/r02).  The emulator build has no such attribute.
#if defined(RL_EMU)
#define RL_K3_ROWS_ATTR
#else
// (RL_K3_WPE_SMALL: experiment builds -- waves per SIMD of the instantiations with D <= 5,
// whose 128-register versions spill 14-16 registers: tools/r05_k3_ab.sh)
#if !defined(RL_K3_WPE_SMALL)
#define RL_K3_WPE_SMALL 4
#endif
#define RL_K3_ROWS_ATTR __attribute__((amdgpu_waves_per_eu(D <= 12 ? (D <= 5 ? RL_K3_WPE_SMALL : 4) : 2)))
#endif

template <int RB>
__device__ __forceinline__ int swz3(int p) {
    constexpr int LB = RB == 16 ? 4 : 3;
    return p ^ ((p >> LB) & 15) ^ ((p >> (LB - 1)) & 1);
}

// v[k] *= w^k (CONJ: conj(w)^k), k = 1 .. R-1, with w^1, w^2, w^4, w^8 taken from
// the table (exact entries) and every other power from at most three
// multiplications: 4 loads instead of R - 1 per thread (the twiddle tables are
// L1-resident, but 15 distinct 16-byte loads per lane and pass cost more L1
// bandwidth than the data itself), powers formed just before they are used
struct TwiddleSeed {
    cplx w1, w2, w4, w8;
};
template <int R>
__device__ __forceinline__ TwiddleSeed twiddle_seed(const cplx* __restrict__ tab, int j) {
    TwiddleSeed t;
    t.w1 = tab[j];
    t.w2 = tab[2 * j];
    t.w4 = tab[4 * j];
    t.w8 = R == 16 ? tab[8 * j] : t.w4;
    return t;
}
template <bool CONJ>
__device__ __forceinline__ cplx c_mul_maybe_conj(cplx a, cplx b) {
    return CONJ ? c_mulc(a, b) : c_mul(a, b);
}
template <int R, bool CONJ>
__device__ __forceinline__ void apply_twiddle_powers(cplx* v, const TwiddleSeed& t) {
    static_assert(R == 8 || R == 16, "radix");
    const cplx w3 = c_mul(t.w2, t.w1), w5 = c_mul(t.w4, t.w1), w6 = c_mul(t.w4, t.w2);
    const cplx w7 = c_mul(t.w4, w3);
    v[1] = c_mul_maybe_conj<CONJ>(v[1], t.w1);
    v[2] = c_mul_maybe_conj<CONJ>(v[2], t.w2);
    v[3] = c_mul_maybe_conj<CONJ>(v[3], w3);
    v[4] = c_mul_maybe_conj<CONJ>(v[4], t.w4);
    v[5] = c_mul_maybe_conj<CONJ>(v[5], w5);
    v[6] = c_mul_maybe_conj<CONJ>(v[6], w6);
    v[7] = c_mul_maybe_conj<CONJ>(v[7], w7);
    if (R == 16) {
        v[8] = c_mul_maybe_conj<CONJ>(v[8], t.w8);
        v[9] = c_mul_maybe_conj<CONJ>(v[9], c_mul(t.w8, t.w1));
        v[10] = c_mul_maybe_conj<CONJ>(v[10], c_mul(t.w8, t.w2));
        v[11] = c_mul_maybe_conj<CONJ>(v[11], c_mul(t.w8, w3));
        v[12] = c_mul_maybe_conj<CONJ>(v[12], c_mul(t.w8, t.w4));
        v[13] = c_mul_maybe_conj<CONJ>(v[13], c_mul(t.w8, w5));
        v[14] = c_mul_maybe_conj<CONJ>(v[14], c_mul(t.w8, w6));
        v[15] = c_mul_maybe_conj<CONJ>(v[15], c_mul(t.w8, w7));
    }
}

// Mix tables, rebuilt whenever the operator's parameters change (k_mix_tables):
//   dc[a][pos] = sum_q kappa_q[a] * spec_q[pos]          (D rows)
//   gs[f][pos] = facW[f] * spec_{facQ[f]}[pos]           (nfac rows)
// so that the per-frequency mix  y_a = dc_a z_a + sum_f A_f[a] gs_f (A_f . z)
// reads D + nfac table values per position and does no arithmetic on the
// spectra (as loops over q and f with the spectrum loads inside, the mix was a
// chain of Q + nfac dependent memory latencies plus Q D multiply-adds per
// point: 6.9 of the row kernel's 21 us at C5, profiles/r02).
static __global__ void __launch_bounds__(256)
k_mix_tables(MixParams mp, int D, int L, double* __restrict__ dc, double* __restrict__ gs) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= L) return;
    const int row = blockIdx.y;
    if (row < D) {
        double acc = 0.0;
        for (int q = 0; q < mp.Q; ++q)
            acc = fma(mp.kappa[q * D + row], mp.spec[(size_t)q * L + pos], acc);
        dc[(size_t)row * L + pos] = acc;
    } else {
        const int f = row - D;
        gs[(size_t)f * L + pos] = mp.facW[f] * mp.spec[(size_t)mp.facQ[f] * L + pos];
    }
}

// the real D x D mix at two adjacent positions o, o + 1 (o even) on one
// component (real or imaginary) of the D values: z <- M(o) z, w <- M(o + 1) w
#define RL_MIXF 6
template <int D>
__device__ __forceinline__ void mix_real2(double* z, double* w, const MixParams& mp, size_t L,
                                          size_t o) {
    if (mp.dc != nullptr && mp.nfac <= RL_MIXF) {
        // factor sums first (they need the unmixed z), then z in place
        cplx g[RL_MIXF];
#pragma unroll
        for (int f = 0; f < RL_MIXF; ++f) {
            const int ff = f < mp.nfac ? f : 0;            // unconditional, clamped
            g[f] = mp.nfac > 0 ? *reinterpret_cast<const cplx*>(mp.gs + (size_t)ff * L + o)
                               : c_make(0.0, 0.0);
        }
        double sz[RL_MIXF], sw[RL_MIXF];
#pragma unroll
        for (int f = 0; f < RL_MIXF; ++f) {
            sz[f] = 0.0;
            sw[f] = 0.0;
            if (f < mp.nfac) {
                const double* af = mp.facA + (size_t)f * D;
                double z0 = 0.0, z1 = 0.0, w0 = 0.0, w1 = 0.0;    // two chains each
#pragma unroll
                for (int b = 0; b + 1 < D; b += 2) {
                    z0 = fma(af[b], z[b], z0);
                    w0 = fma(af[b], w[b], w0);
                    z1 = fma(af[b + 1], z[b + 1], z1);
                    w1 = fma(af[b + 1], w[b + 1], w1);
                }
                if (D & 1) {
                    z0 = fma(af[D - 1], z[D - 1], z0);
                    w0 = fma(af[D - 1], w[D - 1], w0);
                }
                sz[f] = (z0 + z1) * g[f].x;
                sw[f] = (w0 + w1) * g[f].y;
            }
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const cplx d = *reinterpret_cast<const cplx*>(mp.dc + (size_t)a * L + o);
            z[a] *= d.x;
            w[a] *= d.y;
        }
#pragma unroll
        for (int f = 0; f < RL_MIXF; ++f) {
            if (f < mp.nfac) {
                const double* af = mp.facA + (size_t)f * D;
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    z[a] = fma(af[a], sz[f], z[a]);
                    w[a] = fma(af[a], sw[f], w[a]);
                }
            }
        }
        return;
    }
    double yz[D], yw[D];
    {
        double dz[D], dw[D];
#pragma unroll
        for (int a = 0; a < D; ++a) { dz[a] = 0.0; dw[a] = 0.0; }
        for (int q = 0; q < mp.Q; ++q) {
            const cplx sp = *reinterpret_cast<const cplx*>(mp.spec + (size_t)q * L + o);
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const double k = mp.kappa[q * D + a];
                dz[a] = fma(k, sp.x, dz[a]);
                dw[a] = fma(k, sp.y, dw[a]);
            }
        }
#pragma unroll
        for (int a = 0; a < D; ++a) { yz[a] = z[a] * dz[a]; yw[a] = w[a] * dw[a]; }
        for (int f = 0; f < mp.nfac; ++f) {
            const double* af = mp.facA + (size_t)f * D;
            double sz = 0.0, sw = 0.0;
#pragma unroll
            for (int b = 0; b < D; ++b) {
                sz = fma(af[b], z[b], sz);
                sw = fma(af[b], w[b], sw);
            }
            const cplx sp = *reinterpret_cast<const cplx*>(mp.spec + (size_t)mp.facQ[f] * L + o);
            const double fw = mp.facW[f];
            sz *= fw * sp.x;
            sw *= fw * sp.y;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                yz[a] = fma(af[a], sz, yz[a]);
                yw[a] = fma(af[a], sw, yw[a]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < D; ++a) { z[a] = yz[a]; w[a] = yw[a]; }
}

// ---------------------------------------------------------------------------
// k3_rows_mix<D, RA, RB>: N2 = RA * RB * 2.  grid (N1 / R, npairs), any block
// size that is a multiple of 64.  LDS: tile [R * D][N2] complex, no padding.
// Plan convention: plan2.radix = {RA, RB, 2}; spectra are in that plan's
// scrambled order (made by k_rows_spec with the same plan).
// ---------------------------------------------------------------------------
template <int D, int RA, int RB>
__global__ void __launch_bounds__(RL_THREADS3) RL_K3_ROWS_ATTR
k3_rows_mix(cplx* __restrict__ T, Tile2 tp, const cplx* __restrict__ tw2,
            const int* __restrict__ freq1, TwiddleL twl, MixParams mp, int* __restrict__ bump) {
    if (bump != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *bump += 1;
    RL_CENSUS_ENTER(100);
    RL_SMEM(smem);
    cplx* tile = reinterpret_cast<cplx*>(smem);
    constexpr int N2 = RA * RB * 2;
    constexpr int SA = N2 / RA;       // butterflies of pass A per transform
    constexpr int NBF = N2 / RB;      // butterflies of pass B per transform
    constexpr int H = N2 / 2;         // mix items per row
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int N1 = tp.N1, R = tp.R;
    const int cols = R * D;
    // (row tile, pair) of this workgroup.  The mix tables of a row are the same
    // for every pair, so the tiles that share a row should meet in one XCD's L2:
    // with a tile count that is a multiple of 8, XCD k (linear block id mod 8,
    // the dispatcher's observed placement) takes the row tiles k, k + 8, ... and
    // runs the pairs of each one after another.  Otherwise: plain 2-D order.
    int rt = blockIdx.x, pair = blockIdx.y;
    if (tp.aff > 0) {
        if (!affine_tile(tp, N1 / R, &rt, &pair)) return;       // (pair-affine order, rl_kernels2.h)
    } else if ((gridDim.x & 7) == 0 && gridDim.y > 1) {
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = b & 7, slot = b >> 3;
        rt = (slot / (int)gridDim.y) * 8 + xcd;
        pair = slot % (int)gridDim.y;
    }
    const int r0 = rt * R;
    const size_t L = (size_t)N1 * N2;
    cplx* base = T + (size_t)pair * D * L;
    RL_STAMP_AT(60, 200, 1);

    // pass A: registers, straight from global
    for (int w = tid; w < cols * SA; w += nthr) {
        const int j = w % SA, col = w / SA;
        const int rr = col / D, b = col - rr * D;
        const cplx* src = base + (size_t)b * L + (size_t)(r0 + rr) * N2;
        cplx v[RA];
#pragma unroll
        for (int i = 0; i < RA; ++i) v[i] = src[j + SA * i];
        const TwiddleSeed ts = twiddle_seed<RA>(tw2, j);
        SmallDft<RA, false>::run(v);
        apply_twiddle_powers<RA, false>(v, ts);
        cplx* tc = tile + (size_t)col * N2;
#pragma unroll
        for (int k = 0; k < RA; ++k) tc[swz3<RB>(j + SA * k)] = v[k];
    }
    RL_STAMP_AT(61, 200, 1);
    __syncthreads();
    RL_STAMP_AT(62, 200, 1);
    // pass B: sub-transforms of length SA = 2 RB, butterfly distance 2
    for (int w = tid; w < cols * NBF; w += nthr) {
        const int bf = w % NBF, col = w / NBF;
        const int jj = bf & 1, g = (bf >> 1) * SA + jj;
        cplx* tc = tile + (size_t)col * N2;
        cplx v[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) v[i] = tc[swz3<RB>(g + 2 * i)];
        SmallDft<RB, false>::run(v);
        if (jj) {
#pragma unroll
            for (int k = 1; k < RB; ++k) v[k] = c_mul(v[k], tw2[k * RA]);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) tc[swz3<RB>(g + 2 * i)] = v[i];
    }
    RL_STAMP_AT(63, 200, 1);
    __syncthreads();
    RL_STAMP_AT(64, 200, 1);
    // final radix-2 pass + mix + its adjoint.  The mix is REAL, so real and
    // imaginary parts are separate work items (adjacent lanes): half the
    // registers per thread and twice the parallelism of a complex item.
#if defined(RL_TIMING) && !defined(RL_EMU)
    if (rl_timing_buf[120] != 2)
#endif
    for (int w = tid; w < R * H * 2; w += nthr) {
        const int part = w & 1, it = w >> 1;
        const int gp = it % H, rr = it / H;
        const int p0 = swz3<RB>(2 * gp), p1 = swz3<RB>(2 * gp + 1);
        double* tc = reinterpret_cast<double*>(tile + (size_t)rr * D * N2) + part;
        const size_t o = (size_t)(r0 + rr) * N2 + 2 * gp;
        double s[D], t[D];
#pragma unroll
        for (int b = 0; b < D; ++b) {
            const double a0 = tc[2 * ((size_t)b * N2 + p0)], a1 = tc[2 * ((size_t)b * N2 + p1)];
            s[b] = a0 + a1;
            t[b] = a0 - a1;
        }
#if defined(RL_TIMING) && !defined(RL_EMU)
        if (rl_timing_buf[120] != 1)
#endif
        mix_real2<D>(s, t, mp, L, o);
#pragma unroll
        for (int b = 0; b < D; ++b) {
            tc[2 * ((size_t)b * N2 + p0)] = s[b] + t[b];
            tc[2 * ((size_t)b * N2 + p1)] = s[b] - t[b];
        }
    }
    RL_STAMP_AT(65, 200, 1);
    __syncthreads();
    RL_STAMP_AT(66, 200, 1);
    // adjoint of pass B
    for (int w = tid; w < cols * NBF; w += nthr) {
        const int bf = w % NBF, col = w / NBF;
        const int jj = bf & 1, g = (bf >> 1) * SA + jj;
        cplx* tc = tile + (size_t)col * N2;
        cplx v[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) v[k] = tc[swz3<RB>(g + 2 * k)];
        if (jj) {
#pragma unroll
            for (int k = 1; k < RB; ++k) v[k] = c_mulc(v[k], tw2[k * RA]);
        }
        SmallDft<RB, true>::run(v);
#pragma unroll
        for (int i = 0; i < RB; ++i) tc[swz3<RB>(g + 2 * i)] = v[i];
    }
    RL_STAMP_AT(67, 200, 1);
    __syncthreads();
    RL_STAMP_AT(68, 200, 1);
    // adjoint of pass A, conjugate inter-step twiddle, straight to global
    for (int w = tid; w < cols * SA; w += nthr) {
        const int j = w % SA, col = w / SA;
        const int rr = col / D, b = col - rr * D;
        cplx* dst = base + (size_t)b * L + (size_t)(r0 + rr) * N2;
        const cplx* tc = tile + (size_t)col * N2;
        cplx v[RA];
#pragma unroll
        for (int k = 0; k < RA; ++k) v[k] = tc[swz3<RB>(j + SA * k)];
        apply_twiddle_powers<RA, true>(v, twiddle_seed<RA>(tw2, j));
        SmallDft<RA, true>::run(v);
        if (twl.lo != nullptr) {
            // conj W_L^{k1 (j + SA i)} = conj(W_L^{k1 j} * W_L^{k1 SA i}): one
            // per-lane table value and RA values that are the same for a whole row
            const int k1 = freq1[r0 + rr];
            const cplx wj = twiddle_L(twl, k1 * j);
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const cplx wi = i == 0 ? wj : c_mul(wj, twiddle_L(twl, k1 * SA * i));
                v[i] = c_mulc(v[i], wi);
            }
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) dst[j + SA * i] = v[i];
    }
    RL_STAMP_AT(69, 200, 1);
    RL_CENSUS_LEAVE(100);
}
