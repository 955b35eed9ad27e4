// Recursive-filter form of the grid product for EXPONENTIAL-POLYNOMIAL top rows.
//
// A Toeplitz block whose first row is
//
//     t_i = (c0 + c1 i + c2 i^2) rho^i,      0 < rho <= 1,
//
// -- the Matern-3/2 kernel (1 + a r) exp(-a r) of the reference on a regular grid
// (runlmc/kern/matern32.py:40-42: rho = exp(-a h), c1 = a h c0), its derivative
// with respect to the inverse length scale (-3 gamma r^2 exp(-a r),
// matern32.py:50-55: c2 only), a plain exponential -- is EXACTLY semiseparable:
// with the causal sums  F^k_i = sum_{j <= i} (i - j)^k rho^(i - j) x_j  and the
// anti-causal ones  H^k_i = sum_{j >= i} (j - i)^k rho^(j - i) x_j,
//
//     (T x)_i = c0 (F^0_i + H^0_i - x_i) + c1 (F^1_i + H^1_i) + c2 (F^2_i + H^2_i),
//     F^0_i = rho F^0_{i-1} + x_i,   F^1_i = rho (F^1_{i-1} + F^0_{i-1}),
//     F^2_i = rho (F^2_{i-1} + 2 F^1_{i-1} + F^0_{i-1})          (H: mirrored).
//
// So  K x = sum_q B_q (x) T_q x  for such tops is a block-parallel scan that
// reads x and writes y -- no zero-padded complex intermediates, no transform:
//
//   k_sf_carries   per chunk of RL_SF_G grid points and row: the state the chunk
//                  alone leaves at its last point (F) and at its first point (H)
//                  -- 2 NS weighted sums per filter, lanes along the grid;
//   k_sf_scan      per (vector, channel, direction): the chunks' states chained
//                  ( state' = rho^G (F0, F1 + G F0, F2 + 2 G F1 + G^2 F0) + chunk ),
//                  which gives every chunk the state it starts from;
//   k_sf_apply     per (vector, chunk): the D rows of the chunk in LDS, rank-one
//                  factors mixed there (u_f = A_f . x); a row is cut into 16-point
//                  blocks, inside which a filter is a small dense map -- these run
//                  on the fp64 matrix cores (block Toeplitz part, states a block
//                  leaves, response to the states it receives), the states are
//                  chained over the row's 32 blocks with DPP row shifts;
//                  y_a = sum_q kappa_q[a] T_q x_a + sum_f w_f A_f[a] T_q(f) u_f
//                  assembled in LDS and stored.
//
// Every factor a state is multiplied by is a power rho^n that the host computed
// in long double and rounded once (rho^(16 c) between blocks, rho^G between
// chunks): no product of 1e5 rounded rho's ever forms, and the form agrees with a
// long-double evaluation to 1e-15 of |T|_1 |x|_inf (tests; the transform kernels:
// 1e-13).
//
// WHICH tops take this form is decided on the host at set time, from the top row
// itself (runlmc_hip.hip: sf_detect): the parameters are fitted from four
// samples and the fit is accepted only if  sum_i |t_i - model_i| <= 2e-14 sum_i |t_i|
// over the WHOLE row -- a bound on ||T - T_model||_1, hence on the product's
// error for every input (no trial vectors involved).
// Reference semantics: runlmc/linalg/bttb.py:144-148, kronecker.py:39-46.
#pragma once
#include "rl_device.h"

#define RL_SF_G 512                        // grid points per chunk (256 or 512: one or two
                                           // 16-block halves per row in k_sf_apply; measured
                                           // at C5 Matern, 256 with three workgroups per CU at
                                           // 160 registers: apply 0.77 vs 0.76 ms, carries
                                           // 0.30 vs 0.26, scan 0.17 vs 0.08 -- 512 kept)
#define RL_SF_NH (RL_SF_G / 256)
#define RL_SF_S 16                         // points per block
#define RL_SF_LPR (RL_SF_G / RL_SF_S)      // blocks per row of a chunk
#define RL_SF_PAD (RL_SF_G + RL_SF_LPR)    // doubles per LDS row: one pad per block
#define RL_SF_MAXTOPS 16                   // filter tops per operator at most
#define RL_SF_TOL 2e-14                    // accepted sum|t - model| / sum|t|

// position of grid point i of a chunk inside its padded LDS row: a block starts
// 17 doubles after its neighbour, which spreads the 16 columns a matrix fragment
// reads over the banks (ds_read_b64: (17 * 2 * c) mod 64 are distinct even banks)
__device__ __forceinline__ int sf_pad(int i) { return i + (i >> 4); }

struct SfTop {
    double rho;       // decay per grid step
    double c[3];      // t_i = (c0 + c1 i + c2 i^2) rho^i
    double rG;        // rho^G: from chunk to chunk
};

// one operator's filter part (device pointers; NF tops, nfac rank-one factors)
struct SfParams {
    int NF, nfac;
    const SfTop* tops;        // [NF]
    const double* pw;         // [NF][G + 1]: rho^j
    const double* kappa;      // [NF][D] weight of top j on the diagonal of output a
    const double* facA;       // [nfac][D]
    const double* facAW;      // [nfac][D]: w_f A_f
    const int* facJ;          // [nfac]: top of factor f
};

// one grid step of a causal state (the same code runs the anti-causal one over
// descending points); returns rho * F0_old, which is F0_new - x
template <int NS>
__device__ __forceinline__ double sf_step(double (&F)[NS], double rho, double x) {
    const double tt = rho * F[0];
    if constexpr (NS == 3) {
        const double t1 = rho * F[1];
        F[2] = fma(rho, F[2], fma(2.0, t1, tt));
        F[1] = t1 + tt;
    } else {
        F[1] = fma(rho, F[1], tt);
    }
    F[0] = tt + x;
    return tt;
}
// V += r M(n) S:  a state S carried n grid steps further (r = rho^n) added to V
template <int NS>
__device__ __forceinline__ void sf_carry(double (&V)[NS], const double (&S)[NS], double r, double n) {
    V[0] = fma(r, S[0], V[0]);
    V[1] = fma(r, fma(n, S[0], S[1]), V[1]);
    if constexpr (NS == 3) V[2] = fma(r, fma(n * n, S[0], fma(2.0 * n, S[1], S[2])), V[2]);
}

// ---------------------------------------------------------------------------
// sums of NV values per lane over the 64 lanes of a wave.  Afterwards the lanes
// with (lane & 7) == 0 hold NV / 8 of the sums each, out[k] = sum number j0 + k.
// GPU: halving butterfly (at distance 32 a lane keeps one half of its values and
// receives that half from its partner, at 16 a quarter, at 8 an eighth; the last
// NV / 8 go through plain exchanges) -- NV / 2 + NV / 4 + NV / 8 + 3 NV / 8
// cross-lane moves instead of 6 NV.  Emulator (no cross-lane operations): LDS.
// ---------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ int sf_wave_sums(const double (&acc)[NV], double* red, double (&out)[NV / 8]) {
    constexpr int A = NV / 2, B = NV / 4, C = NV / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool h5 = (lane & 32) != 0, h4 = (lane & 16) != 0, h3 = (lane & 8) != 0;
    const int j0 = (h5 ? A : 0) + (h4 ? B : 0) + (h3 ? C : 0);
#if defined(RL_EMU)
    const int wave = tid >> 6;
    for (int j = 0; j < NV; ++j) {
        red[tid] = acc[j];
        __syncthreads();
        if ((lane & 7) == 0 && j >= j0 && j < j0 + C) {
            double s = 0.0;
            for (int l = 0; l < 64; ++l) s += red[wave * 64 + l];
            out[j - j0] = s;
        }
        __syncthreads();
    }
#else
    (void)red;
    double a[A], b[B], c[C];
#pragma unroll
    for (int k = 0; k < A; ++k) {
        const double mine = h5 ? acc[A + k] : acc[k], send = h5 ? acc[k] : acc[A + k];
        a[k] = mine + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
        const double mine = h4 ? a[B + k] : a[k], send = h4 ? a[k] : a[B + k];
        b[k] = mine + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const double mine = h3 ? b[C + k] : b[k], send = h3 ? b[k] : b[C + k];
        c[k] = mine + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
        c[k] += __shfl_xor(c[k], 4, 64);
        c[k] += __shfl_xor(c[k], 2, 64);
        c[k] += __shfl_xor(c[k], 1, 64);
        out[k] = c[k];
    }
#endif
    return j0;
}

// ---------------------------------------------------------------------------
// k_sf_carries<NS>: E[chunk][row][j][dir][k], the state chunk `chunk` of row
// `row` alone leaves behind under filter j:
//   dir 0 (causal, at the chunk's LAST point):   sum_t (G-1-t)^k rho^(G-1-t) x_t
//   dir 1 (anti-causal, at its FIRST point):     sum_t t^k rho^t x_t
// (t: position in the chunk; points past the end of the grid count as zero).
//   grid (nchunks, ceil(nrows / rows_per_wg))   block 256   rows_per_wg % 16 == 0
//   LDS: NF (G + 1) doubles of rho^j  (+ 256 doubles for the emulator's sums)
// Lanes run along the grid (every load is 512 contiguous bytes of a row); a wave
// owns four rows at a time, keeps their 4 x 8 values in registers and walks the
// filters: 2 NS multiply-adds per value and filter, weights from LDS shared by
// the four rows.
// ---------------------------------------------------------------------------
// requests RB rows' share of a chunk (lane: points lane + 64 k) from clamped addresses
template <int NK, int RB>
__device__ __forceinline__ void sf_request_rows(double (&xv)[NK][RB], const double* __restrict__ X,
                                                int nrows, int m, int r0, int g0, int lane) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = r0 + r;
        const double* xr = X + (size_t)(row < nrows ? row : nrows - 1) * m;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int gi = g0 + lane + 64 * k;
            xv[k][r] = xr[gi < m ? gi : m - 1];
        }
    }
}

template <int NS>
__global__ void __launch_bounds__(256)
k_sf_carries(const double* __restrict__ X, int nrows, int m, int NF, const double* __restrict__ pw,
             int rows_per_wg, double* __restrict__ E) {
    constexpr int RB = 4, NK = RL_SF_G / 64, NV = RB * 2 * NS, G = RL_SF_G;
    RL_SMEM(smem);
    double* pwl = reinterpret_cast<double*>(smem);               // [NF][G + 1]
    double* red = pwl + (size_t)NF * (G + 1);                    // emulator only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, rbase = blockIdx.y * rows_per_wg, g0 = chunk * G;
    for (int e = tid; e < NF * (G + 1); e += 256) pwl[e] = pw[e];
    __syncthreads();
    // (the next four rows are requested before the current four are worked on: the wave's
    // loads overlap its sums)
    double xn[NK][RB];
    sf_request_rows<NK, RB>(xn, X, nrows, m, rbase + wave * RB, g0, lane);
    for (int r0 = rbase + wave * RB; r0 < rbase + rows_per_wg; r0 += 4 * RB) {
        double xv[NK][RB];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const double live = (r0 + r < nrows && g0 + lane + 64 * k < m) ? 1.0 : 0.0;
                xv[k][r] = xn[k][r] * live;
            }
        if (r0 + 4 * RB < rbase + rows_per_wg)
            sf_request_rows<NK, RB>(xn, X, nrows, m, r0 + 4 * RB, g0, lane);
        for (int q = 0; q < NF; ++q) {
            const double* p = pwl + (size_t)q * (G + 1);
            double acc[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) acc[j] = 0.0;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int t = lane + 64 * k;
                const double tb = (double)t, tf = (double)(G - 1 - t);
                double wf[NS], wb[NS];
                wb[0] = p[t];
                wf[0] = p[G - 1 - t];
#pragma unroll
                for (int s = 1; s < NS; ++s) {
                    wb[s] = wb[s - 1] * tb;
                    wf[s] = wf[s - 1] * tf;
                }
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const double x = xv[k][r];
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        acc[r * 2 * NS + s] = fma(wf[s], x, acc[r * 2 * NS + s]);
                        acc[r * 2 * NS + NS + s] = fma(wb[s], x, acc[r * 2 * NS + NS + s]);
                    }
                }
            }
            double out[NV / 8];
            const int j0 = sf_wave_sums<NV>(acc, red, out);
            if ((lane & 7) == 0) {
#pragma unroll
                for (int k = 0; k < NV / 8; ++k) {
                    const int j = j0 + k, r = j / (2 * NS), rest = j - r * 2 * NS;
                    if (r0 + r < nrows)
                        E[(((size_t)chunk * nrows + r0 + r) * NF + q) * 2 * NS + rest] = out[k];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_sf_scan<NS>: Cin[chunk][v][chan][dir][k], the state a chunk STARTS from:
//   dir 0: the causal state at the point before the chunk (chunks 0 .. chunk-1),
//   dir 1: the anti-causal state at the point after it (chunks chunk+1 ..).
// Channels of a vector: (output a, top j) -> a NF + j for the diagonal part, then
// one per rank-one factor f, whose chunk states are  sum_b A_f[b] E(row b)  by
// linearity.
//   grid (ceil(2 nchan / 8), nvec)   block 256 = 8 (channel, direction) x 32 segments
// The chunks of a (channel, direction) are cut into 32 segments: a thread chains
// its segment from a zero state, the 32 segment totals are chained through LDS,
// and the thread walks its segment again from the right state -- 2 nchunks / 32
// dependent steps instead of nchunks (C5: 196 chunks; one thread per channel
// took 3 ms, every step a memory round trip; 16 segments 80 us).
// ---------------------------------------------------------------------------
template <int NS>
__device__ __forceinline__ void sf_chunk_state(const double* __restrict__ E, const SfParams& sp,
                                               int c, int nrows, int row0, int D, bool diag, int a,
                                               int f, int j, int dir, double (&e)[NS]) {
    const int NF = sp.NF;
    if (diag) {
        const double* src = E + ((((size_t)c * nrows + row0 + a) * NF + j) * 2 + dir) * NS;
#pragma unroll
        for (int k = 0; k < NS; ++k) e[k] = src[k];
    } else {
#pragma unroll
        for (int k = 0; k < NS; ++k) e[k] = 0.0;
        for (int b = 0; b < D; ++b) {
            const double w = sp.facA[(size_t)f * D + b];
            const double* src = E + ((((size_t)c * nrows + row0 + b) * NF + j) * 2 + dir) * NS;
#pragma unroll
            for (int k = 0; k < NS; ++k) e[k] = fma(w, src[k], e[k]);
        }
    }
}

template <int NS>
__global__ void __launch_bounds__(256)
k_sf_scan(const double* __restrict__ E, int nchunks, int nvec, int D, SfParams sp,
          double* __restrict__ Cin) {
    RL_SMEM(smem);
    constexpr int NCD = 8, NSEG = 32;
    double* agg = reinterpret_cast<double*>(smem);       // [NSEG][NCD][NS]
    const int NF = sp.NF, nchan = D * NF + sp.nfac, ncd = 2 * nchan;
    const int tid = threadIdx.x, cdl = tid & (NCD - 1), seg = tid / NCD, v = blockIdx.y;
    const int cdr = blockIdx.x * NCD + cdl;
    const bool live = cdr < ncd;
    const int cd = live ? cdr : ncd - 1;
    const int dir = cd & 1, chan = cd >> 1;
    const bool diag = chan < D * NF;
    const int f = diag ? 0 : chan - D * NF;
    const int j = diag ? chan % NF : sp.facJ[f];
    const int a = diag ? chan / NF : 0;
    const double rG = sp.tops[j].rG, n = (double)RL_SF_G;
    const int nrows = nvec * D, row0 = v * D;
    const int seglen = (nchunks + NSEG - 1) / NSEG;
    const int p0 = seg * seglen < nchunks ? seg * seglen : nchunks;
    const int p1 = p0 + seglen < nchunks ? p0 + seglen : nchunks;
    // the segment alone
    double st[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) st[k] = 0.0;
    // (the chunk states do not depend on the chain: four steps' loads are requested together)
#pragma unroll 4
    for (int p = p0; p < p1; ++p) {
        const int c = dir == 0 ? p : nchunks - 1 - p;
        double e[NS];
        sf_chunk_state<NS>(E, sp, c, nrows, row0, D, diag, a, f, j, dir, e);
        sf_carry<NS>(e, st, rG, n);
#pragma unroll
        for (int k = 0; k < NS; ++k) st[k] = e[k];
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) agg[(seg * NCD + cdl) * NS + k] = st[k];
    __syncthreads();
    // the segments before this one (all of them full: seglen chunks each)
    double rL = 1.0;
    for (int i = 0; i < seglen; ++i) rL *= rG;
    const double nL = n * seglen;
#pragma unroll
    for (int k = 0; k < NS; ++k) st[k] = 0.0;
    for (int s = 0; s < seg; ++s) {
        double e[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) e[k] = agg[(s * NCD + cdl) * NS + k];
        sf_carry<NS>(e, st, rL, nL);
#pragma unroll
        for (int k = 0; k < NS; ++k) st[k] = e[k];
    }
    // the segment again, from the right state
#pragma unroll 4
    for (int p = p0; p < p1; ++p) {
        const int c = dir == 0 ? p : nchunks - 1 - p;
        if (live) {
            double* dst = Cin + ((((size_t)c * nvec + v) * nchan + chan) * 2 + dir) * NS;
#pragma unroll
            for (int k = 0; k < NS; ++k) dst[k] = st[k];
        }
        double e[NS];
        sf_chunk_state<NS>(E, sp, c, nrows, row0, D, diag, a, f, j, dir, e);
        sf_carry<NS>(e, st, rG, n);
#pragma unroll
        for (int k = 0; k < NS; ++k) st[k] = e[k];
    }
}

// ---------------------------------------------------------------------------
// Cross-lane primitives of k_sf_apply.  GPU: the fp64 matrix instruction and DPP
// row shifts; emulator (one fiber per lane, no cross-lane hardware): the same
// data movement through an LDS scratch of 128 doubles per wave.
// ---------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also drains the
// vector-memory counter, i.e. waits for the next tile's rows that k_sf_apply has in
// flight (measured: 7 us per tile at the first barrier after the request).  Every
// hand-over between the kernel's phases goes through LDS.
__device__ __forceinline__ void sf_lds_barrier() {
#if defined(RL_EMU)
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// value of lane (lane - N) (UP) or (lane + N) of the same 16-lane row, zero beyond it
template <int N, bool UP>
__device__ __forceinline__ double sf_row_shift(double v, double* scr) {
#if defined(RL_EMU)
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15;
    double* w = scr + (size_t)(tid >> 6) * 128;
    w[lane] = v;
    __syncthreads();
    const int src = UP ? c - N : c + N;
    const double r = (src >= 0 && src < 16) ? w[lane - c + src] : 0.0;
    __syncthreads();
    return r;
#else
    (void)scr;
    constexpr int ctrl = (UP ? 0x110 : 0x100) | N;        // row_shr:N / row_shl:N
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
#endif
}
// value of the LAST (lane 15) or the FIRST (lane 0) lane of the 16-lane row, in every lane of it
template <bool LAST>
__device__ __forceinline__ double sf_row_bcast(double v, double* scr) {
#if defined(RL_EMU)
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15;
    double* w = scr + (size_t)(tid >> 6) * 128;
    w[lane] = v;
    __syncthreads();
    const double r = w[lane - c + (LAST ? 15 : 0)];
    __syncthreads();
    return r;
#else
    (void)scr;
    // ds_swizzle, bit-mask mode inside groups of 32: lane' = (lane & 0x10) | (LAST ? 0x0f : 0)
    constexpr int pat = 0x10 | ((LAST ? 0x0F : 0x00) << 5);
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
    return __hiloint2double(hi, lo);
#endif
}

// value of the lane 16 further / nearer (lane ^ 16): the partner lane row of a pair
__device__ __forceinline__ double sf_row_partner(double v, double* scr) {
#if defined(RL_EMU)
    const int tid = threadIdx.x, lane = tid & 63;
    double* w = scr + (size_t)(tid >> 6) * 128;
    w[lane] = v;
    __syncthreads();
    const double r = w[lane ^ 16];
    __syncthreads();
    return r;
#else
    (void)scr;
    // ds_swizzle, bit-mask mode: and 0x1f, or 0, xor 0x10
    constexpr int pat = 0x1F | (0x10 << 10);
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
    return __hiloint2double(hi, lo);
#endif
}

// ---------------------------------------------------------------------------
// k_sf_apply<NS>: Y[v] = filter part of the operator applied to X[v].
//   grid (resident workgroups: each walks tiles (chunk, vector))   block 256
//   LDS: (D + nfac) RL_SF_PAD doubles of rows + NF sizeof(SfBlk) + (D + nfac) 16
//        (+ 128 doubles per wave for the emulator's cross-lane moves)
//
// A chunk row is a 16 x 32 matrix X (16 consecutive points per column).  Inside
// a 16-point block everything a filter does is a small dense map, and these maps
// run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), 16 columns at a time:
//
//   OUT  = Tblk X                 Tblk[i][k] = sum_j kappa_j[a] t_j(|i - k|): what the
//                                 block's own points contribute, ALL filters of the
//                                 row in one 16 x 16 symmetric Toeplitz block;
//   S    = Wst X                  the 2 NS states per filter the block alone leaves
//                                 at its last point (causal) / first point (anti-causal);
//   C    = scan of S over the 32 columns (+ the chunk's incoming states): the state
//          each block starts from -- DPP row shifts inside the 16 columns of a half,
//          lane 15 / lane 0 of the row carries it to the other half;
//   OUT += Rsp C                  Rsp[i][state]: what an incoming state contributes at
//                                 point i (t_j(n), rho^n (c1 + 2 c2 n), rho^n c2 at
//                                 n = i + 1 or 16 - i steps, times kappa_j[a]).
//
// The matrix instruction's D layout (row = (lane >> 4) + 4 reg, column = lane & 15)
// IS its B layout (k = 4 kk + (lane >> 4)): states come out of the second product in
// the registers the fourth one wants them in, and the scan in between runs along
// lane & 15 = along DPP rows.  A 16-lane row of lanes owns filter (lane >> 4) of
// the batch (and filter 4 + (lane >> 4) when NS == 2): registers
// R = NS (2 (filter / 4) + direction) + k.
// One wave = one row slot at a time: the D rows of x (every filter, weight
// kappa_j[a]) and the nfac mixed rows u_f = A_f . x (filter facJ[f], weight 1).
// ---------------------------------------------------------------------------
struct SfBlk {                // per filter, staged in LDS
    double tb[17];            // t(n) = (c0 + c1 n + c2 n^2) rho^n
    double r1[17];            // rho^n (c1 + 2 c2 n)
    double r2[17];            // rho^n c2
    double pw[17];            // rho^n
    double p16[17];           // rho^(16 c)
};
#define RL_SF_BLKD ((int)(sizeof(SfBlk) / sizeof(double)))
// Everything k_sf_apply needs about the operator, one block of doubles built at
// set time and copied to LDS by every workgroup:
//   kappa [NF][D] | facA [nfac][D] | facAW [nfac][D] | facJ [nfac] | block tops
//   [D + nfac][16] (row slot a: sum_j kappa_j[a] t_j(n); slot D + f: t_facJ[f](n)) | SfBlk [NF]
__host__ __device__ inline int sf_blob_doubles(int NF, int nfac, int D) {
    return NF * D + 2 * nfac * D + nfac + (D + nfac) * 16 + NF * RL_SF_BLKD;
}

#if defined(RL_EMU)
#define RL_SF_APPLY_ATTR
#else
// two workgroups per CU = two waves per SIMD: 256 registers, the next tile's rows among them
#if RL_SF_G == 256
#define RL_SF_APPLY_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))
#else
#define RL_SF_APPLY_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#endif

// requests a tile's D rows (thread tid: point tid + 256 (k % NH) of row k / NH in
// xr[k], from clamped addresses -- points past the grid are zeroed when the registers
// go to LDS, not here: a select right behind a load makes the compiler wait for it,
// measured as twelve serial round trips) and its incoming states into registers
template <int XR>
__device__ __forceinline__ void sf_request(double (&xr)[XR], double (&cr)[4],
                                           const double* __restrict__ X,
                                           const double* __restrict__ Cin, int tile, int nch,
                                           int nvec, int D, int m, int ncin, int tid) {
    const int chunk = tile % nch, v = tile / nch, g0 = chunk * RL_SF_G;
    const double* xbase = X + (size_t)v * D * m;
#pragma unroll
    for (int k = 0; k < XR; ++k) {
        if (k / RL_SF_NH < D) {
            const int gi = g0 + tid + 256 * (k % RL_SF_NH);
            xr[k] = xbase[(size_t)(k / RL_SF_NH) * m + (gi < m ? gi : m - 1)];
        }
    }
    const double* src = Cin + ((size_t)chunk * nvec + v) * ncin;
#pragma unroll
    for (int k = 0; k < 4; ++k) cr[k] = src[tid + 256 * k < ncin ? tid + 256 * k : ncin - 1];
}

// A fragment of the states product: state id = 16 mt + col is register R = id / 4 of the
// lane row id % 4 -> filter fb = id % 4 + 4 (R / (2 NS)) of the batch (j0 + fb, or the one
// filter jbase of a mixed row), direction (R / NS) % 2, power R % NS; point k = 4 kk + lg
template <int NS>
__device__ __forceinline__ double sf_state_weight(const SfBlk* bl, int mt, int kk, int col, int lg,
                                                  int nfb, int jbase, bool batch) {
    const int id = 16 * mt + col, R = id >> 2;
    const int fb = (id & 3) + 4 * (R / (2 * NS)), dir = (R / NS) & 1, ks = R % NS;
    const bool on = fb < nfb;
    const int jf = batch ? jbase + (on ? fb : 0) : jbase;
    const int k = 4 * kk + lg, n = dir == 0 ? 15 - k : k;
    double w = bl[jf].pw[n];
    if (ks >= 1) w *= (double)n;
    if (ks >= 2) w *= (double)n;
    return on ? w : 0.0;
}

template <int NS, int D>        // (D at compile time: the row loops, the 2 D registers that
                                // hold the next tile's rows and their predicates are static --
                                // with a runtime D the kernel spilled 128 scalar registers)
__global__ void __launch_bounds__(256) RL_SF_APPLY_ATTR
k_sf_apply(const double* __restrict__ X, double* __restrict__ Y, int nvec, int m, int NF,
           int nfac, const double* __restrict__ blob, const double* __restrict__ Cin) {
    constexpr int G = RL_SF_G, PAD = RL_SF_PAD, NH = RL_SF_NH, XR = NH * D;
    constexpr int BF = NS == 2 ? 8 : 4;                      // filters per batch
    RL_SMEM(smem);
    const int nslots = D + nfac, nchan = D * NF + nfac, nblob = sf_blob_doubles(NF, nfac, D);
    double* xs = reinterpret_cast<double*>(smem);            // [D][PAD]: x, then the rows of y
    double* us = xs + (size_t)D * PAD;                       // [nfac][PAD]: u_f, then T u_f
    double* tab = us + (size_t)nfac * PAD;                   // the operator's block (see above)
    const double* kap = tab;
    const double* facA = kap + NF * D;
    const double* facAW = facA + nfac * D;
    const double* facJ = facAW + nfac * D;
    const double* tcomb = facJ + nfac;
    const SfBlk* bl = reinterpret_cast<const SfBlk*>(tcomb + nslots * 16);
    double* cinl = tab + nblob;                              // [nchan][2][NS]: the chunk's incoming states
    double* scr = cinl + (size_t)nchan * 2 * NS;             // emulator only
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = nthr >> 6, lg = lane >> 4, col = lane & 15;
    RL_CENSUS_ENTER(120);
    // the operator's block -> LDS, once per workgroup
    for (int e = tid; e < nblob; e += nthr) tab[e] = blob[e];
    sf_lds_barrier();
    double wstx[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            wstx[mt][kk] = sf_state_weight<NS>(
                bl, mt, kk, col, lg, (NS == 2 && (NF & 3) == 1) ? (NF < 4 ? NF : 4) : (NF < BF ? NF : BF), 0,
                true);
    // A workgroup walks tiles (chunk, vector) tile0, tile0 + gridDim.x, ...; the NEXT
    // tile's rows and incoming states are requested into registers before the current
    // tile is worked on, so that the memory round trip hides behind the matrix work.
    // (256 threads: thread tid holds point tid (and tid + 256) of every row)
    const int nch = (m + G - 1) / G, ntiles = nch * nvec, ncin = nchan * 2 * NS;
    double xr[XR], cr[4];
    int tile = blockIdx.x;
    if (tile < ntiles) sf_request<XR>(xr, cr, X, Cin, tile, nch, nvec, D, m, ncin, tid);
    for (; tile < ntiles; tile += gridDim.x) {
    const int chunk = tile % nch, v = tile / nch, g0 = chunk * G;
    RL_STAMP_AT(100, 100, 0);
    // registers -> LDS (the rows padded, see sf_pad), then request the next tile
#pragma unroll
    for (int k = 0; k < XR; ++k)
        if (k / NH < D)
            xs[(size_t)(k / NH) * PAD + sf_pad(tid + 256 * (k % NH))] =
                g0 + tid + 256 * (k % NH) < m ? xr[k] : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (tid + 256 * k < ncin) cinl[tid + 256 * k] = cr[k];
    sf_lds_barrier();
    RL_STAMP_AT(101, 100, 0);
    // mixed rows u_f = sum_b A_f[b] x_b: a thread takes points tid and tid + 256, the D
    // values of a point in registers; the weights are broadcast reads of the block in LDS
    // (scalar loads from global memory measured slower: 4.8 against 2.8 us per tile)
    if (nfac > 0) {
        const double* gA = facA;
#pragma unroll
        for (int half = 0; half < NH; ++half) {
            const int pi = sf_pad(tid + 256 * half);
            // (rows beyond D: unconditional reads of a clamped row, zero weight -- a
            // branch around a read makes the compiler wait for every read in turn,
            // measured 5 us per tile)
            double xb[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) xb[b] = xs[(size_t)(b < D ? b : D - 1) * PAD + pi];
            for (int f = 0; f < nfac; ++f) {
                const double* ar = gA + f * D;
                double u = 0.0;
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const double wgt = ar[b < D ? b : D - 1];
                    u = fma(b < D ? wgt : 0.0, xb[b], u);
                }
                us[(size_t)f * PAD + pi] = u;
            }
        }
        sf_lds_barrier();
    }
    // the next tile's rows: requested now, they arrive while the matrix cores work
    if (tile + (int)gridDim.x < ntiles)
        sf_request<XR>(xr, cr, X, Cin, tile + gridDim.x, nch, nvec, D, m, ncin, tid);
    RL_STAMP_AT(102, 100, 0);
    // row slots, one per wave and pass (idle waves repeat a slot and do not store).  The
    // emulator's cross-lane moves are workgroup barriers, so there the rows of x and the
    // mixed rows (which run different numbers of them) take separate passes.
#if defined(RL_EMU)
    const int xpasses = (D + nwaves - 1) / nwaves, upasses = (nfac + nwaves - 1) / nwaves;
    const int npasses = xpasses + upasses;
#else
    const int npasses = (nslots + nwaves - 1) / nwaves;
#endif
    for (int pass = 0; pass < npasses; ++pass) {
        RL_STAMP_AT(103 + (pass < 6 ? pass : 6), 100, 0);
#if defined(RL_EMU)
        const bool xpass = pass < xpasses;
        const int sraw = xpass ? pass * nwaves + wave : D + (pass - xpasses) * nwaves + wave;
        const int send = xpass ? D : nslots;
#else
        const int sraw = pass * nwaves + wave, send = nslots;
#endif
        const int slot = sraw < send ? sraw : send - 1;
        const bool xrow = slot < D;
        const int jfix = xrow ? 0 : (int)facJ[xrow ? 0 : slot - D];    // the mixed row's filter
        double* row = (xrow ? xs + (size_t)slot * PAD : us + (size_t)(slot - D) * PAD);
        // B fragments of X: point 4 kk + lg of column col (+ 16 per half)
        double XB[NH][4];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) XB[h][kk] = row[(16 * h + col) * 17 + 4 * kk + lg];
        sf_v4d OUT[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) OUT[h] = sf_v4d{0.0, 0.0, 0.0, 0.0};
        // filters of the slot, in batches of BF
        // (NS == 2 and one filter over a multiple of four: batches of four, and the last
        // filter alone in the PACKED layout below)
        const int nf_slot = xrow ? NF : 1;
        const int bf = (NS == 2 && (nf_slot & 3) == 1) ? 4 : BF;
        for (int j0 = 0; j0 < nf_slot; j0 += bf) {
            const int nfb = nf_slot - j0 < bf ? nf_slot - j0 : bf;
            const bool two = nfb > 4 || NS == 3;             // second tile of states in use
            if (NS == 2 && nfb == 1) {
                // ---- ONE filter, packed: its four states (F0, F1, H0, H1) sit on the four
                // lane rows of a single register -- state lg of column col --, so the scan
                // below moves one value per lane (all 64 lanes busy) instead of pairs on a
                // quarter of the lanes, and the response is one matrix instruction per half.
                const int jf = xrow ? j0 : jfix;
                const int pdir = lg >> 1, pks = lg & 1;
                sf_v4d SP[NH];
#pragma unroll
                for (int h = 0; h < NH; ++h) SP[h] = sf_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    // A fragment: state id = col (< 4), point k = 4 kk + lg
                    const int k = 4 * kk + lg, adir = (col >> 1) & 1, aks = col & 1;
                    const int n = adir == 0 ? 15 - k : k;
                    double w = bl[jf].pw[n];
                    if (aks) w *= (double)n;
                    w = col < 4 ? w : 0.0;
#pragma unroll
                    for (int h = 0; h < NH; ++h) sf_mma(w, XB[h][kk], SP[h], scr);
                }
                if (j0 == 0) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int k = 4 * kk + lg, dd = col > k ? col - k : k - col;
                        const double ta = tcomb[slot * 16 + dd];
#pragma unroll
                        for (int h = 0; h < NH; ++h) sf_mma(ta, XB[h][kk], OUT[h], scr);
                    }
                }
                const int chan = xrow ? slot * NF + jf : D * NF + (slot - D);
                const double* p16 = bl[jf].p16;
                const double c0 = cinl[(chan * 2 + pdir) * NS], c1 = cinl[(chan * 2 + pdir) * NS + 1];
                const int steps = pdir == 0 ? col : 15 - col;
                const double rp = p16[steps], np = 16.0 * steps;
                const double kw = xrow ? kap[jf * D + slot] : 1.0;
                const int nr = pdir == 0 ? col + 1 : 16 - col;
                const double wr = kw * (pks == 0 ? bl[jf].tb[nr] : bl[jf].r1[nr]);
                // scalar first-order scans: the second state rides as  q = F1 -/+ 16 c F0
                // (F1(c) = Q(c) + 16 c F0(c) causal, H1(c) = Q(c) - 16 c H0(c) anti-causal)
                double v[NH], x[NH];
                const double cpre = pks ? (pdir == 0 ? -16.0 * col : 16.0 * col) : 0.0;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    v[h] = SP[h][0];
                    v[h] = fma(cpre, sf_row_partner(v[h], scr), v[h]);
                }
#define RL_SF_PSCAN(N_)                                                                      \
    _Pragma("unroll") for (int h = 0; h < NH; ++h) {                                         \
        const double up_ = sf_row_shift<N_, true>(v[h], scr);                                \
        const double dn_ = sf_row_shift<N_, false>(v[h], scr);                               \
        v[h] = fma(p16[N_], pdir == 0 ? up_ : dn_, v[h]);                                    \
    }
                RL_SF_PSCAN(1)
                RL_SF_PSCAN(2)
                RL_SF_PSCAN(4)
                RL_SF_PSCAN(8)
#undef RL_SF_PSCAN
                // exclusive values, back to true states: X1 = q +/- 16 (c -/+ 1) X0
                const double cpost = pks ? (pdir == 0 ? 16.0 * (col - 1) : -16.0 * (col + 1)) : 0.0;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const double up_ = sf_row_shift<1, true>(v[h], scr);
                    const double dn_ = sf_row_shift<1, false>(v[h], scr);
                    x[h] = pdir == 0 ? up_ : dn_;
                    const double edge = pdir == 0 ? (col == 0 ? 0.0 : 1.0) : (col == 15 ? 0.0 : 1.0);
                    x[h] = fma(cpost * edge, sf_row_partner(x[h], scr), x[h]);
                }
                // what enters from outside: the chunk's state (c0, c1), carried `steps` blocks
                // on -- a lane takes its own component of  rho^n (s0, s1 + n s0)
                const double cnear = rp * (pks ? fma(np, c0, c1) : c0);
                if constexpr (NH == 2) {
                    // near half's total (true state) + the chunk's state 256 points on
                    double t = pdir == 0 ? sf_row_bcast<true>(v[0], scr) : sf_row_bcast<false>(v[1], scr);
                    double tp = sf_row_partner(t, scr);
                    if (pdir == 0 && pks) t = fma(240.0, tp, t);        // (column 15: Q + 16 * 15 F0)
                    t = fma(p16[16], pks ? fma(256.0, c0, c1) : c0, t);
                    tp = sf_row_partner(t, scr);
                    const double cfar = rp * (pks ? fma(np, tp, t) : t);
                    x[0] += pdir == 0 ? cnear : cfar;
                    x[1] += pdir == 0 ? cfar : cnear;
                } else {
                    x[0] += cnear;
                }
                // response: A fragment = point col, state id = lg
#pragma unroll
                for (int h = 0; h < NH; ++h) sf_mma(wr, x[h], OUT[h], scr);
                continue;
            }
            // --- S = Wst X.  A fragment: state id = 16 mt + col -> lane row id % 4, register id / 4
            // (rows of x with all their filters in one batch: the weights do not depend on
            // the row and were computed once, before the first tile)
            sf_v4d S[NH][2];
#pragma unroll
            for (int h = 0; h < NH; ++h) S[h][0] = S[h][1] = sf_v4d{0.0, 0.0, 0.0, 0.0};
            double wst[2][4];
            if (xrow && j0 == 0) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) wst[mt][kk] = wstx[mt][kk];
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        wst[mt][kk] = sf_state_weight<NS>(bl, mt, kk, col, lg, nfb, xrow ? j0 : jfix,
                                                          xrow);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                if (mt == 1 && !two) break;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int h = 0; h < NH; ++h) sf_mma(wst[mt][kk], XB[h][kk], S[h][mt], scr);
                }
            }
            if (j0 == 0) {
                // the block's own points: OUT = Tblk X (independent of the states: the matrix
                // cores work on it while the scan below runs on the vector pipe)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = 4 * kk + lg, dd = col > k ? col - k : k - col;
                    const double ta = tcomb[slot * 16 + dd];
#pragma unroll
                    for (int h = 0; h < NH; ++h) sf_mma(ta, XB[h][kk], OUT[h], scr);
                }
            }
            // --- scan of S over the 32 columns: the state each block STARTS from (register
            // R = NS (2 fs + dir) + k of a lane).  This lane's filters: fb = lg (+ 4)
#pragma unroll
            for (int fs = 0; fs < (NS == 2 ? 2 : 1); ++fs) {
                if (fs == 1 && !two) break;
                const int fb = lg + 4 * fs;
                const bool on = fb < nfb;
                const int jf = xrow ? j0 + (on ? fb : 0) : jfix;
                const int chan = xrow ? slot * NF + jf : D * NF + (slot - D);
                const double* p16 = bl[jf].p16;
                const int stepsF = col, stepsB = 15 - col;
                const double r1_ = p16[1], r2_ = p16[2], r4_ = p16[4], r8_ = p16[8], r16_ = p16[16];
                const double rpF = p16[stepsF], rpB = p16[stepsB];
#pragma unroll
                for (int dir = 0; dir < 2; ++dir) {
                    const int R0 = NS * (2 * fs + dir);
                    double V[NH][NS], cin[NS];
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        cin[k] = on ? cinl[(chan * 2 + dir) * NS + k] : 0.0;
#pragma unroll
                        for (int h = 0; h < NH; ++h) V[h][k] = S[h][(R0 + k) >> 2][(R0 + k) & 3];
                    }
                    // (response weights of the group: what a state entering a block contributes at
                    // point col of it -- n = col + 1 steps on for the causal state, 16 - col for
                    // the anti-causal one; requested before the scan)
                    double wr[NS];
                    {
                        const int n = dir == 0 ? col + 1 : 16 - col;
                        const double kw = on ? (xrow ? kap[jf * D + slot] : 1.0) : 0.0;
                        wr[0] = kw * bl[jf].tb[n];
                        wr[1] = kw * bl[jf].r1[n];
                        if constexpr (NS == 3) wr[2] = kw * bl[jf].r2[n];
                    }
                    // inclusive scan inside each half (zeros shift in at the row's end)
#define RL_SF_SCAN_STEP(N_, r_)                                                              \
    {                                                                                        \
        double Sv[NH][NS];                                                                   \
        _Pragma("unroll") for (int h = 0; h < NH; ++h)                                       \
            _Pragma("unroll") for (int k = 0; k < NS; ++k)                                   \
                Sv[h][k] = dir == 0 ? sf_row_shift<N_, true>(V[h][k], scr)                   \
                                    : sf_row_shift<N_, false>(V[h][k], scr);                 \
        _Pragma("unroll") for (int h = 0; h < NH; ++h)                                       \
            sf_carry<NS>(V[h], Sv[h], r_, 16.0 * N_);                                        \
    }
                    RL_SF_SCAN_STEP(1, r1_)
                    RL_SF_SCAN_STEP(2, r2_)
                    RL_SF_SCAN_STEP(4, r4_)
                    RL_SF_SCAN_STEP(8, r8_)
#undef RL_SF_SCAN_STEP
                    // exclusive value; with two halves: the near half's total, carried into the
                    // far half with the chunk's incoming state 256 points on
                    double Xh[NH][NS];
#pragma unroll
                    for (int k = 0; k < NS; ++k)
#pragma unroll
                        for (int h = 0; h < NH; ++h)
                            Xh[h][k] = dir == 0 ? sf_row_shift<1, true>(V[h][k], scr)
                                                : sf_row_shift<1, false>(V[h][k], scr);
                    // (blocks between the half's edge and this one: col / 15 - col)
                    if constexpr (NH == 2) {
                        double T[NS];
#pragma unroll
                        for (int k = 0; k < NS; ++k)
                            // causal: total of half 0 (its lane 15); anti-causal: of half 1 (lane 0)
                            T[k] = dir == 0 ? sf_row_bcast<true>(V[0][k], scr)
                                            : sf_row_bcast<false>(V[NH - 1][k], scr);
                        sf_carry<NS>(T, cin, r16_, 256.0);
                        if (dir == 0) {
                            sf_carry<NS>(Xh[0], cin, rpF, 16.0 * stepsF);
                            sf_carry<NS>(Xh[NH - 1], T, rpF, 16.0 * stepsF);
                        } else {
                            sf_carry<NS>(Xh[NH - 1], cin, rpB, 16.0 * stepsB);
                            sf_carry<NS>(Xh[0], T, rpB, 16.0 * stepsB);
                        }
                    } else {
                        if (dir == 0) sf_carry<NS>(Xh[0], cin, rpF, 16.0 * stepsF);
                        else sf_carry<NS>(Xh[0], cin, rpB, 16.0 * stepsB);
                    }
                    // --- OUT += Rsp C for this group's registers R0 .. R0 + NS - 1, at once:
                    // the matrix cores work on it while the next group's scan runs on the
                    // vector pipe.  A fragment: point col, state id = 4 R + lg
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
#pragma unroll
                        for (int h = 0; h < NH; ++h) sf_mma(wr[k], Xh[h][k], OUT[h], scr);
                    }
                }
            }
        }
        // the slot's result replaces the row: point lg + 4 r of column col
        if (sraw < send) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) row[(16 * h + col) * 17 + lg + 4 * r] = OUT[h][r];
        }
    }
    RL_STAMP_AT(110, 100, 0);
    sf_lds_barrier();
    RL_STAMP_AT(111, 100, 0);
    // y_a = diagonal part + sum_f w_f A_f[a] (T u_f): a thread takes points tid and
    // tid + 256, the D results of a point in registers, all stores of a point issued
    // together
    {
        const double* gAW = facAW;
        double* ybase = Y + (size_t)v * D * m + g0;
#pragma unroll
        for (int half = 0; half < NH; ++half) {
            const int i = tid + 256 * half, pi = sf_pad(i);
            double acc[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) acc[a] = xs[(size_t)(a < D ? a : D - 1) * PAD + pi];
            for (int f = 0; f < nfac; ++f) {
                const double* ar = gAW + f * D;
                const double uw = us[(size_t)f * PAD + pi];
#pragma unroll
                for (int a = 0; a < 16; ++a) acc[a] = fma(ar[a < D ? a : D - 1], uw, acc[a]);
            }
            if (g0 + i < m) {
#pragma unroll
                for (int a = 0; a < 16; ++a)
                    if (a < D) ybase[(size_t)a * m + i] = acc[a];
            }
        }
    }
    RL_STAMP_AT(112, 100, 0);
    sf_lds_barrier();         // (the next tile overwrites the rows)
    }
    RL_CENSUS_LEAVE(120);
}
