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
//   k_sf_carries2  per chunk of RL_SF_G grid points and row: the state the chunk
//   (k_sf_carries  alone leaves at its last point (F) and at its first point (H)
//    for three     -- 2 NS weighted sums per filter, lanes along the grid; two-state
//    states)       filters: sums and differences of a point and its mirror halve the
//                  multiply-adds (round 5);
//   k_sf_scan1     per (vector, channel, direction): the chunks' states chained
//   (k_sf_scan     ( state' = rho^G (F0, F1 + G F0, F2 + 2 G F1 + G^2 F0) + chunk ),
//    above 131 072 which gives every chunk the state it starts from; a segment's chunk
//    points)       states read once and kept in registers for both walks (round 5);
//   k_sf_apply     per (vector, chunk): the D rows of the chunk in LDS, rank-one
//                  factors mixed there (u_f = A_f . x); a row is cut into 16 segments
//                  of 32 points and ONE LANE runs the recurrences of a (row, segment)
//                  -- every filter of the row, both directions -- over its 32 points
//                  in registers: once from zero states (what the segment alone
//                  leaves), these chained over the row's 16 segments with DPP row
//                  shifts, then again from the right states with the outputs;
//                  y_a = sum_q kappa_q[a] T_q x_a + sum_f w_f A_f[a] T_q(f) u_f
//                  assembled in LDS and stored.
//
// Every factor a state is carried by ACROSS segments and chunks is a power rho^n
// that the host computed in long double and rounded once (rho^(32 s) between
// segments, rho^G between chunks); rho itself is multiplied in at most 32 times in a
// row (inside a segment): no product of 1e5 rounded rho's ever forms, and the form
// agrees with a long-double evaluation to a few 1e-15 of |T|_1 |x|_inf (tests; the
// transform kernels: 1e-13).
//
// WHICH tops take this form is decided on the host at set time, from the top row
// itself (rl_gridop.hip: sf_detect): the parameters are fitted from four
// samples and the fit is accepted only if  sum_i |t_i - model_i| <= 2e-14 sum_i |t_i|
// over the WHOLE row -- a bound on ||T - T_model||_1, hence on the product's
// error for every input (no trial vectors involved).
// Reference semantics: runlmc/linalg/bttb.py:144-148, kronecker.py:39-46.
#pragma once
#include "rl_device.h"

#define RL_SF_G 512                        // grid points per chunk: 16 segments of 32 points,
                                           // one per lane of a DPP row in k_sf_apply
#define RL_SF_NH (RL_SF_G / 256)
#define RL_SF_S 32                         // points per segment: one lane's share of a row
#define RL_SF_LPR (RL_SF_G / RL_SF_S)      // segments per row of a chunk (= a DPP row of lanes)
#define RL_SF_PAD (RL_SF_G + RL_SF_LPR)    // doubles per LDS row: one pad per segment
#define RL_SF_NSEG 32                      // segments of the chunk chain in k_sf_scan
#define RL_SF_MAXTOPS 16                   // filter tops per operator at most
#define RL_SF_TOL 2e-14                    // accepted sum|t - model| / sum|t|

// position of grid point i of a chunk inside its padded LDS row: a segment starts
// 33 doubles after its neighbour and a row 528 after the row before, so that the 32
// lanes of a half wave -- two rows' 16 segments, each lane at point i of its own
// segment -- read 32 different banks ((16 a + s + i) mod 32, ds_read_b64)
__device__ __forceinline__ int sf_pad(int i) { return i + (i >> 5); }

struct SfTop {
    double rho;       // decay per grid step
    double c[3];      // t_i = (c0 + c1 i + c2 i^2) rho^i
    double rG;        // rho^G: from chunk to chunk
    double rL;        // rho^(G * seglen): from segment to segment of k_sf_scan's chunk chain
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
    const double* pwp;        // [NF][2 G]: the chunk states' parity weights (k_sf_carries2)
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
// the same step on the pair (F0, G = F1 + F0) when NS == 2:  G' = rho G + F0'  -- two
// instructions instead of three (k_sf_apply; three states: the plain step)
template <int NS>
__device__ __forceinline__ void sf_step_sum(double (&F)[NS], double rho, double x) {
    if constexpr (NS == 2) {
        F[0] = fma(rho, F[0], x);
        F[1] = fma(rho, F[1], F[0]);
    } else {
        sf_step<NS>(F, rho, x);
    }
}
// V += r M(n) S:  a state S carried n grid steps further (r = rho^n) added to V
template <int NS>
__device__ __forceinline__ void sf_carry(double (&V)[NS], const double (&S)[NS], double r, double n) {
    V[0] = fma(r, S[0], V[0]);
    V[1] = fma(r, fma(n, S[0], S[1]), V[1]);
    if constexpr (NS == 3) V[2] = fma(r, fma(n * n, S[0], fma(2.0 * n, S[1], S[2])), V[2]);
}

// ---------------------------------------------------------------------------
// sums of NV values per lane over the 64 lanes of a wave.  Afterwards the EVEN lanes
// of the first lane row (lane < 16) hold NV / 8 of the sums each, out[k] = sum number
// j0 + k (the return value is j0; sf_wave_sums_writer says whether a lane holds any).
// GPU: halving butterfly INSIDE the 16-lane rows with DPP moves -- at distance 8
// (row_ror:8) a lane keeps one half of its values and receives that half from its
// partner, at 4 (row_half_mirror: lane l <-> 7 - l, which also differ in bit 2) a
// quarter, at 2 (quad_perm) an eighth; the last NV / 8 values go through plain
// exchanges at distance 1 (DPP) and across the four lane rows (two shuffles).
// (Measured at C5 Matern, k_sf_carries per product: every level a shuffle through the
// LDS crossbar, starting at distance 32, 0.29 ms; this 0.26 ms; the NV x 64 values
// through the wave's own LDS and added there by 64 lanes, 0.27 ms.  The kernel runs at
// the fp64 vector rate: 164 multiply-adds and 126 reduction instructions per filter and
// block of four rows.)  Emulator: LDS.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool sf_wave_sums_writer(int lane) { return lane < 16 && (lane & 1) == 0; }
#if !defined(RL_EMU)
template <int CTRL>
__device__ __forceinline__ double sf_dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#endif
template <int NV>
__device__ __forceinline__ int sf_wave_sums(const double (&acc)[NV], double* red, double (&out)[NV / 8]) {
    constexpr int A = NV / 2, B = NV / 4, C = NV / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool h3 = (lane & 8) != 0, h2 = (lane & 4) != 0, h1 = (lane & 2) != 0;
    const int j0 = (h3 ? A : 0) + (h2 ? B : 0) + (h1 ? C : 0);
#if defined(RL_EMU)
    const int wave = tid >> 6;
    for (int j = 0; j < NV; ++j) {
        red[tid] = acc[j];
        __syncthreads();
        if (sf_wave_sums_writer(lane) && j >= j0 && j < j0 + C) {
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
        const double mine = h3 ? acc[A + k] : acc[k], send = h3 ? acc[k] : acc[A + k];
        a[k] = mine + sf_dpp_move<0x128>(send);                 // row_ror:8 = lane ^ 8
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
        const double mine = h2 ? a[B + k] : a[k], send = h2 ? a[k] : a[B + k];
        b[k] = mine + sf_dpp_move<0x141>(send);                 // row_half_mirror: l <-> 7 - l
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const double mine = h1 ? b[C + k] : b[k], send = h1 ? b[k] : b[C + k];
        c[k] = mine + sf_dpp_move<0x4E>(send);                  // quad_perm [2, 3, 0, 1] = lane ^ 2
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
        c[k] += sf_dpp_move<0xB1>(c[k]);                        // quad_perm [1, 0, 3, 2] = lane ^ 1
        c[k] += __shfl_xor(c[k], 16, 64);
        c[k] += __shfl_xor(c[k], 32, 64);
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
// two neighbouring doubles at an 8-byte-aligned address, moved as ONE 16-byte access
struct __attribute__((packed, aligned(8))) SfPair {
    double a, b;
};
// first point of the pair thread tid requests (clamped into the row: a pair never
// reaches past it; the staging picks the right half, sf_stage)
__device__ __forceinline__ int sf_pair_base(int g0, int tid, int m) {
    const int gi = g0 + 2 * tid;
    return gi <= m - 2 ? gi : m - 2;            // (the host admits grids of 64 points and more)
}
// RB rows' points of a chunk for one wave: lane l takes the pairs 2 l + 128 k, k < NK / 2
// (one 16-byte load each, clamped into the row like sf_pair_base)
template <int NK, int RB>
__device__ __forceinline__ void sf_request_rows(double (&xv)[NK][RB], const double* __restrict__ X,
                                                int nrows, int m, int r0, int g0, int lane) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = r0 + r;
        const double* xr = X + (size_t)(row < nrows ? row : nrows - 1) * m;
#pragma unroll
        for (int k = 0; k < NK / 2; ++k) {
            const SfPair pr = *reinterpret_cast<const SfPair*>(xr + sf_pair_base(g0 + 128 * k, lane, m));
            xv[2 * k][r] = pr.a;
            xv[2 * k + 1][r] = pr.b;
        }
    }
}

// ---------------------------------------------------------------------------
// k_sf_carries2: the same chunk states for TWO-state filters with the point / mirror-point
// parity trick (round 5).  The causal and the anti-causal state of a chunk carry mirrored
// weights -- dir 1: sum_t t^k rho^t x_t, dir 0: sum_t t^k rho^t x_{G-1-t} --, so with
//   s_t = x_t + x_{G-1-t},  d_t = x_t - x_{G-1-t}   (t < G / 2; shared by every filter)
// and host weights  a = (p_t + p_m) / 2, b = (p_t - p_m) / 2  (p_t = rho^t, m = G-1-t;
// c, e: the same of t rho^t)
//   dir 1 = sum a s + sum b d,   dir 0 = sum a s - sum b d:
// 4 multiply-adds per point PAIR, state order and filter instead of 8, no weight
// arithmetic on the device.  A lane holds the pairs (2 l, 2 l + 1) + 128 j of the first half
// (j < 2) next to their mirrors, the pairs 510 - 2 l - 128 j (both 16-byte loads, one
// ascending, one descending); the wave sums (SA, SB) pairs land in one writer lane.
//   grid / block as k_sf_carries;  pwp: [NF][G / 2][4] = (a, b, c, e)
//   LDS: NF (G / 2) 4 doubles  (+ 256 for the emulator's sums)
// ---------------------------------------------------------------------------
// a pair at chunk offset `off` (even) of a row, zeros past the end of the grid; `pr` was
// requested from sf_pair_clamp(g0 + off, m)
__device__ __forceinline__ int sf_pair_clamp(int gi, int m) { return gi <= m - 2 ? gi : m - 2; }
__device__ __forceinline__ void sf_pair_pick(const SfPair& pr, int gi, int m, bool rowl, double& a,
                                             double& b) {
    // (a pair that would reach past the row was requested one point further left)
    a = rowl && gi < m ? (gi > m - 2 ? pr.b : pr.a) : 0.0;
    b = rowl && gi + 1 < m ? pr.b : 0.0;
}
template <int RB>
__device__ __forceinline__ void sf_request_rows2(SfPair (&xa)[2][RB], SfPair (&xm)[2][RB],
                                                 const double* __restrict__ X, int nrows, int m,
                                                 int r0, int g0, int lane) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = r0 + r;
        const double* xr = X + (size_t)(row < nrows ? row : nrows - 1) * m;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            xa[j][r] = *reinterpret_cast<const SfPair*>(xr + sf_pair_clamp(g0 + 2 * lane + 128 * j, m));
            xm[j][r] = *reinterpret_cast<const SfPair*>(
                xr + sf_pair_clamp(g0 + RL_SF_G - 2 - 2 * lane - 128 * j, m));
        }
    }
}
static __global__ void __launch_bounds__(256)
k_sf_carries2(const double* __restrict__ X, int nrows, int m, int NF, const double* __restrict__ pwp,
              int rows_per_wg, double* __restrict__ E) {
    constexpr int RB = 4, NS = 2, NV = RB * 2 * NS, G = RL_SF_G, HG = G / 2;
    RL_SMEM(smem);
    double* wl = reinterpret_cast<double*>(smem);                // [NF][HG][4]
    double* red = wl + (size_t)NF * HG * 4;                      // emulator only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, rbase = blockIdx.y * rows_per_wg, g0 = chunk * G;
    // The weights sit in LDS TRANSPOSED, [NF][p][component][lane]: a lane needs (a, b, c, e) of
    // its four points t = 2 l + (p & 1) + 128 (p >> 1), and read from the host's [t][4] order --
    // 32 bytes per lane at a stride of 64 -- lanes l and l + 2 met in the same banks: 58 % of the
    // kernel's LDS cycles were conflicts (round-5 counters, SQ_LDS_BANK_CONFLICT 3.2e7 of 5.4e7).
    // Now every read is 64 consecutive doubles.
    for (int e = tid; e < NF * HG * 4; e += 256) {
        const int q = e / (HG * 4), rem = e - q * HG * 4;
        const int t = rem >> 2, c = rem & 3;
        const int l = (t & 127) >> 1, p = (t & 1) + 2 * (t >> 7);
        wl[((q * 4 + p) * 4 + c) * 64 + l] = pwp[e];
    }
    __syncthreads();
    SfPair na[2][RB], nm[2][RB];
    sf_request_rows2<RB>(na, nm, X, nrows, m, rbase + wave * RB, g0, lane);
    for (int r0 = rbase + wave * RB; r0 < rbase + rows_per_wg; r0 += 4 * RB) {
        // s and d of the lane's four points t = 2 l + b + 128 j (j, b < 2) of every row
        double sv[4][RB], dv[4][RB];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool rowl = r0 + r < nrows;
                double a0, a1, m0, m1;
                sf_pair_pick(na[j][r], g0 + 2 * lane + 128 * j, m, rowl, a0, a1);
                sf_pair_pick(nm[j][r], g0 + G - 2 - 2 * lane - 128 * j, m, rowl, m0, m1);
                // point t = 2 l + 128 j mirrors G-1-t = (G - 2 - 2 l - 128 j) + 1: the pair's SECOND
                sv[2 * j][r] = a0 + m1;
                dv[2 * j][r] = a0 - m1;
                sv[2 * j + 1][r] = a1 + m0;
                dv[2 * j + 1][r] = a1 - m0;
            }
        if (r0 + 4 * RB < rbase + rows_per_wg)
            sf_request_rows2<RB>(na, nm, X, nrows, m, r0 + 4 * RB, g0, lane);
        for (int q = 0; q < NF; ++q) {
            const double* w = wl + (size_t)q * HG * 4;
            // acc[r][k][0 / 1] = (SA_k, SB_k) of row r: neighbours, so that a writer lane of the
            // wave sums holds a pair
            double acc[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) acc[j] = 0.0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const double* wp = w + p * 256 + lane;
                const double wa = wp[0], wb = wp[64], wc = wp[128], we = wp[192];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    acc[r * 4 + 0] = fma(wa, sv[p][r], acc[r * 4 + 0]);
                    acc[r * 4 + 1] = fma(wb, dv[p][r], acc[r * 4 + 1]);
                    acc[r * 4 + 2] = fma(wc, sv[p][r], acc[r * 4 + 2]);
                    acc[r * 4 + 3] = fma(we, dv[p][r], acc[r * 4 + 3]);
                }
            }
            double out[NV / 8];
            const int j0 = sf_wave_sums<NV>(acc, red, out);
            if (sf_wave_sums_writer(lane)) {
                // out = (SA_k, SB_k) of row r = j0 / 4, state order k = (j0 / 2) & 1
                const int r = j0 >> 2, k = (j0 >> 1) & 1;
                if (r0 + r < nrows) {
                    double* dst = E + (((size_t)chunk * nrows + r0 + r) * NF + q) * 2 * NS;
                    dst[k] = out[0] - out[1];            // dir 0: causal, at the chunk's last point
                    dst[NS + k] = out[0] + out[1];       // dir 1: anti-causal, at its first point
                }
            }
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
            for (int k = 0; k < NK / 2; ++k) {
                // (a pair that would reach past the row was requested one point further left)
                const int gi = g0 + 2 * lane + 128 * k;
                const bool rowl = r0 + r < nrows;
                xv[2 * k][r] = rowl && gi < m ? (gi > m - 2 ? xn[2 * k + 1][r] : xn[2 * k][r]) : 0.0;
                xv[2 * k + 1][r] = rowl && gi + 1 < m ? xn[2 * k + 1][r] : 0.0;
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
                const int t = 2 * lane + (k & 1) + 128 * (k >> 1);      // (the point xv[k] holds)
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
            if (sf_wave_sums_writer(lane)) {
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

// k_sf_scan1<NS>: the same, the chunk states of a thread's segment read ONCE and kept in
// registers for both walks (round 5; grids of at most RL_SF_SEGMAX * RL_SF_NSEG chunks = 131 072
// points, C5: 196 chunks, 7 per segment).  A mixed channel's chunk state is a sum over D rows of
// E: read twice, these were half of k_sf_scan's traffic (472 MB per C5 Matern product).
#define RL_SF_SEGMAX 8
template <int NS>
__global__ void __launch_bounds__(256)
k_sf_scan1(const double* __restrict__ E, int nchunks, int nvec, int D, SfParams sp,
           double* __restrict__ Cin, int* __restrict__ next_tile) {
    RL_SMEM(smem);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *next_tile = 0;
    constexpr int NCD = 8, NSEG = RL_SF_NSEG, SM = RL_SF_SEGMAX;
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
    const int seglen = (nchunks + NSEG - 1) / NSEG;          // <= SM (host)
    const int p0 = seg * seglen < nchunks ? seg * seglen : nchunks;
    const int p1 = p0 + seglen < nchunks ? p0 + seglen : nchunks;
    // the segment's chunk states, requested together (clamped chunks: zero weight)
    double es[SM][NS];
#pragma unroll
    for (int k = 0; k < SM; ++k) {
        const int p = p0 + k < p1 ? p0 + k : (p1 > 0 ? p1 - 1 : 0);
        const int c = dir == 0 ? p : nchunks - 1 - p;
        sf_chunk_state<NS>(E, sp, c < nchunks ? c : nchunks - 1, nrows, row0, D, diag, a, f, j, dir, es[k]);
    }
    double st[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) st[k] = 0.0;
#pragma unroll
    for (int k = 0; k < SM; ++k) {
        if (p0 + k < p1) {
            double e[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) e[q] = es[k][q];
            sf_carry<NS>(e, st, rG, n);
#pragma unroll
            for (int q = 0; q < NS; ++q) st[q] = e[q];
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) agg[(seg * NCD + cdl) * NS + k] = st[k];
    __syncthreads();
    const double rL = sp.tops[j].rL;
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
#pragma unroll
    for (int k = 0; k < SM; ++k) {
        if (p0 + k < p1) {
            const int p = p0 + k;
            const int c = dir == 0 ? p : nchunks - 1 - p;
            if (live) {
                double* dst = Cin + ((((size_t)c * nvec + v) * nchan + chan) * 2 + dir) * NS;
#pragma unroll
                for (int q = 0; q < NS; ++q) dst[q] = st[q];
            }
            double e[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) e[q] = es[k][q];
            sf_carry<NS>(e, st, rG, n);
#pragma unroll
            for (int q = 0; q < NS; ++q) st[q] = e[q];
        }
    }
}

template <int NS>
__global__ void __launch_bounds__(256)
k_sf_scan(const double* __restrict__ E, int nchunks, int nvec, int D, SfParams sp,
          double* __restrict__ Cin, int* __restrict__ next_tile) {
    RL_SMEM(smem);
    // (k_sf_apply, launched behind this kernel, deals its tiles through this counter)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *next_tile = 0;
    constexpr int NCD = 8, NSEG = RL_SF_NSEG;
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
    // (a power the host computed in long double and rounded once, like every factor a
    // state is carried by across chunks: seglen grows with the grid)
    const double rL = sp.tops[j].rL;
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
// Cross-lane primitives of k_sf_apply.  GPU: DPP row shifts; emulator (one fiber
// per lane, no cross-lane hardware): the same data movement through an LDS
// scratch of 128 doubles per wave.
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

// ---------------------------------------------------------------------------
// k_sf_apply<NS, D>: Y[v] = filter part of the operator applied to X[v].
//   grid (resident workgroups: each walks tiles (chunk, vector))   block 256
//   LDS: (D + nfac) RL_SF_PAD doubles of rows + the operator's block + the chunk's
//        incoming states (+ 128 doubles per wave for the emulator's cross-lane moves)
//
// A chunk row of 512 points is cut into 16 SEGMENTS of 32 points, and ONE LANE runs
// the recurrences of a (row, segment) -- every filter of the row, both directions --
// over its 32 points held in registers.  The 16 segments of a row sit on the 16
// lanes of a DPP row:
//
//   A   from zero states: the states each segment alone leaves at its last point
//       (causal) and at its first point (anti-causal);
//   B   those states chained over the 16 segments with DPP row shifts (Kogge-Stone,
//       powers rho^(32 N) from the host) + the chunk's incoming states carried
//       32 s points on: the state every segment STARTS from;
//   C   the recurrences again from these states, now with their outputs:
//       y_i = sum_j kappa_j[a] (c0 (F0 + H0 - x) + c1 (F1 + H1) + c2 (F2 + H2));
//       one or two filters: both directions in one loop, the 32 results in registers;
//       five: the causal pass overwrites the row in LDS, the anti-causal one adds to it.
//
// At NS = 2 a lane keeps the pair (F0, G = F1 + F0) instead of (F0, F1):
//   F0' = rho F0 + x,  G' = rho G + F0'   -- two multiply-adds per step, and
//   c0 F0 + c1 F1 = (c0 - c1) F0 + c1 G   -- two more for the output,
// so a point costs 2 (A) + 4 (C) vector instructions per filter and direction; no
// cross-lane traffic but the 16-lane chain in between.  The kernel is bound by the
// fp64 vector rate (one wave instruction per 2.3 ns and SIMD, tools/valu_rate.hip:
// 2430 instructions per wave of 64 (row, segment) lanes with five filters), three of
// the four SIMDs loaded (10 rows x 16 segments = 2.5 waves).
// (Up to the middle of round 3 the blocks were 16 points and their maps ran on the fp64
// matrix cores, the chain over 32 blocks per row: 430 matrix instructions of 64 cycles
// per tile and about as many vector cycles again for the chains -- 0.97 ms per C5
// Matern product against 0.90 ms for this kernel, which is a third of the code.)
// Row slots: the D rows of x (every filter, weight kappa_j[a]) and the nfac mixed
// rows u_f = A_f . x (filter facJ[f], weight 1); y_a = the row's result +
// sum_f w_f A_f[a] (T u_f), assembled from LDS and stored.
// ---------------------------------------------------------------------------
struct SfBlk {                // per filter, staged in LDS
    double rho;               // decay per grid step
    double c[3];              // t_i = (c0 + c1 i + c2 i^2) rho^i
    double p32[17];           // rho^(32 s)
};
#define RL_SF_BLKD ((int)(sizeof(SfBlk) / sizeof(double)))
// Everything k_sf_apply needs about the operator, one block of doubles built at
// set time and copied to LDS by every workgroup:
//   kappa [NF][D] | facA [nfac][D] | facAW [nfac][D] | facJ [nfac] | SfBlk [NF]
__host__ __device__ inline int sf_blob_doubles(int NF, int nfac, int D) {
    return NF * D + 2 * nfac * D + nfac + NF * RL_SF_BLKD;
}

#if defined(RL_EMU)
#define RL_SF_APPLY_ATTR
#else
// two workgroups per CU = two waves per SIMD: 256 registers (a segment's 32 points,
// the states of five filters and the next tile's rows among them)
#define RL_SF_APPLY_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif

// phase stamps (timing builds): thread 0 of each wave of workgroup 100 at its 21st tile,
// slots k + 30 wave (round-3 timing build; the tool is in the history of tools/)
#define RL_SF_STAMP(k) RL_STAMP_IF((k) + 30 * (threadIdx.x >> 6), blockIdx.x == 100 && (threadIdx.x & 63) == 0 && sf_iter == 20)

// requests a tile's D rows -- thread tid: the points 2 tid and 2 tid + 1 of every row, ONE
// 16-byte load per row (a wave stalls at its 17th vector-memory instruction in flight until
// the first has returned: with two 8-byte loads per row the request of D = 10 rows took
// 1.8 us, with half of them 0.4, measured) -- and its incoming states into registers.
// Loads go to clamped addresses and points past the grid are zeroed when the registers
// go to LDS, not here: a select right behind a load makes the compiler wait for it.
template <int XR>
__device__ __forceinline__ void sf_request(double (&xr)[XR], double (&cr)[4],
                                           const double* __restrict__ X,
                                           const double* __restrict__ Cin, int tile, int nch,
                                           int nvec, int D, int m, int ncin, int tid) {
    const int chunk = tile % nch, v = tile / nch, g0 = chunk * RL_SF_G;
    const double* xbase = X + (size_t)v * D * m + sf_pair_base(g0, tid, m);
#pragma unroll
    for (int k = 0; k < XR / 2; ++k) {
        if (k < D) {
            const SfPair pr = *reinterpret_cast<const SfPair*>(xbase + (size_t)k * m);
            xr[2 * k] = pr.a;
            xr[2 * k + 1] = pr.b;
        }
    }
    const double* src = Cin + ((size_t)chunk * nvec + v) * ncin;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (256 * k < ncin) cr[k] = src[tid + 256 * k < ncin ? tid + 256 * k : ncin - 1];
}

// The unrolled recurrences are long straight-line code; left alone the scheduler moves
// loads and stores across all 32 steps and the allocator then spills hundreds of
// registers.  A fence every few steps keeps what is live to what the steps need.
#if defined(RL_EMU)
#define RL_SF_FENCE(i_)
#else
#define RL_SF_FENCE(i_) if (((i_) & 3) == 3) __builtin_amdgcn_sched_barrier(0)
#endif

// One lane's (row, segment): BF filters j0 .. j0 + BF - 1 of the row over the 32
// points xv; the result overwrites (FIRST) or adds to cell[0 .. 31].
//   kw[f * kstride]: weight of filter j0 + f (nullptr: 1);  cin0: the chunk's incoming
//   states of the row's channels j0 ..., [BF][2][NS];  s: the segment (= lane & 15)
template <int NS, int BF, bool FIRST>
__device__ __forceinline__ void sf_task(double* __restrict__ cell, const double (&xv)[RL_SF_S],
                                        bool live, int s, const SfBlk* __restrict__ bl,
                                        int j0, const double* __restrict__ kw, int kstride,
                                        const double* __restrict__ cin0, double* scr) {
    constexpr int S = RL_SF_S;
    double rho[BF], cw[BF][NS], F[BF][NS], H[BF][NS], c0sum = 0.0;
#pragma unroll
    for (int f = 0; f < BF; ++f) {
        rho[f] = bl[j0 + f].rho;
        const double k = kw ? kw[f * kstride] : 1.0;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            cw[f][q] = k * bl[j0 + f].c[q];
            F[f][q] = 0.0;
            H[f][q] = 0.0;
        }
        if constexpr (NS == 2) {
            c0sum += cw[f][0];
            cw[f][0] -= cw[f][1];       // c0 F0 + c1 F1 = (c0 - c1) F0 + c1 G
        }
    }
    // A: what the segment alone leaves -- the causal states over ascending points and the
    // anti-causal ones over descending points in the same loop: 2 BF independent chains
    // (a dependent fp64 instruction issues every 6 ns, an independent one every 2.3:
    // measured, tools/valu_rate.hip)
#pragma unroll
    for (int i = 0; i < S; ++i) {
#pragma unroll
        for (int f = 0; f < BF; ++f) {
            sf_step_sum<NS>(F[f], rho[f], xv[i]);
            sf_step_sum<NS>(H[f], rho[f], xv[S - 1 - i]);
        }
        RL_SF_FENCE(i);
    }
    // B: chained over the row's 16 segments -> the state the segment starts from
#pragma unroll
    for (int f = 0; f < BF; ++f) {
        const double* p32 = bl[j0 + f].p32;
        const double r1_ = p32[1], r2_ = p32[2], r4_ = p32[4], r8_ = p32[8];
        const double rpF = p32[s], rpB = p32[15 - s];
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            double V[NS], cin[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                V[q] = dir == 0 ? F[f][q] : H[f][q];
                cin[q] = cin0[(f * 2 + dir) * NS + q];
            }
            if constexpr (NS == 2) V[1] -= V[0];             // (F0, G) -> (F0, F1)
#define RL_SF_SCAN_STEP(N_, r_)                                                              \
    {                                                                                        \
        double Sv[NS];                                                                       \
        _Pragma("unroll") for (int q = 0; q < NS; ++q)                                       \
            Sv[q] = dir == 0 ? sf_row_shift<N_, true>(V[q], scr)                             \
                             : sf_row_shift<N_, false>(V[q], scr);                           \
        sf_carry<NS>(V, Sv, r_, (double)(S * N_));                                           \
    }
            RL_SF_SCAN_STEP(1, r1_)
            RL_SF_SCAN_STEP(2, r2_)
            RL_SF_SCAN_STEP(4, r4_)
            RL_SF_SCAN_STEP(8, r8_)
#undef RL_SF_SCAN_STEP
            double Xh[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q)
                Xh[q] = dir == 0 ? sf_row_shift<1, true>(V[q], scr) : sf_row_shift<1, false>(V[q], scr);
            if (dir == 0) sf_carry<NS>(Xh, cin, rpF, (double)(S * s));
            else sf_carry<NS>(Xh, cin, rpB, (double)(S * (15 - s)));
            if constexpr (NS == 2) Xh[1] += Xh[0];           // back to (F0, G)
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                if (dir == 0) F[f][q] = Xh[q];
                else H[f][q] = Xh[q];
            }
        }
    }
    if constexpr (BF <= 2) {
        // C, few filters: the causal pass (the point itself included) over ascending points
        // and the anti-causal one (excluded) over descending points, interleaved like A; the
        // results of the 32 points collect in registers
        double yv[S];
#pragma unroll
        for (int i = 0; i < S; ++i) yv[i] = FIRST ? 0.0 : cell[i];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int j = S - 1 - i;
            double ya = 0.0, yb = 0.0;
            if constexpr (NS == 2) yb = -c0sum * xv[j];
#pragma unroll
            for (int f = 0; f < BF; ++f) {
                sf_step_sum<NS>(F[f], rho[f], xv[i]);
#pragma unroll
                for (int q = 0; q < NS; ++q) ya = fma(cw[f][q], F[f][q], ya);
                if constexpr (NS == 2) {
                    // the states AFTER the step hold the point itself: H0' - x and G' - H0' are
                    // the excluded ones:  c0 (H0' - x) + c1 (G' - H0') = cw0 H0' + cw1 G' - c0 x
                    sf_step_sum<2>(H[f], rho[f], xv[j]);
                    yb = fma(cw[f][0], H[f][0], yb);
                    yb = fma(cw[f][1], H[f][1], yb);
                } else {
                    const double tt = sf_step<NS>(H[f], rho[f], xv[j]);
                    yb = fma(cw[f][0], tt, yb);
#pragma unroll
                    for (int q = 1; q < NS; ++q) yb = fma(cw[f][q], H[f][q], yb);
                }
            }
            yv[i] += ya;
            yv[j] += yb;
        }
#pragma unroll
        for (int i = 0; i < S; ++i)
            if (live) cell[i] = yv[i];
    } else {
        // C, five filters (no registers for 32 results): the causal pass writes the row in
        // LDS, the anti-causal one adds to it; two partial sums per point, so that the
        // ten accumulations of a step do not form one dependent chain
#pragma unroll
        for (int i = 0; i < S; ++i) {
            double ya = FIRST ? 0.0 : cell[i], yb = 0.0;
#pragma unroll
            for (int f = 0; f < BF; ++f) {
                sf_step_sum<NS>(F[f], rho[f], xv[i]);
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    if (f & 1) yb = fma(cw[f][q], F[f][q], yb);
                    else ya = fma(cw[f][q], F[f][q], ya);
                }
            }
            if (live) cell[i] = ya + yb;
            RL_SF_FENCE(i);
        }
#pragma unroll
        for (int i = S - 1; i >= 0; --i) {
            double ya = cell[i], yb = 0.0;
            
            ya = fma(-c0sum, xv[i], ya);
#pragma unroll
            for (int f = 0; f < BF; ++f) {
                sf_step_sum<2>(H[f], rho[f], xv[i]);
                if (f & 1) {
                    yb = fma(cw[f][0], H[f][0], yb);
                    yb = fma(cw[f][1], H[f][1], yb);
                } else {
                    ya = fma(cw[f][0], H[f][0], ya);
                    ya = fma(cw[f][1], H[f][1], ya);
                }
            }
            if (live) cell[i] = ya + yb;
            RL_SF_FENCE(S - 1 - i);
        }
    }
}

// a tile's rows and incoming states from the registers they were requested into to LDS
// (the rows padded, see sf_pad; points past the grid zeroed), then the mixed rows
// u_f = sum_b A_f[b] x_b: a thread takes its two points (2 tid, 2 tid + 1), the D values of a
// point in registers; the weights are broadcast reads of the block in LDS (scalar loads
// from global memory measured slower: 4.8 against 2.8 us per tile).  A thread touches
// ITS OWN columns only -- the ones it also assembles y from --, so no barrier is needed
// between the assembly of one tile, this, and the mixed rows: one at the end.
template <int D, int XR>
__device__ __forceinline__ void sf_stage(const double (&xr)[XR], const double (&cr)[4], double* xs,
                                         double* us, double* cinl, const double* facA, int nfac,
                                         int ncin, int g0, int m, int tid, int sf_iter) {
    constexpr int PAD = RL_SF_PAD, NH = RL_SF_NH;
    static_assert(NH == 2, "a thread owns two neighbouring points of a chunk row");
    {
        // (the pair was requested from sf_pair_base: one point further left when the row
        // ends at the pair's first point)
        const int gi = g0 + 2 * tid, pi = sf_pad(2 * tid);
        const bool shifted = gi > m - 2;
#pragma unroll
        for (int k = 0; k < XR / 2; ++k)
            if (k < D) {
                const double p0 = shifted ? xr[2 * k + 1] : xr[2 * k];
                xs[(size_t)k * PAD + pi] = gi < m ? p0 : 0.0;
                xs[(size_t)k * PAD + pi + 1] = gi + 1 < m ? xr[2 * k + 1] : 0.0;
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (tid + 256 * k < ncin) cinl[tid + 256 * k] = cr[k];
    RL_SF_STAMP(101);
    if (nfac > 0) {
#pragma unroll
        for (int half = 0; half < NH; ++half) {
            const int pi = sf_pad(2 * tid) + half;
            // (D is a template argument: exactly D reads and multiply-adds.  With a runtime D
            // the loops ran to 16 with clamped reads and zero weights -- a branch around a read
            // made the compiler wait for every read in turn, 5 us per tile -- and until round 5
            // they stayed that way: 16 instead of 10 at C5)
            double xb[D];
#pragma unroll
            for (int b = 0; b < D; ++b) xb[b] = xs[(size_t)b * PAD + pi];
            for (int f = 0; f < nfac; ++f) {
                const double* ar = facA + f * D;
                double u = 0.0;
#pragma unroll
                for (int b = 0; b < D; ++b) u = fma(ar[b], xb[b], u);
                us[(size_t)f * PAD + pi] = u;
            }
        }
    }
    sf_lds_barrier();
    RL_SF_STAMP(102);
}

template <int NS, int D>        // (D at compile time: the row loops, the 2 D registers that
                                // hold the next tile's rows and their predicates are static --
                                // with a runtime D the kernel spilled 128 scalar registers)
static __global__ void __launch_bounds__(256) RL_SF_APPLY_ATTR
k_sf_apply(const double* __restrict__ X, double* __restrict__ Y, int nvec, int m, int NF,
           int nfac, const double* __restrict__ blob, const double* __restrict__ Cin,
           int* __restrict__ next_tile) {
    constexpr int G = RL_SF_G, PAD = RL_SF_PAD, NH = RL_SF_NH, XR = NH * D, S = RL_SF_S;
    RL_SMEM(smem);
    const int nchan = D * NF + nfac, nblob = sf_blob_doubles(NF, nfac, D);
    double* xs = reinterpret_cast<double*>(smem);            // [D][PAD]: x, then the rows of y
    double* us = xs + (size_t)D * PAD;                       // [nfac][PAD]: u_f, then T u_f
    double* tab = us + (size_t)nfac * PAD;                   // the operator's block (see above)
    const double* kap = tab;
    const double* facA = kap + NF * D;
    const double* facAW = facA + nfac * D;
    const double* facJ = facAW + nfac * D;
    const SfBlk* bl = reinterpret_cast<const SfBlk*>(facJ + nfac);
    double* cinl = tab + nblob;                              // [nchan][2][NS]: the chunk's incoming states
    int* nxt = reinterpret_cast<int*>(cinl + (size_t)nchan * 2 * NS);   // [2]: the tiles to come
    double* scr = cinl + (size_t)nchan * 2 * NS + 1;         // emulator only
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = nthr >> 6;
    RL_CENSUS_ENTER(120);
#if defined(RL_TIMING) && !defined(RL_EMU)
    const long long sf_t0 = wall_clock64();
#endif
    // the operator's block -> LDS, once per workgroup
    for (int e = tid; e < nblob; e += nthr) tab[e] = blob[e];
    sf_lds_barrier();
    // A workgroup starts with tile blockIdx.x and draws every further tile (chunk, vector)
    // from a counter: workgroups do not all run at the same speed (two share a CU's
    // SIMDs; lifetimes of 0.51 .. 1.01 ms were measured at C5 with every workgroup on its
    // fixed 49 or 50 tiles), and a tile's result does not depend on who computes it.
    // Thread 0 draws the tile after next while the current one is assembled; the number
    // crosses to the other threads through LDS and the barrier that ends the staging.
    // The NEXT tile's rows and incoming states are requested into registers once the
    // recurrences over the rows of x are done (these need the registers themselves:
    // requested earlier, the rows were spilled one by one, every spill waiting for its
    // load); they arrive during the mixed rows, the assembly and the other resident
    // workgroup's recurrences.  A request and its use sit in ONE loop iteration.
    // (256 threads: thread tid holds the points 2 tid and 2 tid + 1 of every row)
    const int nch = (m + G - 1) / G, ntiles = nch * nvec, ncin = nchan * 2 * NS;
    int tile = blockIdx.x, sf_iter = -1;
    if (tid == 0) nxt[0] = (int)gridDim.x + atomicAdd(next_tile, 1);
    if (tile < ntiles) {
        double xr[XR], cr[4];
        sf_request<XR>(xr, cr, X, Cin, tile, nch, nvec, D, m, ncin, tid);
        RL_SF_STAMP(100);
        sf_stage<D, XR>(xr, cr, xs, us, cinl, facA, nfac, ncin, (tile % nch) * G, m, tid, sf_iter);
    }
    int slot = 0;
    while (tile < ntiles) {
    const int chunk = tile % nch, v = tile / nch, g0 = chunk * G;
    ++sf_iter;
    RL_SF_STAMP(99);
    // (row, segment) tasks, one per lane: the rows of x with all their filters (in batches
    // of five (NS = 2 only: three states of five filters in both directions do not fit the
    // registers), two or one), then the mixed rows with their one filter.  A wave without
    // a live task skips the pass (the emulator's cross-lane moves are workgroup barriers:
    // there every wave runs every pass).
    for (int t0 = 0; t0 < D * 16; t0 += nthr) {
#if !defined(RL_EMU)
        if (t0 + wave * 64 >= D * 16) break;
#endif
        const int t = t0 + tid, araw = t >> 4, s = t & 15;
#if defined(RL_EMU)
        const bool live = araw < D;             // (idle lanes walk a clamped row and do not store)
#else
        if (araw >= D) continue;                // (whole DPP rows of lanes: the chains stay intact)
        constexpr bool live = true;
#endif
        const int a = live ? araw : D - 1;
        double* cell = xs + (size_t)a * PAD + s * (S + 1);
        double xv[S];
#pragma unroll
        for (int i = 0; i < S; ++i) xv[i] = cell[i];
        for (int j = 0; j < NF;) {
            const int b = (NS == 2 && NF - j >= 5) ? 5 : (NF - j >= 2 ? 2 : 1);
            const double* kw = kap + j * D + a;
            const double* cin0 = cinl + (size_t)(a * NF + j) * 2 * NS;
            if (j == 0) {
                if (NS == 2 && b == 5) sf_task<2, 5, true>(cell, xv, live, s, bl, j, kw, D, cin0, scr);
                else if (b == 2) sf_task<NS, 2, true>(cell, xv, live, s, bl, j, kw, D, cin0, scr);
                else sf_task<NS, 1, true>(cell, xv, live, s, bl, j, kw, D, cin0, scr);
            } else {
                if (NS == 2 && b == 5) sf_task<2, 5, false>(cell, xv, live, s, bl, j, kw, D, cin0, scr);
                else if (b == 2) sf_task<NS, 2, false>(cell, xv, live, s, bl, j, kw, D, cin0, scr);
                else sf_task<NS, 1, false>(cell, xv, live, s, bl, j, kw, D, cin0, scr);
            }
            j += b;
        }
    }
    RL_SF_STAMP(103);
    const int next = nxt[slot];
    double xr[XR], cr[4];
#if defined(RL_EMU)
    for (int k = 0; k < XR; ++k) xr[k] = 0.0;      // (g++ cannot see that a request always precedes a use)
    for (int k = 0; k < 4; ++k) cr[k] = 0.0;
#endif
    if (next < ntiles) sf_request<XR>(xr, cr, X, Cin, next, nch, nvec, D, m, ncin, tid);
    RL_SF_STAMP(104);
    // (the waves the rows of x left idle share the mixed rows among them, in rounds;
    // every wave when none was idle)
    const int xw = (D * 16 + 63) / 64, uw0 = xw < nwaves ? xw : 0, nuw = nwaves - uw0;
    for (int t0 = 0; t0 < nfac * 16; t0 += nuw * 64) {
        const int t = t0 + (wave - uw0) * 64 + lane, fraw = t >> 4, s = t & 15;
#if defined(RL_EMU)
        const bool live = wave >= uw0 && fraw < nfac;
#else
        if (wave < uw0 || fraw >= nfac) continue;
        constexpr bool live = true;
#endif
        const int f = live ? fraw : 0;
        double* cell = us + (size_t)f * PAD + s * (S + 1);
        double xv[S];
#pragma unroll
        for (int i = 0; i < S; ++i) xv[i] = cell[i];
        sf_task<NS, 1, true>(cell, xv, live, s, bl, (int)facJ[f], nullptr, 0,
                             cinl + (size_t)(D * NF + f) * 2 * NS, scr);
    }
    RL_SF_STAMP(110);
    sf_lds_barrier();
    RL_SF_STAMP(111);
    // (drawn now, handed to LDS after the assembly: the round trip hides behind it)
    slot ^= 1;
    int drawn = 0;
    if (tid == 0 && next < ntiles) drawn = (int)gridDim.x + atomicAdd(next_tile, 1);
    // y_a = diagonal part + sum_f w_f A_f[a] (T u_f): a thread takes its two points, the
    // D results of each in registers, and stores a row's pair with ONE 16-byte store
    {
        const double* gAW = facAW;
        const int i0 = 2 * tid, pi = sf_pad(i0);
        double* ybase = Y + (size_t)v * D * m + g0 + i0;
        double acc[NH][D];
#pragma unroll
        for (int half = 0; half < NH; ++half) {
#pragma unroll
            for (int a = 0; a < D; ++a) acc[half][a] = xs[(size_t)a * PAD + pi + half];
            for (int f = 0; f < nfac; ++f) {
                const double* ar = gAW + f * D;
                const double uw = us[(size_t)f * PAD + pi + half];
#pragma unroll
                for (int a = 0; a < D; ++a) acc[half][a] = fma(ar[a], uw, acc[half][a]);
            }
        }
        if (g0 + i0 + 1 < m) {
#pragma unroll
            for (int a = 0; a < D; ++a)
                *reinterpret_cast<SfPair*>(ybase + (size_t)a * m) = SfPair{acc[0][a], acc[1][a]};
        } else if (g0 + i0 < m) {
#pragma unroll
            for (int a = 0; a < D; ++a) ybase[(size_t)a * m] = acc[0][a];
        }
    }
    RL_SF_STAMP(112);
    if (tid == 0) nxt[slot] = drawn;
    tile = next;
    if (tile < ntiles) {
        RL_SF_STAMP(100);
        sf_stage<D, XR>(xr, cr, xs, us, cinl, facA, nfac, ncin, (tile % nch) * G, m, tid, sf_iter);
    }
    }
    RL_CENSUS_LEAVE(120);
#if defined(RL_TIMING) && !defined(RL_EMU)
    if (threadIdx.x == 0) {
        // a workgroup's life: longest, sum, count, shortest (as 2^40 - duration)
        const unsigned long long d = (unsigned long long)(wall_clock64() - sf_t0);
        atomicMax((unsigned long long*)&rl_timing_buf[210], d);
        atomicAdd((unsigned long long*)&rl_timing_buf[211], d);
        atomicAdd((unsigned long long*)&rl_timing_buf[212], 1ull);
        atomicMax((unsigned long long*)&rl_timing_buf[213], (1ull << 40) - d);
    }
#endif
}
