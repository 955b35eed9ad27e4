// Third-generation grid product: the WHOLE product of one right-hand side on
// chip, one workgroup per vector -- no intermediates in global memory at all.
// HBM-side traffic: x read twice (the second time out of L2 / Infinity Cache),
// y written, read and written once more; the v2 path moves 4 passes over
// intermediates that are twice the size of x on top of x and y.
//
// Used for 1-D grids whose embedding fits a compute unit (8 * D * EP VGPRs of
// state, EP = frequency-pair slots per thread), for batches large enough to
// fill the chip with one workgroup per vector; rl_kernels2.h stays the path for
// small batches and long grids.
//
// Scheme (tests/flow_model.py: twophase_grid_mvm_model is the executable
// specification).  L = 2N = 4H; the padded sequence has x[n] = 0 for n >= N, so
// the length-L real transform splits by the parity of the frequency into two
// H-point complex transforms that are run one after the other ("phases"), each
// keeping only D * H complex values on chip:
//   E (even k = 2k'): the length-N real transform of x[0:N] = the H-point
//     transform of x[2n] + i x[2n+1], untangled pairwise (c, H - c) into the
//     half spectrum, mixed, re-tangled, transformed back;
//   O (odd k = 4j+1): the H-point transform of (x[n] - i x[n+H]) W_L^n, mixed
//     in place, transformed back, y[n] += Re(conj(W_L^n) g[n]),
//     y[n+H] -= Im(conj(W_L^n) g[n]).  Frequencies 4j+3 are conjugates of
//     mirrored 4j'+1 ones and are never formed.
// The H-point transform is a four-step Ha x Hb split INSIDE LDS: column passes
// (elements ld apart, columns side by side), the inter-step twiddle folded into
// the first row pass, row passes (elements side by side, rows ld apart; ld
// odd: both walks are bank-conflict free).  The half spectra live in REGISTERS
// for all D outputs; the real D x D mix happens there.
// Spectra for this kernel: per top row [H + 1 even, natural order k' = 0..H]
// [H odd, by tile position], made by the same kernel (mode 1).
#pragma once
#include "rl_kernels2.h"

#define RL_THREADS4 512

struct Plan4 {
    int N, Na, Nb, ld;      // N = Na * Nb points of the in-LDS transform (H above), ld = Nb | 1
    FftPlan planA, planB;
    const cplx* twA;        // exp(-2 pi i k / Na)
    const cplx* twB;        // exp(-2 pi i k / Nb)
    const int* freqA;       // row position -> column-transform frequency kA
    TwiddleL twN;           // W_N^e
    const int* pos;         // [N] frequency k -> LDS offset row * ld + col
    const cplx* wl;         // [N + 1]  W_L^i, L = 4 N
    unsigned magicNa, magicNb;
    int nlo, nhi;           // entries of twN.lo / twN.hi
};

// LDS bytes of one k4_product workgroup: the tile and a copy of every table
// the passes read (a dependent global load per pass costs more than the pass)
static inline size_t plan4_lds_bytes(const Plan4& p) {
    return ((size_t)p.Na * p.ld + p.Na + p.Nb + p.nlo + p.nhi + p.N + 1) * sizeof(cplx) +
           (size_t)p.Na * sizeof(int);
}

// the tables the passes read, in LDS (k4_product copies them once per workgroup)
struct Tab4 {
    const cplx* twA;
    const cplx* twB;
    const cplx* wl;
    const int* freqA;
    TwiddleL twN;
};

// One radix-R pass over `ntr` transforms of length n held in LDS with element
// stride `es` and transform stride `ts`; threads walk (butterfly, transform)
// with the transform fastest.  Direction and twiddle placement are RUN-TIME
// flags, so the kernel holds one body per radix.  The adjoint pass is computed in the conjugate
// domain: P^H v = conj(F (Tw conj(v))) with the same forward butterfly F.
//   inter: multiply by the inter-step twiddle W_N^{kA(transform) * element}
//          (forward: on the loaded elements of the first row pass; adjoint:
//          on the stored elements of the last one)
template <int R>
__device__ __forceinline__ void fft_pass4(cplx* tile, bool inv, bool inter, int n, int ns,
                                          int ntr, unsigned tr_magic, int es, int ts,
                                          const cplx* __restrict__ tw, const Tab4& tb, int tid,
                                          int nthr) {
    const int sub = ns / R;
    const int twstep = n / ns;
    const int work = (n / R) * ntr;
    const bool sub_pow2 = (sub & (sub - 1)) == 0;
    const double sgn = inv ? -1.0 : 1.0;
    for (int w = tid; w < work; w += nthr) {
        const int bf = tr_magic ? (int)fast_div((unsigned)w, tr_magic) : w / ntr;
        const int c = w - bf * ntr;
        const int j = sub_pow2 ? (bf & (sub - 1)) : bf % sub;
        const int g = (bf - j) * R;
        cplx* p = tile + (size_t)(g + j) * es + (size_t)c * ts;
        const int leg = sub * es;
        cplx v[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            v[i] = p[i * leg];
            v[i].y *= sgn;
        }
        if (inter && !inv) {
            const int kA = tb.freqA[c];
#pragma unroll
            for (int i = 0; i < R; ++i)
                v[i] = c_mul(v[i], twiddle_L(tb.twN, kA * (g + j + i * sub)));
        }
        if (inv && sub > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = c_mul(v[k], tw[j * k * twstep]);
        }
        SmallDft<R, false>::run(v);
        if (!inv && sub > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = c_mul(v[k], tw[j * k * twstep]);
        }
        if (inter && inv) {
            const int kA = tb.freqA[c];
#pragma unroll
            for (int i = 0; i < R; ++i)
                v[i] = c_mul(v[i], twiddle_L(tb.twN, kA * (g + j + i * sub)));
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            v[i].y *= sgn;
            p[i * leg] = v[i];
        }
    }
}

__device__ __forceinline__ void fft_pass4_any(int radix, cplx* tile, bool inv, bool inter, int n,
                                              int ns, int ntr, unsigned tr_magic, int es, int ts,
                                              const cplx* __restrict__ tw, const Tab4& tb,
                                              int tid, int nthr) {
    // no radix 16 here: its 64 data registers do not fit beside the half spectra
    if (radix == 8)
        fft_pass4<8>(tile, inv, inter, n, ns, ntr, tr_magic, es, ts, tw, tb, tid, nthr);
    else if (radix == 5)
        fft_pass4<5>(tile, inv, inter, n, ns, ntr, tr_magic, es, ts, tw, tb, tid, nthr);
    else if (radix == 4)
        fft_pass4<4>(tile, inv, inter, n, ns, ntr, tr_magic, es, ts, tw, tb, tid, nthr);
    else if (radix == 3)
        fft_pass4<3>(tile, inv, inter, n, ns, ntr, tr_magic, es, ts, tw, tb, tid, nthr);
    else
        fft_pass4<2>(tile, inv, inter, n, ns, ntr, tr_magic, es, ts, tw, tb, tid, nthr);
}

// The H-point transform of the tile, forward (natural order in; position
// row * ld + col holds frequency kA(row) + Na * kB(col) out) or adjoint
// (unnormalised inverse); one barrier after every pass.
// `gp` must be the kernel argument itself: its radix arrays are indexed
// dynamically, which is a scalar load from the argument segment -- a private
// copy would live in scratch memory and cost a memory round trip per pass.
__device__ __forceinline__ void onchip_transform(cplx* tile, bool inv, const Plan4& gp,
                                                 const Tab4& tb, int tid, int nthr) {
    const int na = gp.planA.npass, nb = gp.planB.npass;
    int nsA = inv ? 1 : gp.Na, nsB = inv ? 1 : gp.Nb;
#pragma unroll 1
    for (int t = 0; t < na + nb; ++t) {
        // forward: column passes 0..na-1, then row passes 0..nb-1; adjoint: the reverse
        const int u = inv ? na + nb - 1 - t : t;
        const bool rows = u >= na;
        const int sidx = rows ? u - na : u;
        const int r = rows ? gp.planB.radix[sidx] : gp.planA.radix[sidx];
        int ns;
        if (rows) {
            if (inv) nsB *= r;
            ns = nsB;
            if (!inv) nsB /= r;
        } else {
            if (inv) nsA *= r;
            ns = nsA;
            if (!inv) nsA /= r;
        }
        if (rows)
            fft_pass4_any(r, tile, inv, sidx == 0, gp.Nb, ns, gp.Na, gp.magicNa, 1, gp.ld, tb.twB,
                          tb, tid, nthr);
        else
            fft_pass4_any(r, tile, inv, false, gp.Na, ns, gp.Nb, gp.magicNb, gp.ld, 1, tb.twA, tb,
                          tid, nthr);
        __syncthreads();
    }
}

__device__ __forceinline__ cplx c_conj(cplx a) { return c_make(a.x, -a.y); }

// padded input n of this phase: x[n] for n < m; for a circulant column
// (mode 1) the wrapped half c[n + 2H] = t[2H - n] is folded in with `sign`
__device__ __forceinline__ double load4(const double* __restrict__ x, int n, int m, int N2,
                                        int mode, double sign) {
    double v = 0.0;
    if (n < m) v = x[n];
    if (mode == 1 && n > N2 - m) v += sign * x[N2 - n];
    return v;
}

// ---------------------------------------------------------------------------
// k4_product<D, EP>
//   grid (<= nvec, persistent)   block Tn <= RL_THREADS4, EP * Tn >= H / 2 + 1
//   mode 0: Y[v] = K_UU X[v];  X, Y [nvec][D][m];  mp.spec = the v4 spectra
//   mode 1: X = tops [ntop][m] (D == 1): spec_out[top][2H + 1]
// LDS: tile [Ha][ld] + tables
// ---------------------------------------------------------------------------
template <int D, int EP>
__global__ void __launch_bounds__(RL_THREADS4)
k4_product(const double* __restrict__ X, double* __restrict__ Y, int nvec, Geom geo, int mode_,
           Plan4 gp, MixParams mp, double* __restrict__ spec_out) {
    RL_SMEM(smem);
    cplx* tile = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int mode = mode_;
    const int H = gp.N, Hb = gp.Nb, ld = gp.ld, hh = H >> 1;
    const int N2 = 2 * H;          // x has at most N2 entries
    const int L = 4 * H;
    const int m = geo.m;
    const size_t sps = (size_t)2 * H + 1;      // spectrum doubles per top row
    constexpr int NS = 2 * EP;

    // tables into LDS, once per workgroup
    Tab4 tb;
    {
        cplx* l_twA = tile + (size_t)gp.Na * ld;
        cplx* l_twB = l_twA + gp.Na;
        cplx* l_lo = l_twB + gp.Nb;
        cplx* l_hi = l_lo + gp.nlo;
        cplx* l_wl = l_hi + gp.nhi;
        int* l_fa = reinterpret_cast<int*>(l_wl + H + 1);
        for (int i = tid; i < gp.Na; i += nthr) { l_twA[i] = gp.twA[i]; l_fa[i] = gp.freqA[i]; }
        for (int i = tid; i < gp.Nb; i += nthr) l_twB[i] = gp.twB[i];
        for (int i = tid; i < gp.nlo; i += nthr) l_lo[i] = gp.twN.lo[i];
        for (int i = tid; i < gp.nhi; i += nthr) l_hi[i] = gp.twN.hi[i];
        for (int i = tid; i <= H; i += nthr) l_wl[i] = gp.wl[i];
        tb.twA = l_twA;
        tb.twB = l_twB;
        tb.wl = l_wl;
        tb.freqA = l_fa;
        tb.twN.lo = l_lo;
        tb.twN.hi = l_hi;
        tb.twN.shift = gp.twN.shift;
        tb.twN.mask = gp.twN.mask;
    }
    // phase E pair slots of this thread: frequencies c and H - c (c = 0 pairs
    // with itself and carries k' = 0 and k' = H; so does c = H / 2).
    // phase O slots: tile positions tid + s * nthr, row-major without padding.
    int pa[EP], pb[EP], po[NS];
#pragma unroll
    for (int s = 0; s < EP; ++s) {
        const int c = tid + s * nthr;
        pa[s] = c <= hh ? gp.pos[c] : -1;
        pb[s] = c <= hh ? gp.pos[c == 0 ? 0 : H - c] : -1;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int idx = tid + s * nthr;
        const int n1 = (int)fast_div((unsigned)idx, gp.magicNb);
        po[s] = idx < H ? n1 * ld + (idx - n1 * Hb) : -1;
    }
    __syncthreads();

    cplx S[D][NS];
#pragma unroll
    for (int b = 0; b < D; ++b)
#pragma unroll
        for (int i = 0; i < NS; ++i) S[b][i] = c_make(0.0, 0.0);

    // persistent over vectors (the table copy is paid once), two phases each
#pragma unroll 1
    for (int v = blockIdx.x; v < nvec; v += gridDim.x) {
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        const bool odd = ph == 1;
        // ---- forward: every output's (half) spectrum into registers ------------
#pragma unroll 1
        for (int b = 0; b < D; ++b) {
            const double* x = X + ((size_t)v * D + b) * m;
            for (int idx = tid; idx < H; idx += nthr) {
                const int n1 = (int)fast_div((unsigned)idx, gp.magicNb);
                const int n2 = idx - n1 * Hb;
                cplx z;
                if (!odd) {
                    z = c_make(load4(x, 2 * idx, m, N2, mode, 1.0),
                               load4(x, 2 * idx + 1, m, N2, mode, 1.0));
                } else {
                    z = c_mul(c_make(load4(x, idx, m, N2, mode, -1.0),
                                     -load4(x, idx + H, m, N2, mode, -1.0)), tb.wl[idx]);
                }
                tile[(size_t)n1 * ld + n2] = z;
            }
            __syncthreads();
            onchip_transform(tile, false, gp, tb, tid, nthr);
            cplx tmp[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) tmp[i] = c_make(0.0, 0.0);
            if (!odd) {
#pragma unroll
                for (int s = 0; s < EP; ++s)
                    if (pa[s] >= 0) {
                        const cplx A = tile[pa[s]], Bc = c_conj(tile[pb[s]]);
                        const cplx w = tb.wl[2 * (tid + s * nthr)];            // W_N2^c
                        const cplx P = c_add(A, Bc);
                        const cplx Qd = c_mul_pi(c_mul(w, c_sub(A, Bc)));     // i w (A - conj B)
                        tmp[2 * s] = c_sub(P, Qd);
                        tmp[2 * s + 1] = c_conj(c_add(P, Qd));
                    }
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (po[s] >= 0) tmp[s] = tile[po[s]];
            }
            if (mode == 1) {
                double* so = spec_out + (size_t)v * sps;
                const double scale = (odd ? 2.0 : 0.25) / (double)L;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int c = tid + (s >> 1) * nthr;
                    if (odd) {
                        if (po[s] >= 0) so[H + 1 + tid + s * nthr] = tmp[s].x * scale;
                    } else if (pa[s >> 1] >= 0) {
                        so[(s & 1) ? H - c : c] = tmp[s].x * scale;
                    }
                }
            }
#pragma unroll
            for (int bb = 0; bb < D; ++bb)
                if (bb == b) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) S[bb][i] = tmp[i];
                }
            __syncthreads();       // the next fill overwrites the tile
        }
        if (mode == 1) continue;

        // ---- the real D x D mix at every owned frequency -------------------------
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            // phase E: slot s / 2, element s % 2 (frequency c or H - c); phase O: position s
            const int c = tid + (s >> 1) * nthr;
            const bool live = odd ? po[s] >= 0 : pa[s >> 1] >= 0;
            if (live) {
                const size_t o = odd ? (size_t)(H + 1 + tid + s * nthr)
                                     : (size_t)((s & 1) ? H - c : c);
                cplx z[D];
#pragma unroll
                for (int b = 0; b < D; ++b) z[b] = S[b][s];
                mix_point<D>(z, mp, sps, o);
#pragma unroll
                for (int b = 0; b < D; ++b) S[b][s] = z[b];
            }
        }

        // ---- way back, output by output --------------------------------------------
#pragma unroll 1
        for (int a = 0; a < D; ++a) {
            cplx tmp[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) tmp[i] = c_make(0.0, 0.0);
#pragma unroll
            for (int aa = 0; aa < D; ++aa)
                if (aa == a) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) tmp[i] = S[aa][i];
                }
            if (!odd) {
#pragma unroll
                for (int s = 0; s < EP; ++s)
                    if (pa[s] >= 0) {
                        const cplx U = tmp[2 * s], Vc = c_conj(tmp[2 * s + 1]);
                        const cplx w = tb.wl[2 * (tid + s * nthr)];
                        const cplx P = c_add(U, Vc);
                        const cplx Qd = c_mul_pi(c_mulc(c_sub(U, Vc), w));  // i conj(w) (U - conj V)
                        tile[pa[s]] = c_add(P, Qd);
                        tile[pb[s]] = c_conj(c_sub(P, Qd));
                    }
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (po[s] >= 0) tile[po[s]] = tmp[s];
            }
            __syncthreads();
            onchip_transform(tile, true, gp, tb, tid, nthr);
            // thread n owns y[n] and y[n + H] in both phases: E writes them, O
            // adds Re / -Im of conj(W_L^n) g[n] to them
            double* y = Y + ((size_t)v * D + a) * m;
            for (int n = tid; n < H; n += nthr) {
                if (n >= m) continue;
                if (!odd) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int i = n + half * H;
                        if (i < m) {
                            const int idx = i >> 1;
                            const int n1 = (int)fast_div((unsigned)idx, gp.magicNb);
                            const cplx z = tile[(size_t)n1 * ld + (idx - n1 * Hb)];
                            y[i] = (i & 1) ? z.y : z.x;
                        }
                    }
                } else {
                    const int n1 = (int)fast_div((unsigned)n, gp.magicNb);
                    const cplx h = c_mulc(tile[(size_t)n1 * ld + (n - n1 * Hb)], tb.wl[n]);
                    y[n] += h.x;
                    if (n + H < m) y[n + H] -= h.y;
                }
            }
            __syncthreads();   // the next insert overwrites the tile
        }
    }   // phases
    }   // vectors
}

// ---------------------------------------------------------------------------
// Small-batch variant of the same scheme: TWO kernels, the half spectra of
// every (vector, output, phase) travel through global memory once.
//   k5_forward <EP>    grid (D, nvec, 2): pad, H-point transform, untangle ->
//                      Sg[((v * 2 + ph) * D + b)][slot][thread]
//   k5_inverse <D, EP> grid (D, nvec): for each phase read the D half spectra
//                      at the owned frequencies, mix, keep output a, re-tangle,
//                      transform back, write (E) / add (O) the outputs
// k4_product needs one workgroup per vector and runs its 2 * 2 * D transforms
// one after the other -- fine when there are hundreds of vectors, a long serial
// chain at the 17 of a probe batch.  Here every transform of the batch has its
// own workgroup (2 * D * nvec forward, D * nvec backward), for one kernel
// boundary instead of the two of the three-kernel product.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_tab4(cplx* tile, const Plan4& gp, int ld, int H, Tab4& tb,
                                          int tid, int nthr) {
    cplx* l_twA = tile + (size_t)gp.Na * ld;
    cplx* l_twB = l_twA + gp.Na;
    cplx* l_lo = l_twB + gp.Nb;
    cplx* l_hi = l_lo + gp.nlo;
    cplx* l_wl = l_hi + gp.nhi;
    int* l_fa = reinterpret_cast<int*>(l_wl + H + 1);
    for (int i = tid; i < gp.Na; i += nthr) { l_twA[i] = gp.twA[i]; l_fa[i] = gp.freqA[i]; }
    for (int i = tid; i < gp.Nb; i += nthr) l_twB[i] = gp.twB[i];
    for (int i = tid; i < gp.nlo; i += nthr) l_lo[i] = gp.twN.lo[i];
    for (int i = tid; i < gp.nhi; i += nthr) l_hi[i] = gp.twN.hi[i];
    for (int i = tid; i <= H; i += nthr) l_wl[i] = gp.wl[i];
    tb.twA = l_twA;
    tb.twB = l_twB;
    tb.wl = l_wl;
    tb.freqA = l_fa;
    tb.twN.lo = l_lo;
    tb.twN.hi = l_hi;
    tb.twN.shift = gp.twN.shift;
    tb.twN.mask = gp.twN.mask;
}

template <int EP>
__global__ void __launch_bounds__(RL_THREADS4)
k5_forward(const double* __restrict__ X, int D, Geom geo, Plan4 gp, cplx* __restrict__ Sg) {
    RL_SMEM(smem);
    cplx* tile = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int b = blockIdx.x, v = blockIdx.y;
    const bool odd = blockIdx.z == 1;
    const int H = gp.N, Hb = gp.Nb, ld = gp.ld, hh = H >> 1;
    const int N2 = 2 * H;
    const int m = geo.m;
    constexpr int NS = 2 * EP;
    const double* x = X + ((size_t)v * D + b) * m;
    // the input first: it does not depend on the tables
    for (int idx = tid; idx < H; idx += nthr) {
        const int n1 = (int)fast_div((unsigned)idx, gp.magicNb);
        const int n2 = idx - n1 * Hb;
        cplx z;
        if (!odd) {
            z = c_make(load4(x, 2 * idx, m, N2, 0, 1.0), load4(x, 2 * idx + 1, m, N2, 0, 1.0));
        } else {
            z = c_mul(c_make(load4(x, idx, m, N2, 0, -1.0), -load4(x, idx + H, m, N2, 0, -1.0)),
                      gp.wl[idx]);
        }
        tile[(size_t)n1 * ld + n2] = z;
    }
    Tab4 tb;
    load_tab4(tile, gp, ld, H, tb, tid, nthr);
    __syncthreads();
    onchip_transform(tile, false, gp, tb, tid, nthr);
    cplx* out = Sg + ((size_t)(v * 2 + (odd ? 1 : 0)) * D + b) * NS * nthr;
    if (!odd) {
#pragma unroll
        for (int s = 0; s < EP; ++s) {
            const int c = tid + s * nthr;
            if (c <= hh) {
                const cplx A = tile[gp.pos[c]], Bc = c_conj(tile[gp.pos[c == 0 ? 0 : H - c]]);
                const cplx w = tb.wl[2 * c];            // W_N2^c
                const cplx P = c_add(A, Bc);
                const cplx Qd = c_mul_pi(c_mul(w, c_sub(A, Bc)));     // i w (A - conj B)
                out[(size_t)(2 * s) * nthr + tid] = c_sub(P, Qd);
                out[(size_t)(2 * s + 1) * nthr + tid] = c_conj(c_add(P, Qd));
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int idx = tid + s * nthr;
            if (idx < H) {
                const int n1 = (int)fast_div((unsigned)idx, gp.magicNb);
                out[(size_t)s * nthr + tid] = tile[n1 * ld + (idx - n1 * Hb)];
            }
        }
    }
}

template <int D, int EP>
__global__ void __launch_bounds__(RL_THREADS4)
k5_inverse(const cplx* __restrict__ Sg, double* __restrict__ Y, Geom geo, Plan4 gp,
           MixParams mp) {
    RL_SMEM(smem);
    cplx* tile = reinterpret_cast<cplx*>(smem);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int a = blockIdx.x, v = blockIdx.y;
    const int H = gp.N, Hb = gp.Nb, ld = gp.ld, hh = H >> 1;
    const int m = geo.m;
    const size_t sps = (size_t)2 * H + 1;
    constexpr int NS = 2 * EP;
    Tab4 tb;
    load_tab4(tile, gp, ld, H, tb, tid, nthr);
    double* y = Y + ((size_t)v * D + a) * m;
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        const bool odd = ph == 1;
        const cplx* in = Sg + (size_t)(v * 2 + ph) * D * NS * nthr;
        cplx tmp[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = tid + (s >> 1) * nthr;
            const bool live = odd ? tid + s * nthr < H : c <= hh;
            tmp[s] = c_make(0.0, 0.0);
            if (live) {
                const size_t o = odd ? (size_t)(H + 1 + tid + s * nthr)
                                     : (size_t)((s & 1) ? H - c : c);
                cplx z[D];
#pragma unroll
                for (int b = 0; b < D; ++b) z[b] = in[((size_t)b * NS + s) * nthr + tid];
                mix_point<D>(z, mp, sps, o);
#pragma unroll
                for (int b = 0; b < D; ++b)
                    if (b == a) tmp[s] = z[b];
            }
        }
        __syncthreads();        // tables loaded (first phase) / previous outputs read (second)
        if (!odd) {
#pragma unroll
            for (int s = 0; s < EP; ++s) {
                const int c = tid + s * nthr;
                if (c <= hh) {
                    const cplx U = tmp[2 * s], Vc = c_conj(tmp[2 * s + 1]);
                    const cplx w = tb.wl[2 * c];
                    const cplx P = c_add(U, Vc);
                    const cplx Qd = c_mul_pi(c_mulc(c_sub(U, Vc), w));  // i conj(w) (U - conj V)
                    tile[gp.pos[c]] = c_add(P, Qd);
                    tile[gp.pos[c == 0 ? 0 : H - c]] = c_conj(c_sub(P, Qd));
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int idx = tid + s * nthr;
                if (idx < H) {
                    const int n1 = (int)fast_div((unsigned)idx, gp.magicNb);
                    tile[n1 * ld + (idx - n1 * Hb)] = tmp[s];
                }
            }
        }
        __syncthreads();
        onchip_transform(tile, true, gp, tb, tid, nthr);
        // thread n owns y[n] and y[n + H] in both phases (k4_product)
        for (int n = tid; n < H; n += nthr) {
            if (n >= m) continue;
            if (!odd) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int i = n + half * H;
                    if (i < m) {
                        const int idx = i >> 1;
                        const int n1 = (int)fast_div((unsigned)idx, gp.magicNb);
                        const cplx z = tile[(size_t)n1 * ld + (idx - n1 * Hb)];
                        y[i] = (i & 1) ? z.y : z.x;
                    }
                }
            } else {
                const int n1 = (int)fast_div((unsigned)n, gp.magicNb);
                const cplx h = c_mulc(tile[(size_t)n1 * ld + (n - n1 * Hb)], tb.wl[n]);
                y[n] += h.x;
                if (n + H < m) y[n + H] -= h.y;
            }
        }
    }
}
