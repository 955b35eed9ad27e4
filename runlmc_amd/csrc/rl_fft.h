// In-LDS FFT passes for side-by-side transforms.
//
// A tile holds `cols` independent length-N transforms side by side: element e
// of transform c lives at tile[e * ld + c].  Threads walk (butterfly, column)
// pairs with the column fastest, so a wavefront touches consecutive LDS
// addresses in every pass (no bank conflicts at any stride).
//
// Forward = in-place decimation in frequency: natural order in, digit-scrambled
// order out.  The inverse is the conjugate transpose of the same flow graph
// (passes in reverse order, conjugate twiddle BEFORE the conjugate butterfly),
// so it consumes exactly the scrambled order the forward produces and nothing
// is ever bit-reversed.  tests/flow_model.py is the executable specification.
#pragma once
#include "rl_device.h"

#define RL_SQRT1_2 0.70710678118654752440

template <bool INV>
__device__ __forceinline__ cplx c_quarter(cplx a) {
    return INV ? c_mul_pi(a) : c_mul_mi(a);
}

template <bool INV>
__device__ __forceinline__ void dft2(cplx& a, cplx& b) {
    cplx t = c_sub(a, b);
    a = c_add(a, b);
    b = t;
}

template <bool INV>
__device__ __forceinline__ void dft4(cplx& v0, cplx& v1, cplx& v2, cplx& v3) {
    cplx t0 = c_add(v0, v2), t1 = c_sub(v0, v2);
    cplx t2 = c_add(v1, v3), t3 = c_quarter<INV>(c_sub(v1, v3));
    v0 = c_add(t0, t2);
    v1 = c_add(t1, t3);
    v2 = c_sub(t0, t2);
    v3 = c_sub(t1, t3);
}

// multiply by exp(-/+ i pi/4) and exp(-/+ 3 i pi/4)
template <bool INV>
__device__ __forceinline__ cplx c_eighth(cplx a) {
    return INV ? c_make((a.x - a.y) * RL_SQRT1_2, (a.x + a.y) * RL_SQRT1_2)
               : c_make((a.x + a.y) * RL_SQRT1_2, (a.y - a.x) * RL_SQRT1_2);
}
template <bool INV>
__device__ __forceinline__ cplx c_three_eighths(cplx a) {
    return INV ? c_make((-a.x - a.y) * RL_SQRT1_2, (a.x - a.y) * RL_SQRT1_2)
               : c_make((a.y - a.x) * RL_SQRT1_2, (-a.x - a.y) * RL_SQRT1_2);
}

template <int R, bool INV>
struct SmallDft;

template <bool INV>
struct SmallDft<2, INV> {
    static __device__ __forceinline__ void run(cplx* v) { dft2<INV>(v[0], v[1]); }
};
template <bool INV>
struct SmallDft<4, INV> {
    static __device__ __forceinline__ void run(cplx* v) {
        dft4<INV>(v[0], v[1], v[2], v[3]);
    }
};
template <bool INV>
struct SmallDft<8, INV> {
    static __device__ __forceinline__ void run(cplx* v) {
        cplx e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
        cplx o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
        dft4<INV>(e0, e1, e2, e3);
        dft4<INV>(o0, o1, o2, o3);
        o1 = c_eighth<INV>(o1);
        o2 = c_quarter<INV>(o2);
        o3 = c_three_eighths<INV>(o3);
        v[0] = c_add(e0, o0);
        v[4] = c_sub(e0, o0);
        v[1] = c_add(e1, o1);
        v[5] = c_sub(e1, o1);
        v[2] = c_add(e2, o2);
        v[6] = c_sub(e2, o2);
        v[3] = c_add(e3, o3);
        v[7] = c_sub(e3, o3);
    }
};

// Odd radices: only ever the FIRST pass of a column transform whose length is
// 3 * 2^a or 5 * 2^a (embedding lengths just above 2m instead of the next
// power of two).
#define RL_SIN_PI_3 0.86602540378443864676     // sin(2 pi / 3)
#define RL_COS_2PI_5 0.30901699437494742410
#define RL_COS_4PI_5 -0.80901699437494742410
#define RL_SIN_2PI_5 0.95105651629515357212
#define RL_SIN_4PI_5 0.58778525229247312917
template <bool INV>
struct SmallDft<3, INV> {
    static __device__ __forceinline__ void run(cplx* v) {
        const cplx t = c_add(v[1], v[2]);
        const cplx u = c_scale(c_quarter<INV>(c_sub(v[1], v[2])), RL_SIN_PI_3);
        const cplx mid = c_make(v[0].x - 0.5 * t.x, v[0].y - 0.5 * t.y);
        v[0] = c_add(v[0], t);
        v[1] = c_add(mid, u);
        v[2] = c_sub(mid, u);
    }
};
template <bool INV>
struct SmallDft<5, INV> {
    static __device__ __forceinline__ void run(cplx* v) {
        const cplx t1 = c_add(v[1], v[4]), t2 = c_add(v[2], v[3]);
        const cplx d1 = c_sub(v[1], v[4]), d2 = c_sub(v[2], v[3]);
        const cplx a1 = c_make(v[0].x + RL_COS_2PI_5 * t1.x + RL_COS_4PI_5 * t2.x,
                               v[0].y + RL_COS_2PI_5 * t1.y + RL_COS_4PI_5 * t2.y);
        const cplx a2 = c_make(v[0].x + RL_COS_4PI_5 * t1.x + RL_COS_2PI_5 * t2.x,
                               v[0].y + RL_COS_4PI_5 * t1.y + RL_COS_2PI_5 * t2.y);
        // -i b (forward) / +i b (inverse)
        const cplx b1 = c_quarter<INV>(c_make(RL_SIN_2PI_5 * d1.x + RL_SIN_4PI_5 * d2.x,
                                              RL_SIN_2PI_5 * d1.y + RL_SIN_4PI_5 * d2.y));
        const cplx b2 = c_quarter<INV>(c_make(RL_SIN_4PI_5 * d1.x - RL_SIN_2PI_5 * d2.x,
                                              RL_SIN_4PI_5 * d1.y - RL_SIN_2PI_5 * d2.y));
        v[0] = c_add(v[0], c_add(t1, t2));
        v[1] = c_add(a1, b1);
        v[4] = c_sub(a1, b1);
        v[2] = c_add(a2, b2);
        v[3] = c_sub(a2, b2);
    }
};

// 16 = 4 x 4: a radix-4 stage over elements 4 apart, the W16^{jk} twiddles, a
// radix-4 stage over contiguous quads, outputs renamed to natural order.
#define RL_COS_PI_8 0.92387953251128673848
#define RL_SIN_PI_8 0.38268343236508978178
template <bool INV>
__device__ __forceinline__ cplx c_twc(cplx a, double re, double im_fwd) {
    // a * (re + i*im) with im = im_fwd forward, -im_fwd inverse
    const double im = INV ? -im_fwd : im_fwd;
    return c_make(a.x * re - a.y * im, a.x * im + a.y * re);
}
template <bool INV>
struct SmallDft<16, INV> {
    static __device__ __forceinline__ void run(cplx* v) {
        cplx s[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j][0] = v[j];
            s[j][1] = v[j + 4];
            s[j][2] = v[j + 8];
            s[j][3] = v[j + 12];
            dft4<INV>(s[j][0], s[j][1], s[j][2], s[j][3]);
        }
        // twiddles W16^{j k}, j = butterfly index, k = output of stage 1
        s[1][1] = c_twc<INV>(s[1][1], RL_COS_PI_8, -RL_SIN_PI_8);      // W^1
        s[1][2] = c_eighth<INV>(s[1][2]);                              // W^2
        s[1][3] = c_twc<INV>(s[1][3], RL_SIN_PI_8, -RL_COS_PI_8);      // W^3
        s[2][1] = c_eighth<INV>(s[2][1]);                              // W^2
        s[2][2] = c_quarter<INV>(s[2][2]);                             // W^4
        s[2][3] = c_three_eighths<INV>(s[2][3]);                       // W^6
        s[3][1] = c_twc<INV>(s[3][1], RL_SIN_PI_8, -RL_COS_PI_8);      // W^3
        s[3][2] = c_three_eighths<INV>(s[3][2]);                       // W^6
        s[3][3] = c_twc<INV>(s[3][3], -RL_COS_PI_8, RL_SIN_PI_8);      // W^9
        // stage 2: for each k, DFT4 over j; result (k, k2) is frequency k + 4 k2
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dft4<INV>(s[0][k], s[1][k], s[2][k], s[3][k]);
            v[k] = s[0][k];
            v[k + 4] = s[1][k];
            v[k + 8] = s[2][k];
            v[k + 12] = s[3][k];
        }
    }
};

// One radix-R pass over the whole tile.
//   n    transform length, ns  current sub-transform length (n, n/R1, ...)
//   cols side-by-side transforms, ld  leading dimension (>= cols)
//   tw   table of exp(-2 pi i k / n), k in [0, n)
template <int R, bool INV>
__device__ __forceinline__ void fft_pass(cplx* tile, int n, int ns, int cols, int ld,
                                         const cplx* tw, int tid, int nthr,
                                         unsigned cols_magic = 0) {
    const int sub = ns / R;          // distance between butterfly legs
    const int twstep = n / ns;       // W_ns^j = tw[j * twstep]
    const int work = (n / R) * cols;
    for (int w = tid; w < work; w += nthr) {
        const int bf = cols_magic ? (int)fast_div((unsigned)w, cols_magic) : w / cols;
        const int c = w - bf * cols;
        // sub is a power of two except in a leading odd-radix pass of a length
        // with two odd factors
        const int j = (sub & (sub - 1)) == 0 ? (bf & (sub - 1)) : bf % sub;
        const int g = (bf - j) * R;  // (bf / sub) * ns
        cplx* p = tile + (size_t)(g + j) * ld + c;
        const size_t leg = (size_t)sub * ld;
        cplx v[R];
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = p[i * leg];
        if (INV && sub > 1) {      // sub == 1: every twiddle is W^0
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = c_mulc(v[k], tw[j * k * twstep]);
        }
        SmallDft<R, INV>::run(v);
        if (!INV && sub > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = c_mul(v[k], tw[j * k * twstep]);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) p[i * leg] = v[i];
    }
}

template <bool INV>
__device__ __forceinline__ void fft_pass_any(int radix, cplx* tile, int n, int ns, int cols,
                                             int ld, const cplx* tw, int tid, int nthr,
                                             unsigned cols_magic = 0) {
    if (radix == 16)
        fft_pass<16, INV>(tile, n, ns, cols, ld, tw, tid, nthr, cols_magic);
    else if (radix == 5)
        fft_pass<5, INV>(tile, n, ns, cols, ld, tw, tid, nthr, cols_magic);
    else if (radix == 3)
        fft_pass<3, INV>(tile, n, ns, cols, ld, tw, tid, nthr, cols_magic);
    else if (radix == 8)
        fft_pass<8, INV>(tile, n, ns, cols, ld, tw, tid, nthr, cols_magic);
    else if (radix == 4)
        fft_pass<4, INV>(tile, n, ns, cols, ld, tw, tid, nthr, cols_magic);
    else
        fft_pass<2, INV>(tile, n, ns, cols, ld, tw, tid, nthr, cols_magic);
}

// All forward passes; ends with a barrier.
__device__ __forceinline__ void fft_tile_forward(cplx* tile, const FftPlan& plan, int cols,
                                                 int ld, const cplx* tw, int tid, int nthr) {
    int ns = plan.n;
    for (int s = 0; s < plan.npass; ++s) {
        fft_pass_any<false>(plan.radix[s], tile, plan.n, ns, cols, ld, tw, tid, nthr);
        ns /= plan.radix[s];
        __syncthreads();
    }
}

// All adjoint passes (unnormalised inverse); ends with a barrier.
__device__ __forceinline__ void fft_tile_adjoint(cplx* tile, const FftPlan& plan, int cols,
                                                 int ld, const cplx* tw, int tid, int nthr) {
    int ns = 1;
    for (int s = plan.npass - 1; s >= 0; --s) {
        ns *= plan.radix[s];
        fft_pass_any<true>(plan.radix[s], tile, plan.n, ns, cols, ld, tw, tid, nthr);
        __syncthreads();
    }
}
