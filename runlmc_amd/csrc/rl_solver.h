// Batched Krylov solves on the device: MINRES (Paige & Saunders) and CG for
// many right-hand sides at once, one operator.
//
// Replaces Iterative.solve (reference runlmc/approx/iterative.py:23-62) and
// the N+1 independent pool-mapped solves of StochasticDerivService
// (runlmc/lmc/stochastic_deriv.py:39-52).  The reference delegates the
// iteration itself to scipy.sparse.linalg.minres / cg; the recurrences and
// stopping tests below follow those (SciPy 1.15.3 _isolve/minres.py,
// _isolve/iterative.py), restated in oracle/solver.py, so that iterates,
// iteration counts and exit reasons can be compared one to one.
//
// Every right-hand side keeps its own scalars in device memory; nothing
// returns to the host inside an iteration.  A right-hand side that has met a
// stopping rule is frozen (its x stops changing) while the others continue.
// Dot products are two-stage and deterministic: each workgroup writes one
// partial per (rhs, block), and the NEXT kernel sums the partials of its rhs
// in a fixed order.
#pragma once
#include "rl_device.h"

#define RL_SOLVER_THREADS 256

// per-rhs scalar state (doubles)
enum {
    S_BETA1 = 0, S_OLDB, S_BETA, S_DBAR, S_EPSLN, S_PHIBAR, S_RHS1, S_RHS2, S_TNORM2,
    S_GMAX, S_GMIN, S_CS, S_SN, S_ALFA, S_PHI, S_OLDEPS, S_DELTA, S_DENOM, S_ROOT,
    S_GBAR, S_RESID, S_BNORM, S_RHO, S_RHO_PREV, S_CG_ATOL,
    S_NFIELDS
};
// per-rhs integer state
enum { I_ISTOP = 0, I_ITN, I_ACTIVE, I_NFIELDS };

// exit reasons beyond SciPy's istop codes
#define RL_ISTOP_RESIDUAL 10   // reference rule: ||b - A x|| < tol at a check
#define RL_ISTOP_ZERO_RHS 11
#define RL_ISTOP_DIRECT_STALL 12   // rl_solve_direct: tolerance not reached within max_refine refinements

// Deterministic block sums.  On the GPU: shuffles inside each wavefront (no
// barrier), one partial per wavefront through LDS, every thread adds the few
// partials in order.  (The emulator build has no cross-lane operations: plain
// LDS tree there.)  `red` needs blockDim.x doubles (2 blockDim.x for the pair).
#if defined(RL_EMU)
__device__ __forceinline__ double block_reduce_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ void block_reduce_sum2(double& a, double& b, double* red) {
    a = block_reduce_sum(a, red);
    b = block_reduce_sum(b, red);
}
#else
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    return v;       // valid in lane 0
}
__device__ __forceinline__ double block_reduce_sum(double v, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < nw; ++i) r += red[i];
    __syncthreads();
    return r;
}
__device__ __forceinline__ void block_reduce_sum2(double& a, double& b, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const double sa = wave_sum(a), sb = wave_sum(b);
    if (lane == 0) {
        red[wave] = sa;
        red[nw + wave] = sb;
    }
    __syncthreads();
    double ra = 0.0, rb = 0.0;
    for (int i = 0; i < nw; ++i) {
        ra += red[i];
        rb += red[nw + i];
    }
    __syncthreads();
    a = ra;
    b = rb;
}
#endif

// sums of two partial arrays of one system, every thread gets both: one load
// per thread and a block reduction instead of nblk dependent adds per thread
// (nblk <= blockDim.x).  Every workgroup of a system computes bit-identical
// sums (same data, same order).
__device__ __forceinline__ void sum2_partials(const double* pa, const double* pb, int nblk,
                                              double* red, double* sa, double* sb) {
    double a = 0.0, b = 0.0;
    if ((int)threadIdx.x < nblk) {
        if (pa != nullptr) a = pa[threadIdx.x];
        if (pb != nullptr) b = pb[threadIdx.x];
    }
    block_reduce_sum2(a, b, red);
    *sa = a;
    *sb = b;
}

// the same for ANY number of partials (strided per thread, then the block reduction:
// a fixed order)
__device__ __forceinline__ void sum2_partials_long(const double* pa, const double* pb, int na,
                                                   int nb, double* red, double* sa, double* sb) {
    // (eight values of each array requested before the first is added -- same order of the
    // additions, a value past the end adds 0.0: with one partial per expansion workgroup a
    // thread has sixteen of each, and summed load by load the two heads of a round took 14 us each)
    double a = 0.0, b = 0.0;
    constexpr int U = 8;
    if (pa == nullptr) na = 0;
    if (pb == nullptr) nb = 0;
    const int nmax = na > nb ? na : nb;
    const double* qa = na > 0 ? pa : pb;        // (any readable address for the clamped loads)
    const double* qb = nb > 0 ? pb : pa;
    for (int k0 = threadIdx.x; k0 < nmax; k0 += U * blockDim.x) {
        double va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * (int)blockDim.x;
            va[u] = qa[k < na ? k : 0];
            vb[u] = qb[k < nb ? k : 0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * (int)blockDim.x;
            a += k < na ? va[u] : 0.0;
            b += k < nb ? vb[u] : 0.0;
        }
    }
    block_reduce_sum2(a, b, red);
    *sa = a;
    *sb = b;
}

// sqrt(a^2 + b^2) without the libm call (scaled: no spurious overflow)
__device__ __forceinline__ double hypot2(double a, double b) {
    a = fabs(a);
    b = fabs(b);
    const double t = a > b ? a : b;
    if (t == 0.0) return 0.0;
    const double u = a / t, v = b / t;
    return t * sqrt(fma(u, u, v * v));
}

__device__ __forceinline__ double sum_partials(const double* p, int nblk) {
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += p[i];
    return s;
}

__device__ __forceinline__ void block_range(int n, int* lo, int* hi) {
    const int per = (n + gridDim.x - 1) / gridDim.x;
    *lo = blockIdx.x * per;
    *hi = *lo + per < n ? *lo + per : n;
}

// ---- shared by both methods -------------------------------------------------
// partial[rhs][blk] = sum_i a[i] * b[i]
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_dot_partial(const double* __restrict__ a, const double* __restrict__ b, int n,
              double* __restrict__ partial) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y;
    int lo, hi;
    block_range(n, &lo, &hi);
    const double* pa = a + (size_t)rhs * n;
    const double* pb = b + (size_t)rhs * n;
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) acc = fma(pa[i], pb[i], acc);
    acc = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) partial[(size_t)rhs * gridDim.x + blockIdx.x] = acc;
}

// partial[rhs][blk] = sum_i (b[i] - ax[i])^2
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_resid_partial(const double* __restrict__ b, const double* __restrict__ ax, int n,
                double* __restrict__ partial) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y;
    int lo, hi;
    block_range(n, &lo, &hi);
    const double* pb = b + (size_t)rhs * n;
    const double* pa = ax + (size_t)rhs * n;
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const double r = pb[i] - pa[i];
        acc = fma(r, r, acc);
    }
    acc = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) partial[(size_t)rhs * gridDim.x + blockIdx.x] = acc;
}

// one thread per rhs: residual norm from partials; optionally freeze
static __global__ void k_resid_finish(const double* __restrict__ partial, int nblk, int nrhs,
                               double* __restrict__ resid, int* __restrict__ I, double tol,
                               int freeze) {
    const int rhs = blockIdx.x * blockDim.x + threadIdx.x;
    if (rhs >= nrhs) return;
    const double r = sqrt(sum_partials(partial + (size_t)rhs * nblk, nblk));
    resid[rhs] = r;
    if (freeze && I[rhs * I_NFIELDS + I_ACTIVE] && r < tol) {
        I[rhs * I_NFIELDS + I_ACTIVE] = 0;
        I[rhs * I_NFIELDS + I_ISTOP] = RL_ISTOP_RESIDUAL;
    }
}

// *count = number of still-active right-hand sides
static __global__ void k_count_active(const int* __restrict__ I, int nrhs, int* __restrict__ count) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int c = 0;
        for (int r = 0; r < nrhs; ++r) c += I[r * I_NFIELDS + I_ACTIVE] ? 1 : 0;
        *count = c;
    }
}

// ---- MINRES -----------------------------------------------------------------
// Buffers of one batched MINRES run.  Which of the rotating buffers plays which
// role is derived ON THE DEVICE from the global iteration counter, so every
// iteration is launched with identical arguments and a captured hipGraph of a
// few iterations can be replayed any number of times:
//   it = *giter (iterations completed);  p3 = it % 3;  p2 = it & 1
//   r1 = tri[p3], r2 = tri[p3+1], new y -> tri[p3+2]      (indices mod 3)
//   w_{k-2} = w[p2] (overwritten by the new w), w_{k-1} = w[1-p2]
//   scalar state: read S[p2], write S[1-p2]
struct MinresBufs {
    double* tri[3];
    double* w[2];
    double* v;      // Lanczos vector, input of the operator product
    double* q;      // A v, output of the operator product
    double* x;
    double* S[2];
    int* I;
    int* giter;
    double* lanczos;    // [nrhs][lanczos_cap][2] (alfa_k, beta_{k+1}) or NULL
    int lanczos_cap;
};

// init: x = 0, r1 = r2 = b, w = 0, v = b / beta1; partial = b.b comes from
// k_dot_partial(b, b).
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres_init(const double* __restrict__ b, int n, const double* __restrict__ partial,
              MinresBufs mb) {
    const int rhs = blockIdx.y;
    const int nblk = gridDim.x;
    const double bb = sum_partials(partial + (size_t)rhs * nblk, nblk);
    const double beta1 = sqrt(bb);
    int lo, hi;
    block_range(n, &lo, &hi);
    const size_t off = (size_t)rhs * n;
    const double s = beta1 > 0.0 ? 1.0 / beta1 : 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const double bi = b[off + i];
        mb.x[off + i] = 0.0;
        mb.tri[0][off + i] = bi;
        mb.tri[1][off + i] = bi;
        mb.w[0][off + i] = 0.0;
        mb.w[1][off + i] = 0.0;
        mb.v[off + i] = s * bi;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int c = 0; c < 2; ++c) {
            double* st = mb.S[c] + (size_t)rhs * S_NFIELDS;
            for (int f = 0; f < S_NFIELDS; ++f) st[f] = 0.0;
            st[S_BETA1] = beta1;
            st[S_BETA] = beta1;
            st[S_PHIBAR] = beta1;
            st[S_RHS1] = beta1;
            st[S_GMIN] = 1.7976931348623157e308;
            st[S_CS] = -1.0;
            st[S_BNORM] = beta1;
        }
        int* it = mb.I + rhs * I_NFIELDS;
        it[I_ITN] = 0;
        it[I_ISTOP] = beta1 > 0.0 ? 0 : RL_ISTOP_ZERO_RHS;
        it[I_ACTIVE] = beta1 > 0.0 ? 1 : 0;
        if (rhs == 0) *mb.giter = 0;
    }
}

// step A (q = A v already computed): y = q - (beta/oldb) r1 (itn >= 2; y = q
// on the first iteration); partialA = v . y
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres_a(MinresBufs mb, int n, double* __restrict__ partialA) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y;
    if (!mb.I[rhs * I_NFIELDS + I_ACTIVE]) return;
    const int it = *mb.giter;
    const int p3 = it % 3;
    const double* st = mb.S[it & 1] + (size_t)rhs * S_NFIELDS;
    const double* r1 = mb.tri[p3];
    double* y = mb.tri[(p3 + 2) % 3];
    const int itn = mb.I[rhs * I_NFIELDS + I_ITN] + 1;
    const double coef = itn >= 2 ? st[S_BETA] / st[S_OLDB] : 0.0;
    int lo, hi;
    block_range(n, &lo, &hi);
    const size_t off = (size_t)rhs * n;
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        double yi = mb.q[off + i];
        if (itn >= 2) yi = yi - coef * r1[off + i];
        y[off + i] = yi;
        acc = fma(mb.v[off + i], yi, acc);
    }
    acc = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) partialA[(size_t)rhs * gridDim.x + blockIdx.x] = acc;
}

// step B: alfa = sum partialA; y -= (alfa/beta) r2; partialB = y . y
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres_b(MinresBufs mb, int n, const double* __restrict__ partialA,
           double* __restrict__ partialB) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y;
    if (!mb.I[rhs * I_NFIELDS + I_ACTIVE]) return;
    const int it = *mb.giter;
    const int p3 = it % 3;
    const double* r2 = mb.tri[(p3 + 1) % 3];
    double* y = mb.tri[(p3 + 2) % 3];
    const int nblk = gridDim.x;
    const double alfa = sum_partials(partialA + (size_t)rhs * nblk, nblk);
    const double coef = alfa / mb.S[it & 1][(size_t)rhs * S_NFIELDS + S_BETA];
    int lo, hi;
    block_range(n, &lo, &hi);
    const size_t off = (size_t)rhs * n;
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const double yi = y[off + i] - coef * r2[off + i];
        y[off + i] = yi;
        acc = fma(yi, yi, acc);
    }
    acc = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) partialB[(size_t)rhs * nblk + blockIdx.x] = acc;
}

// step C: scalar recurrences (every block recomputes them from the current
// state copy, block 0 publishes to the other copy), new w over w_{k-2},
// x += phi w, partialC = x . x, v <- y / beta_new.
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres_c(MinresBufs mb, int n, const double* __restrict__ partialA,
           const double* __restrict__ partialB, double* __restrict__ partialC) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y;
    const int nblk = gridDim.x;
    const int it = *mb.giter;
    const int p3 = it % 3, p2 = it & 1;
    const double* si = mb.S[p2] + (size_t)rhs * S_NFIELDS;
    double* so = mb.S[1 - p2] + (size_t)rhs * S_NFIELDS;
    if (!mb.I[rhs * I_NFIELDS + I_ACTIVE]) {
        // keep the two copies identical for frozen systems
        if (blockIdx.x == 0 && threadIdx.x < S_NFIELDS) so[threadIdx.x] = si[threadIdx.x];
        return;
    }
    const double* y = mb.tri[(p3 + 2) % 3];
    double* w1 = mb.w[p2];
    const double* w2 = mb.w[1 - p2];
    const double eps = 2.220446049250313e-16;
    const double alfa = sum_partials(partialA + (size_t)rhs * nblk, nblk);
    double beta = sum_partials(partialB + (size_t)rhs * nblk, nblk);
    beta = sqrt(beta > 0.0 ? beta : 0.0);
    const double oldb = si[S_BETA];
    const double tnorm2 = si[S_TNORM2] + alfa * alfa + oldb * oldb + beta * beta;
    const double cs0 = si[S_CS], sn0 = si[S_SN], dbar0 = si[S_DBAR];
    const double oldeps = si[S_EPSLN];
    const double delta = cs0 * dbar0 + sn0 * alfa;
    const double gbar = sn0 * dbar0 - cs0 * alfa;
    const double epsln = sn0 * beta;
    const double dbar = -cs0 * beta;
    const double root = hypot(gbar, dbar);
    double gamma = hypot(gbar, beta);
    gamma = gamma > eps ? gamma : eps;
    const double cs = gbar / gamma;
    const double sn = beta / gamma;
    const double phi = cs * si[S_PHIBAR];
    const double phibar = sn * si[S_PHIBAR];
    const double denom = 1.0 / gamma;

    int lo, hi;
    block_range(n, &lo, &hi);
    const size_t off = (size_t)rhs * n;
    const double sinv = beta > 0.0 ? 1.0 / beta : 0.0;
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const double wn = (mb.v[off + i] - oldeps * w1[off + i] - delta * w2[off + i]) * denom;
        w1[off + i] = wn;
        const double xi = mb.x[off + i] + phi * wn;
        mb.x[off + i] = xi;
        acc = fma(xi, xi, acc);
        mb.v[off + i] = sinv * y[off + i];
    }
    acc = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) partialC[(size_t)rhs * nblk + blockIdx.x] = acc;

    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int f = 0; f < S_NFIELDS; ++f) so[f] = si[f];
        // Lanczos tridiagonal entries of this system (stochastic Lanczos
        // quadrature of log det reuses them; nothing else reads them)
        const int itn = mb.I[rhs * I_NFIELDS + I_ITN];      // completed so far
        if (mb.lanczos != nullptr && itn < mb.lanczos_cap) {
            double* lz = mb.lanczos + ((size_t)rhs * mb.lanczos_cap + itn) * 2;
            lz[0] = alfa;
            lz[1] = beta;
        }
        so[S_OLDB] = oldb;
        so[S_BETA] = beta;
        so[S_TNORM2] = tnorm2;
        so[S_DBAR] = dbar;
        so[S_EPSLN] = epsln;
        so[S_CS] = cs;
        so[S_SN] = sn;
        so[S_PHIBAR] = phibar;
        so[S_PHI] = phi;
        so[S_ALFA] = alfa;
        so[S_OLDEPS] = oldeps;
        so[S_DELTA] = delta;
        so[S_DENOM] = denom;
        so[S_ROOT] = root;
        so[S_GBAR] = gbar;
        const double gmax = si[S_GMAX] > gamma ? si[S_GMAX] : gamma;
        const double gmin = si[S_GMIN] < gamma ? si[S_GMIN] : gamma;
        so[S_GMAX] = gmax;
        so[S_GMIN] = gmin;
        const double z = si[S_RHS1] / gamma;
        so[S_RHS1] = si[S_RHS2] - delta * z;
        so[S_RHS2] = -epsln * z;
    }
}

// step D (ONE workgroup): ynorm from partialC, SciPy's stopping tests for every
// system, then the global iteration counter advances (after every thread of
// this single workgroup has read it).
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres_test(MinresBufs mb, const double* __restrict__ partialC, int nblk, int nrhs,
              double rtol, int maxiter) {
    const int it0 = *mb.giter;
    const double* S = mb.S[1 - (it0 & 1)];      // the copy step C just wrote
    for (int rhs = threadIdx.x; rhs < nrhs; rhs += blockDim.x) {
        int* it = mb.I + rhs * I_NFIELDS;
        if (!it[I_ACTIVE]) continue;
        const double eps = 2.220446049250313e-16;
        const double* st = S + (size_t)rhs * S_NFIELDS;
        const int itn = it[I_ITN] + 1;
        it[I_ITN] = itn;
        int istop = 0;
        const double beta1 = st[S_BETA1];
        if (itn == 1 && st[S_BETA] / beta1 <= 10 * eps) istop = -1;
        const double Anorm = sqrt(st[S_TNORM2]);
        const double ynorm = sqrt(sum_partials(partialC + (size_t)rhs * nblk, nblk));
        const double epsx = Anorm * ynorm * eps;
        const double rnorm = st[S_PHIBAR];
        const double inf = 1.0 / 0.0;
        const double test1 = (ynorm == 0.0 || Anorm == 0.0) ? inf : rnorm / (Anorm * ynorm);
        const double test2 = Anorm == 0.0 ? inf : st[S_ROOT] / Anorm;
        const double Acond = st[S_GMAX] / st[S_GMIN];
        if (istop == 0 && rtol < 0.0) {
            // RL_MINRES_RULE: SciPy's own tests off, the caller's residual rule decides
            if (itn >= maxiter) istop = 6;
        } else if (istop == 0) {
            const double t1 = 1.0 + test1, t2 = 1.0 + test2;
            if (t2 <= 1.0) istop = 2;
            if (t1 <= 1.0) istop = 1;
            if (itn >= maxiter) istop = 6;
            if (Acond >= 0.1 / eps) istop = 4;
            if (epsx >= beta1) istop = 3;
            if (test2 <= rtol) istop = 2;
            if (test1 <= rtol) istop = 1;
        }
        if (istop != 0) {
            it[I_ISTOP] = istop;
            it[I_ACTIVE] = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *mb.giter = it0 + 1;
}

// ---- MINRES, two kernels per round -------------------------------------------
// The same recurrences arranged so that one round needs only the operator
// product and TWO vector kernels, and no normalised Lanczos vector is ever
// stored.  With y_j the unnormalised Lanczos vectors (y_0 = b, beta_{j+1} =
// ||y_j||, v_j = y_{j-1} / beta_j) the operator is applied to y_{k-1} itself
// and the 1/beta_k is folded into the consumers (the operator is linear).
// Round r (r = 1, 2, ...; q' = A y_{r-1} has just been computed):
//   P_r : if r >= 2, FINISH iteration k = r - 1 (its alfa_k came out of P_{r-1},
//         its beta_{k+1} out of B_{r-1}): plane rotation scalars, w_k, x_k,
//         partial ||x_k||^2 -- then START iteration r:
//         y' = q' / beta_r - (beta_r / beta_{r-1}) y_{r-2},  partial alfa_r = v_r . y'
//   B_r : if r >= 2, SciPy's stopping tests for iteration k = r - 1 (they need
//         ||x_k||); for systems that go on: y_r = y' - (alfa_r / beta_r) y_{r-1},
//         partial beta_{r+1}^2.
// Every rotating role has period two (the new y overwrites y_{r-2} in place, its
// last reader is the thread that writes it), so the parity of the round is a
// kernel ARGUMENT: a captured graph of an even number of rounds always starts
// at the same parity, and no kernel needs a memory round trip to find its
// buffers.  The round NUMBER (stopping tests, Lanczos index) is read from a
// counter advanced by the first kernel of the round's operator product.
// A system that stops at iteration k is frozen by B_{k+1}: x_k is already
// final, the started iteration k + 1 is abandoned (one operator product more
// than the textbook order, once per solve).  Iterates, iteration counts and
// exit codes are SciPy's.
//   optional fusion: q' = W g + eps (.) y_{r-1} computed row by row inside P
//   from the grid vector g (CSR W), so that the W product needs no kernel.
struct Minres2Bufs {
    double* tri[2];     // ping-pong: y_{r-2} (overwritten in place by the new y), y_{r-1}
    double* w[2];
    double* q;          // operator output (unfused), scratch of the checks
    double* x;
    int fuse_wt;        // host side: W^T rides inside the first grid kernel
    double* S[2];
    int* I;
    int* giter;         // number of the current round: bumped by the first kernel of the
                        // round's operator product (which does not read it)
    double* partA[2];
    double* partB;
    double* partC;
    double* lanczos;
    int lanczos_cap;
    // fused W product (W_indptr != NULL): q'[i] = sum_k W[i, k] g[col_k] + eps[i] y_{r-1}[i]
    const int* W_indptr;
    const int* W_indices;
    const double* W_data;
    int W_nnz;
    const int* W4_base;     // non-NULL: W as base column + 4 weights per row (SkiTerm)
    const double* W4_w;
    const double* g;        // [nrhs][ngrid]
    const double* eps;      // [n] or NULL
    int ngrid;
    // unfused W product whose kernel left the noise term out (eps_runs > 0): P adds
    // eps (.) y_{r-1} to the operator output it reads from q -- P holds y_{r-1}
    // anyway, the W kernel would read it a second time (same fma, same bits).  The
    // noise is constant per output and the rows of an output are contiguous: rows
    // [eps_end[k-1], eps_end[k]) carry eps_val[k]; no per-row array is read.
    int eps_runs;
    int eps_end[RL_MAX_D];
    double eps_val[RL_MAX_D];
    // SMALL systems through the polynomial form (poly_part != NULL; rl_lowrank.h,
    // "row-wise pieces"): a round is P and B alone.  B ends with the projection of
    // W^T y_r accumulated over its rows (a row block lies inside ONE output: poly_tab),
    // P starts by mixing the r D coefficients of its system (sum over the blocks of
    // every output, then sum_q B_q (x) C_q) and evaluates the four grid values of
    // each of its rows from them.  The round counter is two counters, each written
    // by the tail of one kernel and read by the other (giter: B -> P, giter2: P -> B).
    const int* poly_tab;      // [nblk][RL_PT]: first row, end row, output of a row block, first
                              // grid point (within the output) and number of grid points its rows touch
    const int* poly_ob;       // [D + 1]: first block of each output
    double* poly_part;        // [nrhs][nblk][RL_LR_RS], unnormalised basis
    const double* poly_M;     // [D][r][D][r]: nu_i nu_j sum_q B_q[a][b] C_q[i][j]
    const double* poly_beta;  // [r]
    int poly_D, poly_m;
    int* giter2;
    // B's vector work inside the NEXT round's projection (row-polynomial operator,
    // rl_rowpoly.h RpFuse; fuse_b != 0): B is its scalar head k_minres2_bh, which leaves
    // coef[rhs] = alfa_r / beta_r (0 for a system that stops); the projection forms y_r and
    // its partial squared norms nrmB[rhs][nrm_n], which P sums instead of partB.
    int fuse_b;
    double* coef;             // [nrhs]
    double* nrmB;             // [nrhs][nrm_n]
    int nrm_n;
    // P's vector work inside the round's EXPANSION (rl_rowpoly.h RpPFuse; fuse_p != 0, needs
    // fuse_b): P is its scalar head k_minres2_ph, which leaves the coefficients of the element
    // work in pc; the expansion's workgroups leave np partial sums per system in partA / partC
    // (np = its grid, above the RL_SOLVER_THREADS a block reduction takes in one go).
    int fuse_p;
    double* pc;               // [nrhs][RL_RP_PCW]
    int np;
};

#define RL_PT 5             // ints per entry of Minres2Bufs::poly_tab
#define RL_PG 2048          // grid points of a row block evaluated through LDS at most
// sum of acc[j] over the workgroup -> out[j] (thread j writes), j < RL_LR_RS
__device__ __forceinline__ void block_reduce_rs(const double acc[RL_LR_RS], double* red,
                                                double* out) {
#if defined(RL_EMU)
    for (int j = 0; j < RL_LR_RS; ++j) {
        const double v = block_reduce_sum(acc[j], red);
        if ((int)threadIdx.x == j) out[j] = v;
    }
#else
    // halving butterfly over the 64 lanes: at distance 32 a lane keeps one half of
    // the 24 sums and receives that half from its partner (12 exchanges), at 16 a
    // quarter (6), at 8 an eighth (3); the last three sums of a lane go through plain
    // exchanges at 4, 2, 1 -- 30 cross-lane moves instead of 24 x 6
    static_assert(RL_LR_RS == 24, "butterfly written for 24 sums");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    double a[12], b[6], c[3];
    const bool h5 = (lane & 32) != 0, h4 = (lane & 16) != 0, h3 = (lane & 8) != 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const double mine = h5 ? acc[12 + k] : acc[k], send = h5 ? acc[k] : acc[12 + k];
        a[k] = mine + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double mine = h4 ? a[6 + k] : a[k], send = h4 ? a[k] : a[6 + k];
        b[k] = mine + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double mine = h3 ? b[3 + k] : b[k], send = h3 ? b[k] : b[3 + k];
        c[k] = mine + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c[k] += __shfl_xor(c[k], 4, 64);
        c[k] += __shfl_xor(c[k], 2, 64);
        c[k] += __shfl_xor(c[k], 1, 64);
    }
    if ((lane & 7) == 0) {
        const int j0 = (h5 ? 12 : 0) + (h4 ? 6 : 0) + (h3 ? 3 : 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) red[wave * RL_LR_RS + j0 + k] = c[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < RL_LR_RS) {
        double r = 0.0;
        for (int i = 0; i < nw; ++i) r += red[i * RL_LR_RS + threadIdx.x];
        out[threadIdx.x] = r;
    }
    __syncthreads();
#endif
}

// projection partials of an arbitrary batch of data-space vectors (the first
// round's y_0 = b):  part[rhs][blk][j] = sum_{i in block} y_i (W Phi~)[i, j]
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_poly_project_rows(const double* __restrict__ Yv, int n, Minres2Bufs mb) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int lo = mb.poly_tab[RL_PT * blk], hi = mb.poly_tab[RL_PT * blk + 1];
    const int dout = mb.poly_tab[RL_PT * blk + 2];
    double acc[RL_LR_RS];
#pragma unroll
    for (int j = 0; j < RL_LR_RS; ++j) acc[j] = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        double w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = mb.W4_w[(size_t)4 * i + e];
        lr_row_accumulate(acc, mb.poly_beta, mb.W4_base[i] - dout * mb.poly_m, mb.poly_m, w,
                          Yv[(size_t)rhs * n + i]);
    }
    block_reduce_rs(acc, red, mb.poly_part + ((size_t)rhs * nblk + blk) * RL_LR_RS);
}


// init: x = 0, y_{-1} unused, y_0 = b, w = 0; partial = b.b from k_dot_partial(b, b)
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres2_init(const double* __restrict__ b, int n, const double* __restrict__ partial,
               Minres2Bufs mb) {
    const int rhs = blockIdx.y;
    const int nblk = gridDim.x;
    const double bb = sum_partials(partial + (size_t)rhs * nblk, nblk);
    const double beta1 = sqrt(bb);
    int lo, hi;
    block_range(n, &lo, &hi);
    const size_t off = (size_t)rhs * n;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const double bi = b[off + i];
        mb.x[off + i] = 0.0;
        mb.tri[0][off + i] = 0.0;      // y_{-1}: only ever multiplied by zero
        mb.tri[1][off + i] = bi;       // y_0
        mb.w[0][off + i] = 0.0;
        mb.w[1][off + i] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int c = 0; c < 2; ++c) {
            double* st = mb.S[c] + (size_t)rhs * S_NFIELDS;
            for (int f = 0; f < S_NFIELDS; ++f) st[f] = 0.0;
            st[S_BETA1] = beta1;
            st[S_BETA] = beta1;
            st[S_PHIBAR] = beta1;
            st[S_RHS1] = beta1;
            st[S_GMIN] = 1.7976931348623157e308;
            st[S_CS] = -1.0;
            st[S_BNORM] = beta1;
        }
        int* it = mb.I + rhs * I_NFIELDS;
        it[I_ITN] = 0;
        it[I_ISTOP] = beta1 > 0.0 ? 0 : RL_ISTOP_ZERO_RHS;
        it[I_ACTIVE] = beta1 > 0.0 ? 1 : 0;
        if (rhs == 0) {
            *mb.giter = mb.poly_part != nullptr ? 1 : 0;
            if (mb.poly_part != nullptr) *mb.giter2 = 0;
        }
    }
}

// (Measured and dropped, round 3: the plain large-system round compiled apart from the
// variants with the W product / the polynomial form inside -- 128 instead of 209 registers
// for P, 54 instead of 176 for B, four and eight waves per SIMD instead of two: P 1.65
// against 1.56-1.63 ms, B 0.53 against 0.52-0.59 at C5.  Nine and three vector streams at
// 5.6 TB/s: the kernels are at the rate HBM gives mixed reads and writes, not short of
// loads in flight.)
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres2_p(Minres2Bufs mb, int n, int par) {
    RL_STAMP(0);
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y;
    const int nblk = gridDim.x;
    const int p2 = par;                     // (round - 1) & 1, from the host
    const double* si = mb.S[p2] + (size_t)rhs * S_NFIELDS;
    double* so = mb.S[1 - p2] + (size_t)rhs * S_NFIELDS;
    // (polynomial rounds: hand the round number on to B, which reads giter2 only)
    if (mb.poly_part != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        *mb.giter2 = *mb.giter;
    if (!mb.I[rhs * I_NFIELDS + I_ACTIVE]) {
        // keep the two copies identical for frozen systems
        if (blockIdx.x == 0 && threadIdx.x < S_NFIELDS) so[threadIdx.x] = si[threadIdx.x];
        return;
    }
    double* r1 = mb.tri[p2];                // y_{r-2}; receives the new y
    const double* r2 = mb.tri[1 - p2];      // y_{r-1}
    double* y = r1;
    double* w1 = mb.w[p2];
    const double* w2 = mb.w[1 - p2];
    const double eps = 2.220446049250313e-16;
    const int round = *mb.giter;            // a number, not an address: off the critical path
    const bool fin = round >= 2;            // there is an iteration to finish

    // Every operand that does not depend on the scalars is requested FIRST (the
    // scalar chain below -- partial sums, square roots -- is a long dependent
    // sequence; the vector loads then overlap it).  PF rows per thread cover
    // n <= PF * 256 * gridDim.x; longer systems loop.
    constexpr int PF = 4;
    int lo, hi;
    block_range(n, &lo, &hi);
    const bool poly = mb.poly_part != nullptr;
    int dout = 0;
    if (poly) {
        lo = mb.poly_tab[RL_PT * blockIdx.x];
        hi = mb.poly_tab[RL_PT * blockIdx.x + 1];
        dout = mb.poly_tab[RL_PT * blockIdx.x + 2];
    }
    const size_t off = (size_t)rhs * n;
    const double* g = mb.W_indptr != nullptr ? mb.g + (size_t)rhs * mb.ngrid : nullptr;
    double pr2[PF], pr1[PF], pq[PF], pw1[PF], pw2[PF], px[PF];
    // the fused W rows: three dependent levels (row pointers -> entries -> grid
    // values), each level requested for all PF rows before the next is touched;
    // NZ entries per row are unrolled (cubic interpolation has 4), longer rows
    // finish in a loop
    constexpr int NZ = 4;
    // (the partial sums of the iteration being finished: one load per thread,
    // requested before everything else, reduced further down)
    double part_a = 0.0, part_b = 0.0;
    if ((int)threadIdx.x < nblk) {
        part_a = mb.partA[p2][(size_t)rhs * nblk + threadIdx.x];
        if (!mb.fuse_b) part_b = mb.partB[(size_t)rhs * nblk + threadIdx.x];
    }
    // (fused B: the projection's nrm_n partial norms of this system, RL_NRM_PF loads per thread
    // requested here with everything else, the rare rest in a loop further down)
    constexpr int RL_NRM_PF = 2;
    double pnb[RL_NRM_PF];
#pragma unroll
    for (int k = 0; k < RL_NRM_PF; ++k) {
        pnb[k] = 0.0;
        if (mb.fuse_b) {
            const int kk = threadIdx.x + k * blockDim.x;
            const double v = mb.nrmB[(size_t)rhs * mb.nrm_n + (kk < mb.nrm_n ? kk : 0)];
            pnb[k] = kk < mb.nrm_n ? v : 0.0;
        }
    }
    // (all loads unconditional, from clamped rows / entries, masked afterwards:
    // a conditional load is a branch with a full memory wait behind it)
    const int rlast = hi > lo ? hi - 1 : lo;       // lo < n: a valid row
    const int elast = mb.W_nnz > 0 ? mb.W_nnz - 1 : 0;
    const bool ell = g != nullptr && mb.W4_base != nullptr;
    int k0[PF], k1[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int i = lo + threadIdx.x + u * blockDim.x;
        const int ic = i < hi ? i : rlast;
        k0[u] = k1[u] = 0;
        if (g != nullptr && !ell) {
            const int a0 = mb.W_indptr[ic], a1 = mb.W_indptr[ic + 1];
            k0[u] = i < hi ? a0 : 0;
            k1[u] = i < hi ? a1 : 0;
        }
    }
    // structured W: base column and the four weights come with the first level
    int eb[PF];
    double ew[PF][NZ];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int i = lo + threadIdx.x + u * blockDim.x;
        const int ic = i < hi ? i : rlast;
        eb[u] = 0;
#pragma unroll
        for (int j = 0; j < NZ; ++j) ew[u][j] = 0.0;
        if (ell) {
            eb[u] = mb.W4_base[ic];
#pragma unroll
            for (int j = 0; j < NZ; ++j) ew[u][j] = mb.W4_w[(size_t)4 * ic + j];
        }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int i = lo + threadIdx.x + u * blockDim.x;
        const int ic = i < hi ? i : rlast;
        // (round 1 reads zeros for the finish operands: not waiting for the round number)
        pr2[u] = r2[off + ic];
        pr1[u] = r1[off + ic];
        pw1[u] = w1[off + ic];
        pw2[u] = w2[off + ic];
        px[u] = mb.x[off + ic];
        pq[u] = g == nullptr ? mb.q[off + ic] : 0.0;
    }
    // noise runs that meet this workgroup's rows [lo, hi): almost always one
    int ka = 0, kb = 0;
    if (g == nullptr && mb.eps_runs > 0) {
        while (ka + 1 < mb.eps_runs && lo >= mb.eps_end[ka]) ++ka;
        kb = ka;
        while (kb + 1 < mb.eps_runs && rlast >= mb.eps_end[kb]) ++kb;
    }
    auto eps_row = [&](int i) {
        double e = mb.eps_val[ka];
        for (int k = ka; k < kb; ++k)
            if (i >= mb.eps_end[k]) e = mb.eps_val[k + 1];
        return e;
    };
    if (g == nullptr && mb.eps_runs > 0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = lo + threadIdx.x + u * blockDim.x;
            pq[u] = fma(eps_row(i < hi ? i : rlast), pr2[u], pq[u]);
        }
    }
    if (ell && poly) {
        // mixed coefficients of this block's output from B's projection partials:
        //   Z[b][j] = sum_{blocks of output b} part[rhs][blk][j]
        //   Zh[i]   = sum_{b, j} M[dout][i][b][j] Z[b][j],
        //   M[a][i][b][j] = nu_i nu_j sum_q B_q[a][b] C_q[i][j]  (host, per parameter update)
        RL_STAMP(7);
        constexpr int RS = RL_LR_RS;
        const int D = mb.poly_D;
        double* Zs = red + 2 * RL_SOLVER_THREADS;      // [D][RS]
        double* Zh = Zs + D * RS;                       // [RS]
        for (int e = threadIdx.x; e < D * RS; e += blockDim.x) {
            const int b = e / RS, j = e - b * RS;
            double sum = 0.0;
            for (int kb = mb.poly_ob[b]; kb < mb.poly_ob[b + 1]; ++kb)
                sum += mb.poly_part[((size_t)rhs * nblk + kb) * RS + j];
            Zs[e] = sum;
        }
        __syncthreads();
        RL_STAMP(8);
        {
            // 24 dot products of length D * RS: ten threads each, then ten partial sums
            constexpr int NP = 10;
            double* Ps = Zh + RS;                       // [RS][NP]
            const int i = threadIdx.x % RS, part = threadIdx.x / RS;
            if (part < NP) {
                const double* mrow = mb.poly_M + ((size_t)dout * RS + i) * D * RS;
                double t = 0.0;
                for (int e = part; e < D * RS; e += NP) t = fma(mrow[e], Zs[e], t);
                Ps[i * NP + part] = t;
            }
            __syncthreads();
            if ((int)threadIdx.x < RS) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < NP; ++k) t += Ps[threadIdx.x * NP + k];
                Zh[threadIdx.x] = t;
            }
            __syncthreads();
        }
        RL_STAMP(9);
        double zr[RS];
#pragma unroll
        for (int j = 0; j < RS; ++j) zr[j] = Zh[j];
        // Where the block's rows are about as dense as the grid points they touch
        // (C2: 1000 rows on 1003 points) the grid range is evaluated ONCE into LDS,
        // four points per thread, and the rows gather from it -- a quarter of the
        // recurrences of evaluating four points per row.
        const int g0 = mb.poly_tab[RL_PT * blockIdx.x + 3], glen = mb.poly_tab[RL_PT * blockIdx.x + 4];
        const bool via_grid = glen <= RL_PG && glen <= 2 * (hi - lo);
        double* Gs = Zh + RS + 10 * RS;                  // [RL_PG]
        if (via_grid) {
            for (int p0 = threadIdx.x; p0 < glen; p0 += 4 * blockDim.x) {
                int np[4];
                double gp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) np[e] = g0 + p0 + e * (int)blockDim.x;
                lr_point_values(zr, mb.poly_beta, np, mb.poly_m, gp);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (p0 + e * (int)blockDim.x < glen) Gs[p0 + e * blockDim.x] = gp[e];
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (lo + u * (int)blockDim.x >= hi) break;       // (uniform: no row of this group)
            const int i = lo + threadIdx.x + u * blockDim.x;
            const int ic = i < hi ? i : rlast;
            double gv[NZ];
            if (via_grid) {
                const int p = eb[u] - dout * mb.poly_m - g0;
#pragma unroll
                for (int j = 0; j < NZ; ++j) gv[j] = Gs[p + j < glen ? p + j : glen - 1];
            } else {
                lr_row_values(zr, mb.poly_beta, eb[u] - dout * mb.poly_m, mb.poly_m, gv);
            }
            double qi = mb.eps != nullptr ? mb.eps[ic] * pr2[u] : 0.0;
#pragma unroll
            for (int j = 0; j < NZ; ++j) qi = fma(ew[u][j], gv[j], qi);
            pq[u] = qi;
        }
    } else if (ell) {
        double gv[PF][NZ];
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int j = 0; j < NZ; ++j) gv[u][j] = g[eb[u] + j];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = lo + threadIdx.x + u * blockDim.x;
            const int ic = i < hi ? i : rlast;
            double qi = mb.eps != nullptr ? mb.eps[ic] * pr2[u] : 0.0;
#pragma unroll
            for (int j = 0; j < NZ; ++j) qi = fma(ew[u][j], gv[u][j], qi);
            pq[u] = qi;
        }
    } else if (g != nullptr) {
        double wa[PF][NZ];
        int wc[PF][NZ];
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int j = 0; j < NZ; ++j) {
                const int k = k0[u] + j < elast ? k0[u] + j : elast;
                const double a = mb.W_data[k];
                wc[u][j] = mb.W_indices[k];
                wa[u][j] = k0[u] + j < k1[u] ? a : 0.0;
            }
        double gv[PF][NZ];
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int j = 0; j < NZ; ++j) gv[u][j] = g[wc[u][j]];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = lo + threadIdx.x + u * blockDim.x;
            const int ic = i < hi ? i : rlast;
            double qi = mb.eps != nullptr ? mb.eps[ic] * pr2[u] : 0.0;
#pragma unroll
            for (int j = 0; j < NZ; ++j) qi = fma(wa[u][j], gv[u][j], qi);
            pq[u] = qi;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u)
            for (int k = k0[u] + NZ; k < k1[u]; ++k)
                pq[u] = fma(mb.W_data[k], g[mb.W_indices[k]], pq[u]);
    }

    // scalars of the iteration being finished (k = round - 1); statement by
    // statement SciPy's (see k_minres_c)
    double alfa = 0.0, beta = si[S_BETA], oldb = si[S_BETA];
    double tnorm2 = 0.0, delta = 0.0, gbar = 0.0, epsln = 0.0, dbar = 0.0, root = 0.0;
    double gamma = 1.0, cs = 0.0, sn = 0.0, phi = 0.0, phibar = 0.0, denom = 0.0, oldeps = 0.0;
    RL_STAMP(1);
    if (mb.fuse_b) {
#pragma unroll
        for (int k = 0; k < RL_NRM_PF; ++k) part_b += pnb[k];
        for (int k = threadIdx.x + RL_NRM_PF * blockDim.x; k < mb.nrm_n; k += blockDim.x)
            part_b += mb.nrmB[(size_t)rhs * mb.nrm_n + k];
    }
    block_reduce_sum2(part_a, part_b, red);
    RL_STAMP(2);
    if (fin) {
        alfa = part_a;
        beta = sqrt(part_b > 0.0 ? part_b : 0.0);
        tnorm2 = si[S_TNORM2] + alfa * alfa + oldb * oldb + beta * beta;
        const double cs0 = si[S_CS], sn0 = si[S_SN], dbar0 = si[S_DBAR];
        oldeps = si[S_EPSLN];
        delta = cs0 * dbar0 + sn0 * alfa;
        gbar = sn0 * dbar0 - cs0 * alfa;
        epsln = sn0 * beta;
        dbar = -cs0 * beta;
        root = hypot2(gbar, dbar);
        gamma = hypot2(gbar, beta);
        gamma = gamma > eps ? gamma : eps;
        cs = gbar / gamma;
        sn = beta / gamma;
        phi = cs * si[S_PHIBAR];
        phibar = sn * si[S_PHIBAR];
        denom = 1.0 / gamma;
    }
    // v_k = y_{k-1} / beta_k = r1 / oldb (finish);  v_r = r2 / beta (start)
    const double oinv = oldb > 0.0 ? 1.0 / oldb : 0.0;
    const double sinv = beta > 0.0 ? 1.0 / beta : 0.0;
    const double coef = fin ? beta * oinv : 0.0;

    RL_STAMP(3);
    double accA = 0.0, accC = 0.0;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int i = lo + threadIdx.x + u * blockDim.x;
        if (i < hi) {
            if (fin) {
                const double wn = (pr1[u] * oinv - oldeps * pw1[u] - delta * pw2[u]) * denom;
                w1[off + i] = wn;
                const double xi = px[u] + phi * wn;
                mb.x[off + i] = xi;
                accC = fma(xi, xi, accC);
            }
            const double yi = pq[u] * sinv - coef * pr1[u];
            y[off + i] = yi;
            accA = fma(pr2[u] * sinv, yi, accA);
        }
    }
    int it0 = lo + threadIdx.x + PF * blockDim.x;
    if (g == nullptr) {
        // long systems (operator output in q): the remaining rows PF at a time,
        // every load of a group requested before the first is used
        // (round 4, measured and dropped: two ADJACENT rows per thread and stream, twelve
        // 16-byte loads per group instead of twenty-four 8-byte ones, block borders kept even:
        // 1741 vs 1725-1860 us at 129 systems, 245 vs 244-252 at 17 -- inside the box-to-box
        // range; the kernel moves its nine streams at 5.2-5.4 TB/s either way)
        for (; it0 + (PF - 1) * (int)blockDim.x < hi; it0 += PF * blockDim.x) {
            double tr2[PF], tr1[PF], tq[PF], tw1[PF], tw2[PF], tx[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const size_t i = off + it0 + u * blockDim.x;
                tr2[u] = r2[i];
                tr1[u] = r1[i];
                tq[u] = mb.q[i];
                tw1[u] = w1[i];
                tw2[u] = w2[i];
                tx[u] = mb.x[i];
            }
            if (mb.eps_runs > 0) {
#pragma unroll
                for (int u = 0; u < PF; ++u)
                    tq[u] = fma(eps_row(it0 + u * (int)blockDim.x), tr2[u], tq[u]);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const size_t i = off + it0 + u * blockDim.x;
                if (fin) {
                    const double wn = (tr1[u] * oinv - oldeps * tw1[u] - delta * tw2[u]) * denom;
                    w1[i] = wn;
                    const double xi = tx[u] + phi * wn;
                    mb.x[i] = xi;
                    accC = fma(xi, xi, accC);
                }
                const double yi = tq[u] * sinv - coef * tr1[u];
                y[i] = yi;
                accA = fma(tr2[u] * sinv, yi, accA);
            }
        }
    }
    for (int i = it0; i < hi; i += blockDim.x) {
        const double r2i = r2[off + i];
        double qi;
        if (ell) {
            qi = mb.eps != nullptr ? mb.eps[i] * r2i : 0.0;
            const int b = mb.W4_base[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) qi = fma(mb.W4_w[(size_t)4 * i + j], g[b + j], qi);
        } else if (g != nullptr) {
            qi = mb.eps != nullptr ? mb.eps[i] * r2i : 0.0;
            const int k1 = mb.W_indptr[i + 1];
            for (int k = mb.W_indptr[i]; k < k1; ++k) qi = fma(mb.W_data[k], g[mb.W_indices[k]], qi);
        } else {
            qi = mb.q[off + i];
            if (mb.eps_runs > 0) qi = fma(eps_row(i), r2i, qi);
        }
        double r1i = 0.0;
        if (fin) {
            r1i = r1[off + i];
            const double wn = (r1i * oinv - oldeps * w1[off + i] - delta * w2[off + i]) * denom;
            w1[off + i] = wn;
            const double xi = mb.x[off + i] + phi * wn;
            mb.x[off + i] = xi;
            accC = fma(xi, xi, accC);
        }
        const double yi = qi * sinv - coef * r1i;
        y[off + i] = yi;
        accA = fma(r2i * sinv, yi, accA);
    }
    RL_STAMP(4);
    block_reduce_sum2(accA, accC, red);
    RL_STAMP(5);
    if (threadIdx.x == 0) {
        mb.partA[1 - p2][(size_t)rhs * nblk + blockIdx.x] = accA;
        mb.partC[(size_t)rhs * nblk + blockIdx.x] = accC;
    }

    // publish the new scalar state: one field per thread of workgroup 0 (every
    // thread holds the scalars; a single thread writing 45 values one after the
    // other was 2.9 us of this kernel)
    if (blockIdx.x == 0 && threadIdx.x < S_NFIELDS) {
        const int f = threadIdx.x;
        double v = si[f];
        if (fin) {
            const double gmax = si[S_GMAX] > gamma ? si[S_GMAX] : gamma;
            const double gmin = si[S_GMIN] < gamma ? si[S_GMIN] : gamma;
            const double z = si[S_RHS1] / gamma;
            switch (f) {
                case S_OLDB: v = oldb; break;
                case S_BETA: v = beta; break;
                case S_TNORM2: v = tnorm2; break;
                case S_DBAR: v = dbar; break;
                case S_EPSLN: v = epsln; break;
                case S_CS: v = cs; break;
                case S_SN: v = sn; break;
                case S_PHIBAR: v = phibar; break;
                case S_PHI: v = phi; break;
                case S_ALFA: v = alfa; break;
                case S_OLDEPS: v = oldeps; break;
                case S_DELTA: v = delta; break;
                case S_DENOM: v = denom; break;
                case S_ROOT: v = root; break;
                case S_GBAR: v = gbar; break;
                case S_GMAX: v = gmax; break;
                case S_GMIN: v = gmin; break;
                case S_RHS1: v = si[S_RHS2] - delta * z; break;
                case S_RHS2: v = -epsln * z; break;
                default: break;
            }
        }
        so[f] = v;
        if (f == 0 && fin) {
            // Lanczos tridiagonal entries (stochastic Lanczos quadrature of log det)
            const int itn = round - 2;          // iterations finished before this one
            if (mb.lanczos != nullptr && itn < mb.lanczos_cap) {
                double* lz = mb.lanczos + ((size_t)rhs * mb.lanczos_cap + itn) * 2;
                lz[0] = alfa;
                lz[1] = beta;
            }
        }
    }
    RL_STAMP(6);
}

// SciPy's stopping tests for the iteration whose state is `st` (k_minres_test)
__device__ __forceinline__ int minres_stop_test(const double* st, double ynorm, int itn,
                                                double rtol, int maxiter) {
    const double eps = 2.220446049250313e-16;
    int istop = 0;
    const double beta1 = st[S_BETA1];
    if (itn == 1 && st[S_BETA] / beta1 <= 10 * eps) istop = -1;
    const double Anorm = sqrt(st[S_TNORM2]);
    const double epsx = Anorm * ynorm * eps;
    const double rnorm = st[S_PHIBAR];
    const double inf = 1.0 / 0.0;
    const double test1 = (ynorm == 0.0 || Anorm == 0.0) ? inf : rnorm / (Anorm * ynorm);
    const double test2 = Anorm == 0.0 ? inf : st[S_ROOT] / Anorm;
    const double Acond = st[S_GMAX] / st[S_GMIN];
    if (istop == 0 && rtol < 0.0) {
        // RL_MINRES_RULE: SciPy's own tests off, the caller's residual rule decides
        if (itn >= maxiter) istop = 6;
    } else if (istop == 0) {
        const double t1 = 1.0 + test1, t2 = 1.0 + test2;
        if (t2 <= 1.0) istop = 2;
        if (t1 <= 1.0) istop = 1;
        if (itn >= maxiter) istop = 6;
        if (Acond >= 0.1 / eps) istop = 4;
        if (epsx >= beta1) istop = 3;
        if (test2 <= rtol) istop = 2;
        if (test1 <= rtol) istop = 1;
    }
    return istop;
}

static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres2_b(Minres2Bufs mb, int n, int par, double rtol, int maxiter) {
    RL_STAMP(40);
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y;
    const int nblk = gridDim.x;
    const int p2 = par;
    const double* so = mb.S[1 - p2] + (size_t)rhs * S_NFIELDS;    // published by P this round
    int* iv = mb.I + rhs * I_NFIELDS;
    bool go = iv[I_ACTIVE] != 0;
    // operands first: they do not depend on the scalar chain of the tests
    const double* r2 = mb.tri[1 - p2];
    double* y = mb.tri[p2];
    int lo, hi;
    block_range(n, &lo, &hi);
    const bool poly = mb.poly_part != nullptr;
    int dout = 0;
    if (poly) {
        lo = mb.poly_tab[RL_PT * blockIdx.x];
        hi = mb.poly_tab[RL_PT * blockIdx.x + 1];
        dout = mb.poly_tab[RL_PT * blockIdx.x + 2];
        // (the next P reads giter only: written here, read by no workgroup of B)
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *mb.giter = *mb.giter2 + 1;
    }
    const size_t off = (size_t)rhs * n;
    constexpr int PF = 4;
    double py[PF], pr[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int i = lo + threadIdx.x + u * blockDim.x;
        py[u] = pr[u] = 0.0;
        if (go && i < hi) {
            py[u] = y[off + i];
            pr[u] = r2[off + i];
        }
    }
    // (polynomial rounds: base column and weights of the rows, requested with them)
    int pb[PF];
    double pw[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        pb[u] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) pw[u][e] = 0.0;
        const int i = lo + threadIdx.x + u * blockDim.x;
        if (poly && go && i < hi) {
            pb[u] = mb.W4_base[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) pw[u][e] = mb.W4_w[(size_t)4 * i + e];
        }
    }
    // ||x||^2 of the iteration under test and alfa of the one under way (the
    // same reduction P uses: alfa must be bit-identical in both kernels)
    double xx = 0.0, alfa = 0.0;
    sum2_partials(go ? mb.partC + (size_t)rhs * nblk : nullptr,
                  go ? mb.partA[1 - p2] + (size_t)rhs * nblk : nullptr, nblk, red, &xx, &alfa);
    const int round = poly ? *mb.giter2 : *mb.giter;
    if (go && round >= 2) {
        // every workgroup of this system takes the same decision from the same
        // numbers; workgroup 0 records it
        const double ynorm = sqrt(xx);
        const int istop = minres_stop_test(so, ynorm, round - 1, rtol, maxiter);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            iv[I_ITN] = round - 1;
            if (istop != 0) {
                iv[I_ISTOP] = istop;
                iv[I_ACTIVE] = 0;
            }
        }
        if (istop != 0) go = false;
    }
    if (go) {
        const double beta = so[S_BETA];
        const double coef = alfa / beta;      // SciPy: y -= (alfa / beta) r2, r2 unnormalised
        double acc = 0.0;
        double yn[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = lo + threadIdx.x + u * blockDim.x;
            yn[u] = 0.0;
            if (i < hi) {
                const double yi = py[u] - coef * pr[u];
                y[off + i] = yi;
                yn[u] = yi;
                acc = fma(yi, yi, acc);
            }
        }
        RL_STAMP(42);
        if (poly) {
            // projection of W^T y_r over this block's rows (all inside output dout;
            // a block has at most PF * blockDim.x rows)
            double pa[RL_LR_RS];
#pragma unroll
            for (int j = 0; j < RL_LR_RS; ++j) pa[j] = 0.0;
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (lo + u * (int)blockDim.x >= hi) break;   // (uniform: no row of this group)
                const int i = lo + threadIdx.x + u * blockDim.x;
                // (rows past the block: zero weights, zero y)
                lr_row_accumulate(pa, mb.poly_beta, (i < hi ? pb[u] : dout * mb.poly_m) -
                                  dout * mb.poly_m, mb.poly_m, pw[u], yn[u]);
            }
            RL_STAMP(43);
            block_reduce_rs(pa, red + 2 * RL_SOLVER_THREADS,
                            mb.poly_part + ((size_t)rhs * nblk + blockIdx.x) * RL_LR_RS);
            RL_STAMP(44);
        }
        int it0 = lo + threadIdx.x + PF * blockDim.x;
        for (; it0 + (PF - 1) * (int)blockDim.x < hi; it0 += PF * blockDim.x) {
            double ty[PF], tr[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                ty[u] = y[off + it0 + u * blockDim.x];
                tr[u] = r2[off + it0 + u * blockDim.x];
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const double yi = ty[u] - coef * tr[u];
                y[off + it0 + u * blockDim.x] = yi;
                acc = fma(yi, yi, acc);
            }
        }
        for (int i = it0; i < hi; i += blockDim.x) {
            const double yi = y[off + i] - coef * r2[off + i];
            y[off + i] = yi;
            acc = fma(yi, yi, acc);
        }
        acc = block_reduce_sum(acc, red);
        if (threadIdx.x == 0) mb.partB[(size_t)rhs * nblk + blockIdx.x] = acc;
    }
    RL_STAMP(41);
}

// B's scalar head (Minres2Bufs::fuse_b): the stopping tests of iteration round - 1 and the
// coefficient of  y_r = y' - (alfa_r / beta_r) y_{r-1};  the vector update itself and
// ||y_r||^2 are the next round's projection (rl_rowpoly.h, RpFuse).   grid (nrhs)
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres2_bh(Minres2Bufs mb, int nblk, int par, double rtol, int maxiter) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.x;
    const int p2 = par;
    const double* so = mb.S[1 - p2] + (size_t)rhs * S_NFIELDS;
    int* iv = mb.I + rhs * I_NFIELDS;
    bool go = iv[I_ACTIVE] != 0;
    double xx = 0.0, alfa = 0.0;
    if (nblk <= (int)blockDim.x)
        sum2_partials(go ? mb.partC + (size_t)rhs * nblk : nullptr,
                      go ? mb.partA[1 - p2] + (size_t)rhs * nblk : nullptr, nblk, red, &xx, &alfa);
    else        // (P inside the expansion: one partial per expansion workgroup)
        sum2_partials_long(go ? mb.partC + (size_t)rhs * nblk : nullptr,
                           go ? mb.partA[1 - p2] + (size_t)rhs * nblk : nullptr, nblk, nblk, red,
                           &xx, &alfa);
    const int round = *mb.giter;
    if (go && round >= 2) {
        const int istop = minres_stop_test(so, sqrt(xx), round - 1, rtol, maxiter);
        if (threadIdx.x == 0) {
            iv[I_ITN] = round - 1;
            if (istop != 0) {
                iv[I_ISTOP] = istop;
                iv[I_ACTIVE] = 0;
            }
        }
        if (istop != 0) go = false;
    }
    if (threadIdx.x == 0) mb.coef[rhs] = go ? alfa / so[S_BETA] : 0.0;
}

// B's vector work as a kernel of its own, for rounds whose P runs inside the W product
// (k_spmv_w_staged_p, rl_rowpoly.h) but whose operator has no projection to carry B: after
// k_minres2_bh (stopping tests, coef)   y_r = y' - coef y_{r-1},  partB = partial ||y_r||^2 --
// k_minres2_b's statements without its scalar head (every workgroup of B re-summed the
// partial sums of its system: with one partial per W row block that is 2 x 3907 values).
//   grid (nblk, nrhs)
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres2_bv(Minres2Bufs mb, int n, int par) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y;
    if (mb.I[rhs * I_NFIELDS + I_ACTIVE] == 0) return;      // (frozen, or stopped by this round's head)
    const double coef = mb.coef[rhs];
    const double* r2 = mb.tri[1 - par];
    double* y = mb.tri[par];
    int lo, hi;
    block_range(n, &lo, &hi);
    const size_t off = (size_t)rhs * n;
    constexpr int PF = 4;
    double acc = 0.0;
    int it0 = lo + threadIdx.x;
    for (; it0 + (PF - 1) * (int)blockDim.x < hi; it0 += PF * blockDim.x) {
        double ty[PF], tr[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            ty[u] = y[off + it0 + u * blockDim.x];
            tr[u] = r2[off + it0 + u * blockDim.x];
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const double yi = ty[u] - coef * tr[u];
            y[off + it0 + u * blockDim.x] = yi;
            acc = fma(yi, yi, acc);
        }
    }
    for (int i = it0; i < hi; i += blockDim.x) {
        const double yi = y[off + i] - coef * r2[off + i];
        y[off + i] = yi;
        acc = fma(yi, yi, acc);
    }
    acc = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) mb.partB[(size_t)rhs * gridDim.x + blockIdx.x] = acc;
}

// P's scalar head (Minres2Bufs::fuse_p): everything of k_minres2_p that is per SYSTEM --
// the sums of the partial dot products, the plane rotation of the iteration being finished,
// the new scalar state, the Lanczos record -- once per system instead of once per
// workgroup, and the coefficients of the element work for the expansion (rl_rowpoly.h
// RpPFuse, which lists pc's fields).  Statement by statement k_minres2_p's.   grid (nrhs)
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_minres2_ph(Minres2Bufs mb, int par) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.x;
    const int p2 = par;
    const double* si = mb.S[p2] + (size_t)rhs * S_NFIELDS;
    double* so = mb.S[1 - p2] + (size_t)rhs * S_NFIELDS;
    double* pc = mb.pc + (size_t)rhs * RL_RP_PCW;
    if (!mb.I[rhs * I_NFIELDS + I_ACTIVE]) {
        if (threadIdx.x < S_NFIELDS) so[threadIdx.x] = si[threadIdx.x];
        if (threadIdx.x == 0) pc[0] = 0.0;
        return;
    }
    const double eps = 2.220446049250313e-16;
    const int round = *mb.giter;
    const bool fin = round >= 2;
    double part_a = 0.0, part_b = 0.0;
    sum2_partials_long(mb.partA[p2] + (size_t)rhs * mb.np, mb.nrmB + (size_t)rhs * mb.nrm_n,
                       fin ? mb.np : 0, fin ? mb.nrm_n : 0, red, &part_a, &part_b);
    double alfa = 0.0, beta = si[S_BETA], oldb = si[S_BETA];
    double tnorm2 = 0.0, delta = 0.0, gbar = 0.0, epsln = 0.0, dbar = 0.0, root = 0.0;
    double gamma = 1.0, cs = 0.0, sn = 0.0, phi = 0.0, phibar = 0.0, denom = 0.0, oldeps = 0.0;
    if (fin) {
        alfa = part_a;
        beta = sqrt(part_b > 0.0 ? part_b : 0.0);
        tnorm2 = si[S_TNORM2] + alfa * alfa + oldb * oldb + beta * beta;
        const double cs0 = si[S_CS], sn0 = si[S_SN], dbar0 = si[S_DBAR];
        oldeps = si[S_EPSLN];
        delta = cs0 * dbar0 + sn0 * alfa;
        gbar = sn0 * dbar0 - cs0 * alfa;
        epsln = sn0 * beta;
        dbar = -cs0 * beta;
        root = hypot2(gbar, dbar);
        gamma = hypot2(gbar, beta);
        gamma = gamma > eps ? gamma : eps;
        cs = gbar / gamma;
        sn = beta / gamma;
        phi = cs * si[S_PHIBAR];
        phibar = sn * si[S_PHIBAR];
        denom = 1.0 / gamma;
    }
    const double oinv = oldb > 0.0 ? 1.0 / oldb : 0.0;
    const double sinv = beta > 0.0 ? 1.0 / beta : 0.0;
    const double coef = fin ? beta * oinv : 0.0;
    if (threadIdx.x < RL_RP_PCW) {
        double v = 0.0;
        switch (threadIdx.x) {
            case 0: v = 1.0; break;
            case 1: v = fin ? 1.0 : 0.0; break;
            case 2: v = oinv; break;
            case 3: v = oldeps; break;
            case 4: v = delta; break;
            case 5: v = denom; break;
            case 6: v = phi; break;
            case 7: v = sinv; break;
            case 8: v = coef; break;
            default: break;
        }
        pc[threadIdx.x] = v;
    }
    if (threadIdx.x < S_NFIELDS) {
        const int f = threadIdx.x;
        double v = si[f];
        if (fin) {
            const double gmax = si[S_GMAX] > gamma ? si[S_GMAX] : gamma;
            const double gmin = si[S_GMIN] < gamma ? si[S_GMIN] : gamma;
            const double z = si[S_RHS1] / gamma;
            switch (f) {
                case S_OLDB: v = oldb; break;
                case S_BETA: v = beta; break;
                case S_TNORM2: v = tnorm2; break;
                case S_DBAR: v = dbar; break;
                case S_EPSLN: v = epsln; break;
                case S_CS: v = cs; break;
                case S_SN: v = sn; break;
                case S_PHIBAR: v = phibar; break;
                case S_PHI: v = phi; break;
                case S_ALFA: v = alfa; break;
                case S_OLDEPS: v = oldeps; break;
                case S_DELTA: v = delta; break;
                case S_DENOM: v = denom; break;
                case S_ROOT: v = root; break;
                case S_GBAR: v = gbar; break;
                case S_GMAX: v = gmax; break;
                case S_GMIN: v = gmin; break;
                case S_RHS1: v = si[S_RHS2] - delta * z; break;
                case S_RHS2: v = -epsln * z; break;
                default: break;
            }
        }
        so[f] = v;
        if (f == 0 && fin) {
            const int itn = round - 2;
            if (mb.lanczos != nullptr && itn < mb.lanczos_cap) {
                double* lz = mb.lanczos + ((size_t)rhs * mb.lanczos_cap + itn) * 2;
                lz[0] = alfa;
                lz[1] = beta;
            }
        }
    }
}

// ---- CG ---------------------------------------------------------------------
// init: x = 0, r = b, p = 0; partial = b.b
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_cg_init(const double* __restrict__ b, int n, const double* __restrict__ partial,
          double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
          double* __restrict__ S, int* __restrict__ I, double rtol) {
    const int rhs = blockIdx.y;
    const int nblk = gridDim.x;
    const double bb = sum_partials(partial + (size_t)rhs * nblk, nblk);
    int lo, hi;
    block_range(n, &lo, &hi);
    const size_t off = (size_t)rhs * n;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        x[off + i] = 0.0;
        r[off + i] = b[off + i];
        p[off + i] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double* st = S + (size_t)rhs * S_NFIELDS;
        for (int f = 0; f < S_NFIELDS; ++f) st[f] = 0.0;
        st[S_BNORM] = sqrt(bb);
        st[S_CG_ATOL] = rtol * sqrt(bb);
        st[S_RHO] = bb;        // r.r with r = b
        int* it = I + rhs * I_NFIELDS;
        it[I_ITN] = 0;
        it[I_ISTOP] = bb > 0.0 ? 0 : RL_ISTOP_ZERO_RHS;
        it[I_ACTIVE] = bb > 0.0 ? 1 : 0;
    }
}

// loop head (one thread per rhs): rho = sum partial (r.r) unless first;
// SciPy tests ||r|| < atol BEFORE the update; maxiter exhaustion
static __global__ void k_cg_head(double* __restrict__ S, int* __restrict__ I,
                          const double* __restrict__ partialR, int nblk, int nrhs, int first,
                          int maxiter) {
    const int rhs = blockIdx.x * blockDim.x + threadIdx.x;
    if (rhs >= nrhs) return;
    int* it = I + rhs * I_NFIELDS;
    if (!it[I_ACTIVE]) return;
    double* st = S + (size_t)rhs * S_NFIELDS;
    if (!first) {
        st[S_RHO_PREV] = st[S_RHO];
        st[S_RHO] = sum_partials(partialR + (size_t)rhs * nblk, nblk);
    }
    if (sqrt(st[S_RHO]) < st[S_CG_ATOL]) {
        it[I_ISTOP] = 1;
        it[I_ACTIVE] = 0;
    } else if (it[I_ITN] >= maxiter) {
        it[I_ISTOP] = 6;
        it[I_ACTIVE] = 0;
    }
}

// p = r + (rho/rho_prev) p   (p = r on the first iteration)
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_cg_p(double* __restrict__ p, const double* __restrict__ r, int n,
       const double* __restrict__ S, const int* __restrict__ I) {
    const int rhs = blockIdx.y;
    if (!I[rhs * I_NFIELDS + I_ACTIVE]) return;
    const double* st = S + (size_t)rhs * S_NFIELDS;
    const int first = I[rhs * I_NFIELDS + I_ITN] == 0;
    const double beta = first ? 0.0 : st[S_RHO] / st[S_RHO_PREV];
    int lo, hi;
    block_range(n, &lo, &hi);
    const size_t off = (size_t)rhs * n;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x)
        p[off + i] = first ? r[off + i] : beta * p[off + i] + r[off + i];
}

// alpha = rho / sum partialPQ; x += alpha p; r -= alpha q; partialR = r.r
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_cg_update(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
            const double* __restrict__ q, int n, const double* __restrict__ S,
            int* __restrict__ I, const double* __restrict__ partialPQ,
            double* __restrict__ partialR) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y;
    if (!I[rhs * I_NFIELDS + I_ACTIVE]) return;
    const int nblk = gridDim.x;
    const double pq = sum_partials(partialPQ + (size_t)rhs * nblk, nblk);
    const double alpha = S[(size_t)rhs * S_NFIELDS + S_RHO] / pq;
    int lo, hi;
    block_range(n, &lo, &hi);
    const size_t off = (size_t)rhs * n;
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        x[off + i] += alpha * p[off + i];
        const double ri = r[off + i] - alpha * q[off + i];
        r[off + i] = ri;
        acc = fma(ri, ri, acc);
    }
    acc = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) partialR[(size_t)rhs * nblk + blockIdx.x] = acc;
}

// bump iteration counters of active systems (callback count in the reference)
static __global__ void k_count_iter(int* __restrict__ I, int nrhs) {
    const int rhs = blockIdx.x * blockDim.x + threadIdx.x;
    if (rhs >= nrhs) return;
    if (I[rhs * I_NFIELDS + I_ACTIVE]) I[rhs * I_NFIELDS + I_ITN] += 1;
}

// ---- gradient partial sums --------------------------------------------------
// out[v][a][b] = sum_i U[v][a*m + i] * V[v][b*m + i]   (D x D Gram per vector)
//   grid (D*D, nvec)
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_cross_dots(const double* __restrict__ U, const double* __restrict__ V, int D, int m,
             double* __restrict__ out) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int a = blockIdx.x / D, b = blockIdx.x % D, v = blockIdx.y;
    const double* u = U + ((size_t)v * D + a) * m;
    const double* w = V + ((size_t)v * D + b) * m;
    double acc = 0.0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) acc = fma(u[i], w[i], acc);
    acc = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) out[((size_t)v * D + a) * D + b] = acc;
}

// The same with a 4 x 8 block of (a, b) pairs per workgroup: every loaded value
// feeds 8 (or 4) multiply-adds, and a row of U / V is read D/8 (D/4) times
// instead of D times (the one-pair-per-workgroup form above re-reads both
// operands D times: 20.8 GB per top row at C5).
//   grid (ceil(D / 4) * ceil(D / 8), nvec)
#define RL_XD_A 4
#define RL_XD_B 8
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_cross_dots_tiled(const double* __restrict__ U, const double* __restrict__ V, int D, int m,
                   double* __restrict__ out) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int nbb = (D + RL_XD_B - 1) / RL_XD_B;
    const int a0 = (blockIdx.x / nbb) * RL_XD_A, b0 = (blockIdx.x % nbb) * RL_XD_B;
    const int v = blockIdx.y;
    const double* u = U + (size_t)v * D * m;
    const double* w = V + (size_t)v * D * m;
    double acc[RL_XD_A][RL_XD_B];
#pragma unroll
    for (int a = 0; a < RL_XD_A; ++a)
#pragma unroll
        for (int b = 0; b < RL_XD_B; ++b) acc[a][b] = 0.0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        double ua[RL_XD_A], wb[RL_XD_B];
        // (rows past D are read from the last valid row and never stored)
#pragma unroll
        for (int a = 0; a < RL_XD_A; ++a) ua[a] = u[(size_t)(a0 + a < D ? a0 + a : D - 1) * m + i];
#pragma unroll
        for (int b = 0; b < RL_XD_B; ++b) wb[b] = w[(size_t)(b0 + b < D ? b0 + b : D - 1) * m + i];
#pragma unroll
        for (int a = 0; a < RL_XD_A; ++a)
#pragma unroll
            for (int b = 0; b < RL_XD_B; ++b) acc[a][b] = fma(ua[a], wb[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < RL_XD_A; ++a)
#pragma unroll
        for (int b = 0; b < RL_XD_B; ++b) {
            const double s = block_reduce_sum(acc[a][b], red);
            if (threadIdx.x == 0 && a0 + a < D && b0 + b < D)
                out[((size_t)v * D + a0 + a) * D + b0 + b] = s;
        }
}

// out[v][d] = sum_{i in segment d} U[v][i] * V[v][i];  segments given by
// offsets[D+1] (per-output slices of a data-space vector)   grid (D, nvec)
static __global__ void __launch_bounds__(RL_SOLVER_THREADS)
k_segment_dots(const double* __restrict__ U, const double* __restrict__ V,
               const int* __restrict__ offsets, int n, int D, double* __restrict__ out) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int d = blockIdx.x, v = blockIdx.y;
    const double* u = U + (size_t)v * n;
    const double* w = V + (size_t)v * n;
    double acc = 0.0;
    for (int i = offsets[d] + threadIdx.x; i < offsets[d + 1]; i += blockDim.x)
        acc = fma(u[i], w[i], acc);
    acc = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) out[(size_t)v * D + d] = acc;
}
