// Direct solves through the polynomial form (round 6).
//
// When every top row of the grid operator is in the polynomial-subspace form
// (rl_lowrank.h) the SKI operator of reference approx/ski.py:13-16 is, to the 2e-13 the
// form is verified to at set time,
//
//     K~ = F M F^T + E,     F = W Phi  block-diagonal by output (n x D r; rl_rowpoly.h),
//                           M = sum_q B_q (x) C_q  (D r x D r),   E = diag(eps_d) per output
//
// -- a diagonal plus rank D r (240 at BASELINE's C5).  With G = F^T E^-1 F (block-diagonal:
// G_d = F_d^T F_d / eps_d, the Gram matrices F_d^T F_d built ONCE per handle and rank) and
// G = L L^T,  S = I + L^T M L  (symmetric positive definite, D r x D r, host Cholesky):
//
//     K~^-1 b = E^-1 b - E^-1 F Z F^T E^-1 b,     Z = L^-T (I - S^-1) L^-1
//     log det K~ = sum_d n_d log eps_d + log det S                        (exactly)
//
// i.e. ONE projection (k_rp_project), one dense D r x D r map on the coefficients
// (k_dz_mix below) and ONE expansion (k_rp_expand with 1/eps as its diagonal) per batch,
// where the Krylov solve of reference approx/iterative.py:23-62 runs hundreds of rounds
// and, at C5's conditioning, never reaches the reference's own 1e-4 residual rule in fp64.
// The reference's hook for exactly this is its preconditioner argument
// (iterative.py:47-51: M = getattr(K, 'preconditioner', None)); with M = K~^-1 to roundoff
// a preconditioned iteration is iterative refinement: x += M (b - K~ x) until the
// reference's rule ||b - K~ x||_2 < tol holds (host loop: rl_solve.hip, rl_solve_direct).
//
// Scalings: the streaming kernels work with the UNNORMALISED polynomials q_j (Phi_j = nu_j
// q_j), so the host folds nu and 1/eps into the map it uploads:
//     Zs[(a,i)][(b,j)] = -(nu_i / eps_a) Z[(a,i)][(b,j)] (nu_j / eps_b)
//     x = (1/eps) (.) b + F_q (Zs (F_q^T b))
#pragma once
#include "rl_device.h"

#define RL_DZ_VB 4            // vectors per k_dz_mix workgroup (the map's rows are read once for them)

// ---------------------------------------------------------------------------
// k_dz_mix: zhat[v][e] = sum_f Zt[f][e] S[v][f],   S[v][(b, j)] = sum_{runs c of output b}
// part[c][v][j]  (k_rp_project's partial sums, ascending run order as in k_lr_mix).
//   grid (ceil(nvec / RL_DZ_VB))   block 256   LDS: S [RL_DZ_VB][D r]
// Zt is the map TRANSPOSED (it is symmetric; the layout only says that consecutive
// threads read consecutive addresses).  A thread owns output coefficients e, e + 256, ...
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
k_dz_mix(const double* __restrict__ part, const int* __restrict__ run_ptr, int nvec, int D, int r,
         const double* __restrict__ Zt, double* __restrict__ zhat) {
    RL_SMEM(smem);
    double* S = reinterpret_cast<double*>(smem);          // [RL_DZ_VB][Dr]
    const int tid = threadIdx.x, Dr = D * r;
    const int v0 = blockIdx.x * RL_DZ_VB;
    for (int e = tid; e < Dr; e += 256) {
        const int b = e / r, j = e - b * r;
        const int c0 = run_ptr[b], c1 = run_ptr[b + 1];
#pragma unroll
        for (int k = 0; k < RL_DZ_VB; ++k) {
            const int v = v0 + k < nvec ? v0 + k : nvec - 1;
            const double* src = part + (size_t)v * r + j;
            double s = 0.0;
            for (int c = c0; c < c1; ++c) s += src[(size_t)c * nvec * r];
            S[k * Dr + e] = s;
        }
    }
    __syncthreads();
    for (int e = tid; e < Dr; e += 256) {
        double acc[RL_DZ_VB];
#pragma unroll
        for (int k = 0; k < RL_DZ_VB; ++k) acc[k] = 0.0;
        for (int f = 0; f < Dr; ++f) {
            const double z = Zt[(size_t)f * Dr + e];
#pragma unroll
            for (int k = 0; k < RL_DZ_VB; ++k) acc[k] = fma(z, S[k * Dr + f], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < RL_DZ_VB; ++k)
            if (v0 + k < nvec) zhat[(size_t)(v0 + k) * Dr + e] = acc[k];
    }
}

// ---------------------------------------------------------------------------
// k_dz_resid: R[v][i] = B[v][i] - R[v][i]  (R holds K~ x on entry), partial[v][blk] = the
// block's sum of squares (fixed order).   grid (nblk, nrhs)   block 256
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
k_dz_resid(const double* __restrict__ B, double* __restrict__ R, int n,
           double* __restrict__ partial) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    const double* pb = B + (size_t)rhs * n;
    double* pr = R + (size_t)rhs * n;
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        const double d = pb[i] - pr[i];
        pr[i] = d;
        acc = fma(d, d, acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)rhs * gridDim.x + blockIdx.x] = red[0];
}

// resid[v] = sqrt(sum of the system's partials)   grid (ceil(nrhs / 64))   block 64
static __global__ void __launch_bounds__(64)
k_dz_norms(const double* __restrict__ partial, int nblk, int nrhs, double* __restrict__ resid) {
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= nrhs) return;
    double s = 0.0;
    for (int c = 0; c < nblk; ++c) s += partial[(size_t)v * nblk + c];
    resid[v] = sqrt(s);
}

// ---------------------------------------------------------------------------
// k_dz_axpy: X[v] += T[v] for the systems still being refined (go[v] != 0).
//   grid (nblk, nrhs)   block 256
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
k_dz_axpy(double* __restrict__ X, const double* __restrict__ T, int n,
          const int* __restrict__ go) {
    const int rhs = blockIdx.y;
    if (!go[rhs]) return;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double* px = X + (size_t)rhs * n;
    const double* pt = T + (size_t)rhs * n;
    for (int i = lo + threadIdx.x; i < hi; i += 256) px[i] += pt[i];
}

// ---------------------------------------------------------------------------
// k_dz_scale: Y[v][i] = d[i] X[v][i]   (rows scaled by a per-row vector: the 1 / sqrt(eps) in front
// of the preconditioner's square-root factor, rl_ski_precond_sample).   grid (nblk, nvec)   block 256
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
k_dz_scale(const double* __restrict__ X, const double* __restrict__ d, int n, double* __restrict__ Y) {
    const int v = blockIdx.y;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    const size_t off = (size_t)v * n;
    for (int i = lo + threadIdx.x; i < hi; i += 256) Y[off + i] = d[i] * X[off + i];
}

// ---------------------------------------------------------------------------
// k_dz_coeffs: out[v][b][j] = nu_j sum_{runs c of output b} part[c][v][j] -- the coefficients
// Phi^T W^T x of a batch on the NORMALISED basis from k_rp_project's partial sums (ascending run
// order).   grid (nvec)   block 256
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
k_dz_coeffs(const double* __restrict__ part, const int* __restrict__ run_ptr, int nvec, int D, int r,
            const double* __restrict__ nu, double* __restrict__ out) {
    const int v = blockIdx.x, Dr = D * r;
    for (int e = threadIdx.x; e < Dr; e += 256) {
        const int b = e / r, j = e - b * r;
        const double* src = part + (size_t)v * r + j;
        double s = 0.0;
        for (int c = run_ptr[b]; c < run_ptr[b + 1]; ++c) s += src[(size_t)c * nvec * r];
        out[(size_t)v * Dr + e] = nu[j] * s;
    }
}

// ---------------------------------------------------------------------------
// Preconditioned conjugate gradients with M = (F M_r F^T + E)^-1, the Woodbury inverse of the
// operator's projection on the polynomial subspace (round 6): for operators that are NOT wholly
// in the polynomial form (a Matern row next to smooth ones, Matern rows alone) the factorisation
// of rl_solve.hip is no longer K~^-1 but still the reference's preconditioner argument
// (approx/iterative.py:47-51, sla.cg(op, y, M=M)): SciPy's statements of preconditioned CG --
//   z = M r;  rho = r.z;  p = z + (rho / rho_prev) p;  q = K~ p;  alpha = rho / p.q;
//   x += alpha p;  r -= alpha q
// -- all systems in lockstep, each with its own scalars scal[v] = (rho, rho_prev); go[v] = 0
// freezes a system that met the reference's residual rule.  Dot products: k_dot_partial
// (rl_solver.h) into per-block partial sums, summed in a fixed order here.
//   grid (nblk, nrhs) block 256 for the vector kernels; k_pcg_head: grid (ceil(nrhs / 64)) block 64
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(64)
k_pcg_head(const double* __restrict__ partRZ, int nblk, int nrhs, double* __restrict__ scal,
           const int* __restrict__ go) {
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= nrhs || !go[v]) return;
    double s = 0.0;
    for (int c = 0; c < nblk; ++c) s += partRZ[(size_t)v * nblk + c];
    scal[2 * v + 1] = scal[2 * v];
    scal[2 * v] = s;
}
static __global__ void __launch_bounds__(256)
k_pcg_p(double* __restrict__ p, const double* __restrict__ z, int n, const double* __restrict__ scal,
        const int* __restrict__ go, int first) {
    const int rhs = blockIdx.y;
    if (!go[rhs]) return;
    const double beta = first ? 0.0 : scal[2 * rhs] / scal[2 * rhs + 1];
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    const size_t off = (size_t)rhs * n;
    for (int i = lo + threadIdx.x; i < hi; i += 256)
        p[off + i] = first ? z[off + i] : fma(beta, p[off + i], z[off + i]);
}
static __global__ void __launch_bounds__(256)
k_pcg_update(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
             const double* __restrict__ q, int n, const double* __restrict__ scal,
             const double* __restrict__ partPQ, double* __restrict__ partRR,
             const int* __restrict__ go, double* __restrict__ rec = nullptr) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const int rhs = blockIdx.y, nblk = gridDim.x;
    if (!go[rhs]) return;
    double pq = 0.0;
    for (int c = 0; c < nblk; ++c) pq += partPQ[(size_t)rhs * nblk + c];
    const double alpha = scal[2 * rhs] / pq;
    // (rec: this iteration's row of the recurrence's scalars, [nrhs][2] = (rho, p.q) -- the
    // Lanczos matrix of the preconditioned operator is made of them, rl_solve_pcg_lanczos)
    if (rec != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        rec[2 * rhs] = scal[2 * rhs];
        rec[2 * rhs + 1] = pq;
    }
    const int per = (n + nblk - 1) / nblk;
    const int lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    const size_t off = (size_t)rhs * n;
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        x[off + i] = fma(alpha, p[off + i], x[off + i]);
        const double ri = fma(-alpha, q[off + i], r[off + i]);
        r[off + i] = ri;
        acc = fma(ri, ri, acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) partRR[(size_t)rhs * nblk + blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------------
// The dense map of a factorisation whose basis is NB blocks of RL_HZ_BLK functions (the larger
// preconditioner, rl_solve.hip: hz_*): the projection ran block by block (k_rp_project<48> on 48
// columns of F at a time), the map is up to 2048 x 2048 -- one workgroup per four vectors walking all
// of it (k_dz_mix's scheme: 0.94 ms at C5's 129 x 1920, a quarter of the chip busy) became three
// launches, ~10x less:
//   k_hz_sums:    S[v][(b, 48 k + j)] = sum_{runs c of output b} part[k][c][v][j]
//   k_hz_map:     P[s][v][e] = sum_{f in slice s} Zt[f][e] S[v][f]       (RL_HZ_FS slices of f)
//   k_hz_collect: zhat[k][v][a][i] = sum_s P[s][v][(a, 48 k + i)]        (fixed order: no atomics)
// each block's coefficients a standalone [nvec][D][48] array for its expansion.
// ---------------------------------------------------------------------------
#define RL_HZ_BLK 48
#define RL_HZ_VB 8             // vectors per k_hz_map workgroup
#define RL_HZ_FS 4             // slices of the contraction index
#define RL_HZ_FMAX 512         // longest slice (2048 / RL_HZ_FS): S rows in LDS, [f][RL_HZ_VB]
//   grid (ceil(D R / 256), nvec)   block 256
static __global__ void __launch_bounds__(256)
k_hz_sums(const double* __restrict__ part, const int* __restrict__ run_ptr, int nruns, int nvec,
          int D, int NB, double* __restrict__ S) {
    const int R = NB * RL_HZ_BLK, Dr = D * R;
    const int e = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    if (e >= Dr) return;
    const int b = e / R, jg = e - b * R, k = jg / RL_HZ_BLK, j = jg - k * RL_HZ_BLK;
    const double* src = part + ((size_t)k * nruns * nvec + v) * RL_HZ_BLK + j;
    double s = 0.0;
    for (int c = run_ptr[b]; c < run_ptr[b + 1]; ++c) s += src[(size_t)c * nvec * RL_HZ_BLK];
    S[(size_t)v * Dr + e] = s;
}
//   grid (ceil(D R / 256), ceil(nvec / RL_HZ_VB), RL_HZ_FS)   block 256
//   LDS: [slice length][RL_HZ_VB] doubles.  A thread owns one output coefficient e of RL_HZ_VB
//   vectors; Zt rows are read along e (the map is symmetric: Zt = Z).
static __global__ void __launch_bounds__(256)
k_hz_map(const double* __restrict__ S, const double* __restrict__ Zt, int nvec, int Dr,
         double* __restrict__ P) {
    RL_SMEM(smem);
    double* Sl = reinterpret_cast<double*>(smem);         // [flen][RL_HZ_VB]
    const int tid = threadIdx.x, e = blockIdx.x * 256 + tid, v0 = blockIdx.y * RL_HZ_VB;
    const int per = (Dr + gridDim.z - 1) / gridDim.z;
    const int f0 = blockIdx.z * per, f1 = f0 + per < Dr ? f0 + per : Dr, flen = f1 > f0 ? f1 - f0 : 0;
    for (int t = tid; t < flen * RL_HZ_VB; t += 256) {
        const int q = t / flen, f = t - q * flen;             // (consecutive threads: consecutive f)
        const int v = v0 + q < nvec ? v0 + q : nvec - 1;
        Sl[f * RL_HZ_VB + q] = S[(size_t)v * Dr + f0 + f];
    }
    __syncthreads();
    if (e >= Dr) return;
    double acc[RL_HZ_VB];
#pragma unroll
    for (int q = 0; q < RL_HZ_VB; ++q) acc[q] = 0.0;
    const double* z = Zt + (size_t)f0 * Dr + e;
#pragma unroll 4
    for (int f = 0; f < flen; ++f) {
        const double zf = z[(size_t)f * Dr];
#pragma unroll
        for (int q = 0; q < RL_HZ_VB; ++q) acc[q] = fma(zf, Sl[f * RL_HZ_VB + q], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < RL_HZ_VB; ++q)
        if (v0 + q < nvec) P[((size_t)blockIdx.z * nvec + v0 + q) * Dr + e] = acc[q];
}
//   grid (ceil(D R / 256), nvec)   block 256
// (nvp > 0: the layout of k_hz_expand_mm at the end of this file instead, zT[a][48 k + i][v] with rows
// of nvp vectors, entries of vectors nvec .. nvp - 1 zero)
static __global__ void __launch_bounds__(256)
k_hz_collect(const double* __restrict__ P, int nvec, int D, int NB, int slices,
             double* __restrict__ zhat, int nvp = 0) {
    const int R = NB * RL_HZ_BLK, Dr = D * R;
    const int e = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    if (e >= Dr) return;
    if (nvp > 0) {
        double s = 0.0;
        if (v < nvec)
            for (int t = 0; t < slices; ++t) s += P[((size_t)t * nvec + v) * Dr + e];
        zhat[(size_t)e * nvp + v] = s;            // (e = a R + degree)
        return;
    }
    double s = 0.0;
    for (int t = 0; t < slices; ++t) s += P[((size_t)t * nvec + v) * Dr + e];
    const int a = e / R, ig = e - a * R, k = ig / RL_HZ_BLK, i = ig - k * RL_HZ_BLK;
    zhat[(((size_t)k * nvec + v) * D + a) * RL_HZ_BLK + i] = s;
}

// ---------------------------------------------------------------------------
// k_hz_expand_mm<NVT>: out[v][i] = diag[i] X2[v][i] + sum_{j < R} F[j][i] zT[d(i)][j][v] for ALL
// R = 48 NB functions of the larger basis in one pass over the rows, on the fp64 matrix cores.
// The rank-48 expansion keeps 48 values of F per row in registers and streams the vectors, so R
// functions took R / 48 passes, each reading the previous one's output and writing its own (four
// passes at C5: 2.55 ms an application, 9.8 GB).  As a product  D(vectors x rows) += A(vectors x
// degrees) B(degrees x rows)  the accumulators of 16 vectors x 16 rows are four registers a lane,
// so a wave holds ALL the batch's vectors for 16 RT rows and walks the degrees once:
//   A[v][k] = zT[d][j0 + k][v0 + v]   (lane v + 16 k; the coefficients transposed, [D][R][nvp],
//             nvp = nvec rounded up to 16: 221 KB per output at C5 -- L2)
//   B[k][i] = F[j0 + k][i0 + i]       (lane i + 16 k: 16 consecutive rows of the degree-major table)
//   D: register r of lane l = vector (l >> 4) + 4 r, row l & 15 -- loads of X2 and stores of out are
//      16 consecutive rows of one vector per quarter wave.
// (Operand layouts: rl_rowpoly.h.)  A 16-row tile that straddles outputs runs the degrees once per
// output with the other outputs' rows of B zeroed.  Vector tiles beyond nvp / 16 are skipped.
//   grid (ceil(n / (64 RT)), ceil(nvp / (16 NVT)))   block 256: wave w owns 16 RT rows (RT = 2: 144
//   accumulation registers, two waves a SIMD, 1.83 ms at C5; RT = 3: 216, one wave, 2.73 ms with one
//   step of request-ahead where RT = 2 took 2.05)
// (emulator: a thread owns a row and sums the degrees itself)
// ---------------------------------------------------------------------------
template <int NVT, int RT = 2>
__global__ void __launch_bounds__(256)
k_hz_expand_mm(const double* __restrict__ zT, const double* __restrict__ F, int n, int nvec, int nvp,
               int D, int R, const int* __restrict__ out_end, double* __restrict__ out,
               const double* __restrict__ diag, const double* __restrict__ X2) {
    const int tid = threadIdx.x;
    const int vt0 = blockIdx.y * NVT, nvt = nvp / 16 - vt0 < NVT ? nvp / 16 - vt0 : NVT;
    constexpr int WR = 16 * RT;                     // rows of a wave
#if !defined(RL_EMU)
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int r0 = blockIdx.x * (4 * WR) + wave * WR;
    if (r0 >= n) return;
    rp_double4 C[RT][NVT];
    // C = diag (.) X2
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int i = r0 + 16 * rt + li;
        const double dg = i < n ? diag[i] : 0.0;
#pragma unroll
        for (int t = 0; t < NVT; ++t) {
            rp_double4 c = rp_double4{0.0, 0.0, 0.0, 0.0};
            if (t < nvt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int v = (vt0 + t) * 16 + lk + 4 * r;
                    c[r] = i < n && v < nvec ? dg * X2[(size_t)v * n + i] : 0.0;
                }
            }
            C[rt][t] = c;
        }
    }
    const int rlast = r0 + WR - 1 < n ? r0 + WR - 1 : n - 1;
    const int dlo = RL_LR_UNIFORM(rp_output_of(out_end, D, r0));
    const int dhi = RL_LR_UNIFORM(rp_output_of(out_end, D, rlast));
    int irow[RT], drow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        irow[rt] = r0 + 16 * rt + li;
        drow[rt] = dlo == dhi ? dlo : rp_output_of(out_end, D, irow[rt] < n ? irow[rt] : n - 1);
    }
    for (int d = dlo; d <= dhi; ++d) {
        bool on[RT];
        const double* fr[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            on[rt] = irow[rt] < n && drow[rt] == d;
            fr[rt] = F + (size_t)lk * n + (irow[rt] < n ? irow[rt] : n - 1);
        }
        const double* za = zT + ((size_t)d * R + lk) * nvp + (size_t)vt0 * 16 + li;
        // (operands of the NEXT four degrees are requested before this step's RT NVT matrix
        // instructions are issued: without that every step waited for its own loads -- 3.05 ms an
        // application against the four passes' 2.55)
        // a ring of three operand sets: step s computes on set s % 3 while the loads of steps s + 1
        // and s + 2 are in flight (R / 4 is a multiple of 3 for every basis size: 24, 36, 48 steps)
        double a[3][NVT], b[3][RT];
        auto request = [&](int slot, int j) {
            const int jc = j < R ? j : R - 4;
            const double* zj = za + (size_t)jc * nvp;
#pragma unroll
            for (int t = 0; t < NVT; ++t) a[slot][t] = t < nvt ? zj[16 * t] : 0.0;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) b[slot][rt] = on[rt] ? fr[rt][(size_t)jc * n] : 0.0;
        };
        request(0, 0);
        request(1, 4);
        for (int j0 = 0; j0 < R; j0 += 12) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                request((u + 2) % 3, j0 + 4 * u + 8);
#pragma unroll
                for (int t = 0; t < NVT; ++t) {
                    if (t < nvt) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            C[rt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][t], b[u][rt], C[rt][t], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int i = r0 + 16 * rt + li;
#pragma unroll
        for (int t = 0; t < NVT; ++t) {
            if (t < nvt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int v = (vt0 + t) * 16 + lk + 4 * r;
                    if (i < n && v < nvec) out[(size_t)v * n + i] = C[rt][t][r];
                }
            }
        }
    }
#else
    if (tid >= 4 * WR) return;
    const int i = blockIdx.x * (4 * WR) + tid;
    if (i >= n) return;
    const int d = rp_output_of(out_end, D, i);
    for (int t = 0; t < nvt; ++t)
        for (int q = 0; q < 16; ++q) {
            const int v = (vt0 + t) * 16 + q;
            if (v >= nvec) continue;
            double acc = diag[i] * X2[(size_t)v * n + i];
            for (int j = 0; j < R; ++j) acc = fma(F[(size_t)j * n + i], zT[((size_t)d * R + j) * nvp + v], acc);
            out[(size_t)v * n + i] = acc;
        }
#endif
}
