/* runlmc_hip.h -- C ABI of the MI355X-native LMC inference hot path.
 *
 * Drop-in boundary for vlad17/runlmc's matrix-free path.  Every entry point
 * names the reference interface it replaces (file:line in the reference
 * tree).  INTEGRATION.md shows the ctypes binding a runlmc maintainer adds.
 *
 * Conventions
 *   - every function returns 0 on success, nonzero on failure;
 *     rl_last_error() then holds a message (thread-local).  Shape / argument
 *     errors are RL_EINVAL (the Python wrapper raises ValueError, as the
 *     reference does: runlmc/linalg/matrix.py:20-21, bttb.py:93-101).
 *   - "host" pointers are ordinary CPU memory, copied during the call.
 *   - "dev" pointers are device memory owned by the caller (e.g.
 *     torch.Tensor.data_ptr()); fp64, contiguous, vectors stored one after
 *     another: X[v][i].  Grid vectors are output-major, index d*m + i
 *     (reference kronecker.py:42-46); data vectors are the concatenation of
 *     the outputs (reference likelihood.py:30-31).
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls enqueue work
 *     and return; rl_solve_batch / rl_*_host helpers synchronise themselves.
 *   - handles are not thread-safe; one handle per device.
 */
#ifndef RUNLMC_HIP_H
#define RUNLMC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RL_OK 0
#define RL_EINVAL 1   /* bad argument / shape */
#define RL_EHIP 2     /* HIP runtime error */
#define RL_ENOMEM 3
#define RL_ELIMIT 4   /* problem exceeds a documented kernel limit */

typedef struct rl_gridop rl_gridop;
typedef struct rl_ski rl_ski;

/* Version of this ABI: bumped whenever a declared signature changes (2: rl_solve_batch_lanczos
 * gained `method`, round 4; 3: rl_gridop_form_stats added, round 5; 4: rl_ski_factor,
 * rl_solve_direct, rl_solve_pcg, rl_ski_project,
 * rl_gridop_project, rl_gridop_set_rank_hint, rl_gridop_poly_coeffs, rl_slq_log_quadrature,
 * rl_probes_to_int8 added, round 6; callers built against an older version must be rebuilt).  A binding
 * compares rl_abi_version() with the RL_ABI_VERSION it was written against before its
 * first call (runlmc_amd/_lib.py does) instead of finding out through shifted arguments. */
#define RL_ABI_VERSION 4
int rl_abi_version(void);

const char* rl_last_error(void);
/* "hip-gfx950" for the product library. */
const char* rl_backend(void);
int rl_device_count(int* count);

/* ---- grid operator  K_UU = sum_q B_q (x) T_q  ------------------------------
 * Replaces BTTB.__init__/_cyclic_extend_n (runlmc/linalg/bttb.py:91-120),
 * BTTB.matvec (bttb.py:144-148), Kronecker.matvec (kronecker.py:39-46),
 * SumMatrix.matvec (sum_matrix.py:31-32) and the three grid representations
 * _gen_sum_grid/_gen_bt_grid/_gen_slfm_grid (runlmc/lmc/grid_kernel.py:77-136),
 * which are the same linear operator.
 *   D outputs, m grid points (1-D grid), embedding length L = the smallest
 *   odd * 2^k >= 2m with odd in {1, 3, 5, 9, 15, 25} (any length >= 2m - 1
 *   embeds the Toeplitz matrix exactly; the reference takes the next power
 *   of two, bttb.py:16-19).  max_tops bounds Q in later rl_gridop_set_* calls.
 *   D is not limited (the reference's Kronecker has no limit, kronecker.py:39-46):
 *   up to 16 outputs the D x D mix runs inside the product's kernels; above, the
 *   handle applies T_q to the nvec * D rows through a one-output handle of its own
 *   and mixes the outputs in a pass of its own (Q products + Q mix passes).      */
int rl_gridop_create(int device, int D, int m, int max_tops, rl_gridop** out);
/* Same for a two-dimensional m1 x m2 grid: T_q is then a block-Toeplitz matrix
 * of Toeplitz blocks, BTTB(top, (m1, m2)) (bttb.py:91-148 with two sizes; grid
 * index i1*m2 + i2, top rows are k_q of the distances to grid point (0, 0)).
 * The embedding is an N1 x N2 two-dimensional circulant, N_k = pow2 >= 2 m_k.
 * Every other entry point is unchanged (m = m1*m2 points per output).        */
int rl_gridop_create_2d(int device, int D, int m1, int m2, int max_tops, rl_gridop** out);
int rl_gridop_destroy(rl_gridop* g);
/* L = N1*N2 and tile parameters actually chosen (any pointer may be NULL). */
int rl_gridop_info(const rl_gridop* g, int* L, int* N1, int* N2, int* colsA, int* rowsB);
/* Which form of the product the CURRENT parameters run in (valid after a
 * rl_gridop_set_* call; any pointer may be NULL):
 *   *rank          0: transform (FFT) kernels only;  r > 0: every top row is,
 *                  to 2e-13 of its products, Phi C_q Phi^T with Phi the r
 *                  orthonormal polynomials on the grid, and batches of at
 *                  least *min_elements = nvec*D*m elements run as
 *                  project -> r x r map -> expand (csrc/rl_lowrank.h) -- the
 *                  same operator of bttb.py:144-148, verified per top row
 *                  against the transform kernels inside the set call.
 *   RUNLMC_NO_LOWRANK=1 keeps every handle on the transform kernels (an A/B
 *   hook like RUNLMC_NO_FILTER and RUNLMC_LR_MIN below: read ONLY when
 *   RUNLMC_DEBUG=1 is set as well -- without it the library runs its defaults and
 *   names every hook it ignored once on stderr; the switches that are always
 *   read are RUNLMC_POW2_ONLY, RUNLMC_WS_CACHE_MB and RUNLMC_TRACE).           */
int rl_gridop_form(const rl_gridop* g, int* rank, long long* min_elements);
/* The form EACH top row runs in for the current parameters (forms host [Q], may
 * be NULL): 0 transform kernels; 1 polynomial-subspace form; 2 recursive
 * filter -- an exponential-polynomial row t_i = (c0 + c1 i + c2 i^2) rho^i, i.e.
 * the reference's Matern-3/2 kernel on a regular grid and its derivative
 * (runlmc/kern/matern32.py:40-55), is exactly semiseparable and its product a
 * two-sided scan (csrc/rl_filter.h), detected from the row itself.
 * *structured = 1 when no top needs the transform kernels, so that operator
 * products of batches above the gate move x and y and nothing else.  Single-top
 * products (rl_gridop_mvm_top) take each top's own form either way.
 * RUNLMC_NO_FILTER=1 (with RUNLMC_DEBUG=1) switches form 2 off.                  */
int rl_gridop_top_forms(const rl_gridop* g, int* forms, int* structured);
/* What the last set-time verification of the polynomial form measured for top row q
 * (host out4; zeros when the row was never a candidate): [0] max|T x - Phi C Phi^T x| /
 * max|T x| for the fixed trial vector, [1] the largest response to the first omitted
 * polynomials relative to the largest response inside the subspace, [2] the power
 * iteration's estimate of ||T - Phi C Phi^T||_2 (8 steps through the two products of this
 * handle, from the unit-norm trial vector), [3] the same iteration's ||T||_2.  The row is
 * accepted when [0], [1] <= 2e-13 and [2] <= 2e-13 [3].  [2] and [3] are ESTIMATES FROM
 * BELOW (a power iteration climbs towards the norm; after 8 steps from a vector with a
 * ~1/sqrt(m) component along the leading eigenvector it holds roughly half of it or
 * more), not upper bounds: the test catches an error the sampled tests miss in whatever
 * direction it lies, with a margin of three decades to the honest rows' 2e-16, it does
 * not prove ||E||_2 <= 2e-13 ||T||_2.
 * (Diagnostics of this library's own forms; the operator is bttb.py:144-148 either way.) */
int rl_gridop_form_stats(const rl_gridop* g, int q, double* out4);
/* Moves that batch gate for this handle (0: every batch; < 0: back to the
 * default, 2^20 elements or RUNLMC_LR_MIN under RUNLMC_DEBUG=1).  Below the gate the polynomial
 * form is slower than the transform kernels (too few workgroups).           */
int rl_gridop_set_form_gate(rl_gridop* g, long long min_elements);
/* Parameters of the LMC kernel, in the reference's own factored form
 * B_q = A_q^T A_q + diag(kappa_q) (runlmc/lmc/functional_kernel.py:280-287):
 *   tops        host [Q][m]   first rows k_q(grid distances)
 *   ranks       host [Q]      R_q >= 0
 *   coreg_vecs  host [sum R_q][D]   rows of A_0, then A_1, ...
 *   coreg_diags host [Q][D]
 * Rebuilds the Q circulant spectra on the device (per optimiser step).      */
int rl_gridop_set_lmc(rl_gridop* g, int Q, const double* tops, const int* ranks,
                      const double* coreg_vecs, const double* coreg_diags);
/* Same with arbitrary symmetric B_q given densely, host [Q][D][D]
 * (Kronecker(NumpyMatrix(B), BTTB(top)), kronecker.py:30-46).               */
int rl_gridop_set_dense(rl_gridop* g, int Q, const double* tops, const double* B);

/* Y[v] = K_UU X[v], v < nvec.  X, Y dev [nvec][D*m]; may not alias.         */
int rl_gridop_mvm(rl_gridop* g, const double* X, double* Y, int nvec, void* stream);
/* Y[v] = (I_D (x) T_q) X[v]: one top row applied to every output block
 * (BTTB.matmat, matrix.py:55-67 + bttb.py:144-148).                          */
int rl_gridop_mvm_top(rl_gridop* g, int q, const double* X, double* Y, int nvec, void* stream);
/* Circulant spectrum of top q in NATURAL frequency order, host out[L]
 * (test hook; the reference's BTTB._circ_fft real part, bttb.py:108).       */
int rl_gridop_spectrum_host(rl_gridop* g, int q, double* out);

/* ---- SKI operator  K~ = W K_UU W^T + diag(eps)  ----------------------------
 * Replaces SKI / Composition.matvec (runlmc/approx/ski.py:8-16,
 * linalg/composition.py:14-17), Diag.matvec (linalg/diag.py:24-25) and the
 * SumMatrix assembled by gen_grid_kernel (lmc/grid_kernel.py:70-74).
 *   W   CSR n x (D*m)  (host; int32 indices; reference
 *       approx/interpolation.py:119-176), WT its transpose in CSR.
 * The handle keeps a pointer to `g` (not owned): every product and solve uses
 * it, so destroy the SKI handle BEFORE its grid operator(s).  (rl_ski_destroy
 * itself does not touch `g`, so the other order only leaks nothing and crashes
 * nothing -- but no other call on the handle is valid once `g` is gone.)     */
int rl_ski_create(rl_gridop* g, int n, const int* W_indptr, const int* W_indices,
                  const double* W_data, const int* WT_indptr, const int* WT_indices,
                  const double* WT_data, rl_ski** out);
int rl_ski_destroy(rl_ski* s);
/* Kernels split over several active-dimension sets live on several grids:
 * K~ = sum_t W_t K_t W_t^T + diag(eps) (the SumMatrix of one GridKernel per
 * active-dimension set that gen_grid_kernel builds, grid_kernel.py:51-74).
 * Adds term t >= 1: its grid operator (same D, same device, not owned) and its
 * interpolant pair, n x (D*m_t).  Terms are numbered in the order added.     */
int rl_ski_add_term(rl_ski* s, rl_gridop* g, const int* W_indptr, const int* W_indices,
                    const double* W_data, const int* WT_indptr, const int* WT_indices,
                    const double* WT_data);
/* noise host [D], lens host [D] (sum lens == n): eps repeated per output
 * (np.repeat(noise, lens), grid_kernel.py:70).                               */
int rl_ski_set_noise(rl_ski* s, const double* noise, const int* lens);
/* Y[v] = K~ X[v];  X, Y dev [nvec][n]; may not alias.                        */
int rl_ski_mvm(rl_ski* s, const double* X, double* Y, int nvec, void* stream);
/* G[v] = W^T X[v] (dev [nvec][D*m])  /  Y[v] = W G[v] (dev [nvec][n]).      */
int rl_ski_apply_wt(rl_ski* s, const double* X, double* G, int nvec, void* stream);
int rl_ski_apply_w(rl_ski* s, const double* G, double* Y, int nvec, void* stream);
/* The same with the interpolants of term `term` (0 = the first).             */
int rl_ski_apply_wt_term(rl_ski* s, int term, const double* X, double* G, int nvec, void* stream);
int rl_ski_apply_w_term(rl_ski* s, int term, const double* G, double* Y, int nvec, void* stream);

/* ---- batched Krylov solves  K~ X = B  --------------------------------------
 * Replaces Iterative.solve (runlmc/approx/iterative.py:23-62) and the N+1
 * pool-mapped solves of StochasticDerivService._concurrent_solve
 * (runlmc/lmc/stochastic_deriv.py:39-52).  All right-hand sides advance
 * together, each with its own recurrence scalars and stopping state.
 *   B, X      dev [nrhs][n]   (X is written; initial guess is 0 as in the
 *                              reference)
 *   method    RL_MINRES (reference default, stochastic_deriv.py:37), RL_CG, or
 *             RL_MINRES_RULE: MINRES with SciPy's own stopping tests (istop 1-4:
 *             test1 / test2 / Acond / epsx) switched off, so that a system ends
 *             only on the reference's explicit residual rule below or at
 *             maxiter -- how the reference's PUBLISHED runs ended (iteration
 *             counts are multiples of 100 and residuals < 1e-4 in
 *             benchmarks/representation-cmp/out/inv-run-1.txt:3-20; SciPy
 *             1.15's test1 <= rtol exit fires long before on the same systems,
 *             DESIGN.md section 5).  Needs check_every > 0.
 *   tol       absolute residual target of the reference's rule; the inner
 *             method runs with rtol = min(1e-10, tol) (iterative.py:50-51)
 *   check_every  explicit-residual check period (reference: 100,
 *             iterative.py:39); a system whose ||b - K~x||_2 < tol at a check
 *             is frozen; 0 disables the rule
 *   maxiter   <= 0 means n (iterative.py:51)
 *   iters_out host [nrhs] iterations run (the reference's callback count),
 *   resid_out host [nrhs] final ||b - K~x||_2 (iterative.py:54),
 *   istop_out host [nrhs] exit reason: SciPy minres istop codes 1..6, -1;
 *             10 = reference residual rule; 11 = zero right-hand side.
 * Any of the three output pointers may be NULL.  Synchronises before it
 * returns.  Non-convergence is NOT an error (the reference logs and returns
 * the iterate, iterative.py:55-58).  The solver's work vectors stay on the
 * handle between calls (freed by rl_ski_destroy; environment variable
 * RUNLMC_WS_CACHE_MB bounds what is kept).                                     */
#define RL_MINRES 0
#define RL_CG 1
#define RL_MINRES_RULE 2
int rl_solve_batch(rl_ski* s, const double* B, double* X, int nrhs, int method, double tol,
                   int check_every, int maxiter, int* iters_out, double* resid_out,
                   int* istop_out, void* stream);

/* Same solve, additionally returning the Lanczos tridiagonal MINRES builds for
 * each system: lanczos_out host [nrhs][lanczos_cap][2] = (alfa_k, beta_{k+1})
 * for k = 1..min(iterations, lanczos_cap) (untouched beyond).  With
 * Rademacher right-hand sides these give a stochastic-Lanczos-quadrature
 * estimate of log det K~ at no extra operator products -- the matrix-free
 * log-determinant the reference lists as future work (README.md:88-89; its
 * own log_det_K is a dense Cholesky, models/interpolated_llgp.py:262-276).
 * method: RL_MINRES or RL_MINRES_RULE; lanczos_out may be NULL (then identical to
 * rl_solve_batch). */
int rl_solve_batch_lanczos(rl_ski* s, const double* B, double* X, int nrhs, int method,
                           double tol, int check_every, int maxiter, int* iters_out,
                           double* resid_out, int* istop_out, double* lanczos_out,
                           int lanczos_cap, void* stream);

/* Host helper for that estimate: out[v] = sqnorms[v] * e_1^T log(T_v) e_1 for the Lanczos
 * tridiagonal T_v of system v (lanczos host [nrhs][cap][2] as rl_solve_batch_lanczos fills
 * it, k = min(iters[v], cap) steps) -- Gauss quadrature of r^T log(K~) r by the implicit QL
 * iteration carrying one row of the eigenvector matrix, O(k^2) per system, systems spread
 * over `nthreads` host threads; out[v] = NaN for a system whose iteration did not settle
 * (the caller falls back to LAPACK for it).  No device work.  (The mean of out over Rademacher probes,
 * sqnorms = n, estimates log det K~.)                                                      */
int rl_slq_log_quadrature(const double* lanczos, int nrhs, int cap, const int* iters,
                          const double* sqnorms, double* out, int nthreads);
/* Host helper: the reference draws its Rademacher probes as an int64 matrix
 * (lmc/stochastic_deriv.py:35: randint(0, 2, (N, n)) * 2 - 1 -- 1 GB at BASELINE's C5).  One
 * pass over it on `nthreads` host threads: dst[r][k] = (int8) src[r * row_stride + k] for
 * nrows rows of n entries, *all_pm1 = 1 iff every entry is +1 or -1 (else the caller sends the
 * matrix the plain way).  dst may be pinned memory; no device work.                        */
int rl_probes_to_int8(const long long* src, int nrows, long long row_stride, long long n,
                      signed char* dst, int nthreads, int* all_pm1);

/* ---- direct solves through the polynomial form ------------------------------
 * The reference's Iterative.solve takes a preconditioner from the operator
 * (runlmc/approx/iterative.py:47-51: M = getattr(K, 'preconditioner', None), handed to
 * SciPy's minres / cg) and no reference operator provides one.  This library does, for
 * operators whose top rows are ALL in the polynomial-subspace form (rl_gridop_form,
 * rank r): such a K~ is  F M F^T + diag(eps)  with F = W Phi block-diagonal by output
 * (n x D r) -- a diagonal plus rank D r -- and the Woodbury identity gives
 *     K~^-1 b = E^-1 b - E^-1 F Z F^T E^-1 b          (Z: D r x D r, host Cholesky)
 *     log det K~ = sum_d n_d log eps_d + log det(I + L^T M L),   F^T E^-1 F = L L^T
 * (csrc/rl_direct.h).  With M = K~^-1 to roundoff a preconditioned iteration IS
 * iterative refinement, and the reference's own residual rule ends it.
 *
 * rl_ski_factor: (re)builds the factorisation for the CURRENT parameters and noise
 * (cached on the handle until either changes; runs the pending set-time verification of
 * the polynomial form whatever the batch gate).  *available = 0 when the operator has
 * no such form (a top row on the transform or filter kernels, several grids, a 2-D
 * grid, D > 16, noise not constant per output, an output with fewer rows than the
 * basis, D r > 576 on a system of fewer than 10^5 (D r / 576)^3 rows: there the host's
 * ~1.5 (D r)^3 multiply-adds cost more than the Krylov solve) -- rl_last_error() then says
 * which; that is not an error.  *logdet receives
 * log det K~ of the handle's operator -- the quantity the reference computes by a dense
 * Cholesky (models/interpolated_llgp.py:262-276) -- and *cond an estimate of the condition
 * number of the D r x D r system (squared ratio of its Cholesky pivots).  Any pointer may
 * be NULL.                                                                              */
int rl_ski_factor(rl_ski* s, int* available, double* logdet, double* cond);
/* *available = 2: NOT every top row is in the polynomial form (a Matern row, say, on the filter
 * kernels), but the polynomial subspace still holds at least 0.8 of every row's spectrum: the
 * factorisation then inverts the operator's PROJECTION on the subspace,  F M_r F^T + E  with
 * C_q = Phi^T T_q Phi of every row -- not K~^-1 (no log det), but the symmetric positive definite
 * M of the reference's  sla.cg(op, y, M=M)  (iterative.py:47-51).  rl_solve_pcg runs SciPy's
 * statements of preconditioned conjugate gradients with it, all systems in lockstep, each ended
 * by the reference's rule ||b - K~ x||_2 < tol (the recurrence's residual norm every iteration,
 * the explicit residual before a system is let go), at most maxiter iterations (<= 0: n).
 * BASELINE's 'mix' family (four smooth rows and a Matern row) ends in 4-5 iterations where
 * MINRES runs 590 without meeting the rule; Matern rows alone take ~100.
 * *available = 3: the same on a LARGER basis of the handle's own -- operators of >= 10^5 rows
 * not entirely in the polynomial form: up to 192 polynomials per output (as many blocks of 48
 * as D * R <= 2048, m >= 8 R and n >= 10^5 (D R / 960)^3 allow; a factorisation uses the
 * first blocks that hold all but 1e-5 of every row's trace), table, Gram matrices once per
 * handle, the map per parameter update
 * (csrc/rl_solve.hip hz_*).  C5 (n = 10^6): Matern rows 38 iterations (48 functions: 1358),
 * the mix family 11 (59).  Callers treat 2 and 3 alike.  Also valid with
 * *available = 1 (then M is K~^-1 and one iteration suffices: rl_solve_direct is the shorter way).
 *   iters_out / resid_out / istop_out as rl_solve_direct (istop 6: maxiter reached).       */
int rl_solve_pcg(rl_ski* s, const double* B, double* X, int nrhs, double tol, int maxiter,
                 int* iters_out, double* resid_out, int* istop_out, void* stream);
/* log det K~ on that path (the reference's log_det_K, models/interpolated_llgp.py:262-276, is a dense
 * Cholesky; rounds 1-5 answered with Lanczos quadrature of the UNpreconditioned solves, which these
 * operators' solves no longer run):  log det K~ = log det P + tr log(P^-1/2 K~ P^-1/2),  P the
 * factorised matrix (its log det exact, determinant lemma), the trace by Hutchinson + Gauss
 * quadrature on the Lanczos matrices conjugate gradients build anyway -- unbiased when the right-hand
 * sides have covariance P.  The preconditioned operator is close to I, so the estimator's variance
 * is orders of magnitude below the plain quadrature's: a dozen probes do.
 * rl_ski_precond_sample: Rout[v] = P^1/2 Win[v]  (P = E^1/2 B B^T E^1/2, B = I + Q (C - I) Q^T from
 *   the factorisation's own Cholesky factors: one projection, one dense map, one expansion; Win, Rout
 *   dev [nvec][n], caller's row order; Win rows of identity covariance, e.g. the reference's +-1
 *   probes), *logdet_p = log det P.  The first call makes the handle keep that map with every later
 *   factorisation.
 * rl_solve_pcg_lanczos: rl_solve_pcg that also leaves each system's Lanczos matrix of P^-1/2 K~ P^-1/2
 *   (lanczos_out host [nrhs][cap][2]: diagonal, off-diagonal; as rl_solve_batch_lanczos) and
 *   sqnorms_out[v] = r0^T P^-1 r0 -- what rl_slq_log_quadrature takes; at most cap iterations, no
 *   restarts from explicit residuals.                                                       */
int rl_ski_precond_sample(rl_ski* s, const double* Win, double* Rout, int nvec, double* logdet_p,
                          void* stream);
int rl_solve_pcg_lanczos(rl_ski* s, const double* B, double* X, int nrhs, double tol, int maxiter,
                         int* iters_out, double* resid_out, int* istop_out, double* lanczos_out,
                         int cap, double* sqnorms_out, void* stream);
/* The pieces of that form, for callers that work in its coefficient space (the gradient's
 * Gram terms: with T = Phi C Phi^T,  u~_a . T v~_b = c_u[a]^T C c_v[b],  c_u = Phi^T W^T u
 * -- runlmc_amd/lmc/likelihood.py; reference loops: lmc/likelihood.py:48-96):
 * rl_ski_project: out[v][d][j] = (Phi^T W^T X[v])[d][j], the r coefficients per output on the
 * ORTHONORMAL polynomials (dev out [nvec][D][r], *rank = r; X dev [nvec][n], caller's row
 * order); RL_ELIMIT when rl_ski_factor reports *available = 0.
 * rl_gridop_poly_coeffs: C_q = Phi^T T_q Phi (symmetrised) of top row q into host out
 * [r][r] (cap >= r * r values) and *rank = r -- or *rank = 0 when the row is not in the
 * polynomial form (the pending verification runs first; out may be NULL to ask only).   */
int rl_ski_project(rl_ski* s, const double* X, int nvec, double* out, int* rank, void* stream);
/* The same for GRID vectors and any of the basis sizes: out[v][d][j] = (Phi_rank^T X[v][d])[j]
 * (X dev [nvec][D*m], out dev [nvec][D][rank], rank one of 24, 32, 36, 40, 48) -- for callers
 * whose derivative rows need a larger basis than the operator's own (Phi is nested: the first
 * r functions of a larger basis are the smaller one).                                       */
int rl_gridop_project(rl_gridop* g, const double* X, int nvec, int rank, double* out, void* stream);
/* Where the set-time verification of the polynomial form starts its ladder of basis sizes
 * (24, 32, 36, 40, 48; 0 = from the bottom): a handle holding the derivative rows of kernels
 * whose own rows were accepted at rank r elsewhere need not try the smaller ones.  Only speed
 * depends on it (a rank that is larger than needed costs proportionally more).           */
int rl_gridop_set_rank_hint(rl_gridop* g, int rank);
int rl_gridop_poly_coeffs(rl_gridop* g, int q, double* out, int cap, int* rank);
/* X[v] = K~^-1 B[v]:  x = M b, then  x += M (b - K~ x)  while the reference's rule
 * ||b - K~ x||_2 < tol (iterative.py:36-42,54-58) does not hold, at most max_refine
 * times (the residual is taken through the handle's ordinary product, rl_ski_mvm's
 * path).  B, X dev [nrhs][n], may not alias.
 *   iters_out host [nrhs]  applications of M (1 + refinements) -- the count a
 *                          preconditioned Krylov method would report
 *   resid_out host [nrhs]  final ||b - K~ x||_2
 *   istop_out host [nrhs]  10 = the reference's residual rule; 12 = tolerance not
 *                          reached within max_refine refinements (the iterate is
 *                          returned, as the reference returns its own: iterative.py:55-58)
 * RL_ELIMIT when rl_ski_factor would report *available = 0.  Synchronises.            */
int rl_solve_direct(rl_ski* s, const double* B, double* X, int nrhs, double tol,
                    int max_refine, int* iters_out, double* resid_out, int* istop_out,
                    void* stream);

/* ---- partial sums of the Hutchinson gradient --------------------------------
 * Replace the P*(N+1) operator products of StochasticDeriv.d_normal_quadratic
 * / d_logdet_K (runlmc/lmc/stochastic_deriv.py:69-78) driven by
 * LMCLikelihood's loops (runlmc/lmc/likelihood.py:48-96): with u~ = W^T u
 * reshaped D x m,  u^T W (dB (x) T) W^T v = sum_ab dB[a,b] u~_a . (T v~_b), so
 * one D x D Gram matrix per (vector pair, top row) serves every
 * coregionalisation parameter at once.
 *   out[v][a][b] = sum_i U[v][a*m+i] * V[v][b*m+i];  U, V dev [nvec][D*m],
 *   out dev [nvec][D][D].                                                     */
int rl_cross_dots(const double* U, const double* V, int nvec, int D, int m, double* out,
                  void* stream);
/* out[v][d] = sum_{i in output d} U[v][i] * V[v][i] over data-space vectors
 * (noise gradient, likelihood.py:89-96 with Diag(repeat(e_d, lens))):
 * offsets dev int[D+1], U, V dev [nvec][n], out dev [nvec][D].                */
int rl_segment_dots(const double* U, const double* V, const int* offsets, int nvec, int n,
                    int D, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RUNLMC_HIP_H */
