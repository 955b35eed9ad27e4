// Batched Krylov solves (rl_solver.h), direct solves through the polynomial form (rl_direct.h),
// host helpers (Lanczos quadrature, probe narrowing), gradient partial sums (one of the three
// translation units of librunlmc_hip.so, rl_host.h).
#include "rl_host.h"
#include "rl_solver.h"
#include "rl_direct.h"

void free_work(SolverWork& w) {
    for (double*& p : w.vec) { if (p) (void)hipFree(p); p = nullptr; }
    for (double*& p : w.S) { if (p) (void)hipFree(p); p = nullptr; }
    for (double*& p : w.part) { if (p) (void)hipFree(p); p = nullptr; }
    if (w.I) (void)hipFree(w.I);
    if (w.count) (void)hipFree(w.count);
    if (w.resid) (void)hipFree(w.resid);
    w.I = nullptr; w.count = nullptr; w.resid = nullptr; w.lanczos = nullptr;
}

static size_t ws_cache_limit(const rl_ski* s) {
    const size_t mb = s->kn.ws_cache_mb >= 0 ? (size_t)s->kn.ws_cache_mb : 16384;
    return mb << 20;
}

// hands the buffers back to the handle (or frees them) and destroys a captured
// graph at scope exit; kept apart so the plain struct can be copied into launch
// closures
struct SolverWorkGuard {
    rl_ski* s;
    SolverWork* w;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    SolverWorkGuard(rl_ski* s_, SolverWork* w_) : s(s_), w(w_) {}
    ~SolverWorkGuard() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        w->lanczos = nullptr;       // owned by the handle (rl_ski::lanczos_buf)
        size_t nv = 0;
        for (double* p : w->vec) nv += p != nullptr;
        const size_t bytes = nv * s->ws_vec_cap * sizeof(double);
        if (w->I != nullptr && !s->ws_valid && bytes <= ws_cache_limit(s)) {
            s->ws = *w;
            s->ws_valid = true;
        } else {
            free_work(*w);
        }
    }
};

// `need`: bit i set = vec[i] is used by this call
static int solver_alloc(rl_ski* s, SolverWork& w, unsigned need, int nrhs, int n, int nblk,
                        hipStream_t st) {
    const size_t ve = (size_t)nrhs * n, pe = (size_t)nrhs * nblk;
    if (s->ws_valid) {
        if (s->ws_vec_cap >= ve && s->ws_rhs_cap >= (size_t)nrhs && s->ws_part_cap >= pe)
            w = s->ws;              // checked out; the guard hands it back
        else
            free_work(s->ws);
        s->ws = SolverWork();
        s->ws_valid = false;
    }
    if (w.I == nullptr) {
        s->ws_vec_cap = ve;
        s->ws_rhs_cap = (size_t)nrhs;
        s->ws_part_cap = pe;
        for (int i = 0; i < 2; ++i)
            RL_HIP(hipMalloc((void**)&w.S[i], (size_t)nrhs * S_NFIELDS * sizeof(double)));
        RL_HIP(hipMalloc((void**)&w.I, (size_t)nrhs * I_NFIELDS * sizeof(int)));
        for (int i = 0; i < 4; ++i)
            RL_HIP(hipMalloc((void**)&w.part[i], pe * sizeof(double)));
        RL_HIP(hipMalloc((void**)&w.count, 4 * sizeof(int)));
        RL_HIP(hipMalloc((void**)&w.resid, (size_t)nrhs * sizeof(double)));
    }
    for (int i = 0; i < 10; ++i)
        if (((need >> i) & 1u) && w.vec[i] == nullptr)
            RL_HIP(hipMalloc((void**)&w.vec[i], s->ws_vec_cap * sizeof(double)));
    RL_HIP(hipMemsetAsync(w.count, 0, 4 * sizeof(int), st));
    RL_HIP(hipMemsetAsync(w.resid, 0, (size_t)nrhs * sizeof(double), st));
    return RL_OK;
}

static int active_count(SolverWork& w, int nrhs, hipStream_t st, int* out) {
    RL_LAUNCH(k_count_active, dim3(1), dim3(64), 0, st, w.I, nrhs, w.count);
    RL_HIP(hipMemcpyAsync(out, w.count, sizeof(int), hipMemcpyDeviceToHost, st));
    RL_HIP(hipStreamSynchronize(st));
    return RL_OK;
}

// explicit residual ||b - K x|| of every system into w.resid; freeze those
// below tol when `freeze`
static int residual_check(rl_ski* s, SolverWork& w, const double* B, const double* X,
                          double* scratch, int nrhs, int n, int nblk, double tol, int freeze,
                          hipStream_t st) {
    RL_TRY(ski_mvm_int(s, X, scratch, nrhs, st));
    dim3 grid(nblk, nrhs), blk(RL_SOLVER_THREADS);
    const size_t red = RL_SOLVER_THREADS * sizeof(double);
    RL_LAUNCH(k_resid_partial, grid, blk, red, st, B, (const double*)scratch, n, w.part[2]);
    RL_LAUNCH(k_resid_finish, dim3((nrhs + 63) / 64), dim3(64), 0, st,
              (const double*)w.part[2], nblk, nrhs, w.resid, w.I, tol, freeze);
    return RL_OK;
}

// one MINRES iteration: identical arguments every time (buffer roles rotate on
// the device), so a captured run of these can be replayed
static int minres_iteration(rl_ski* s, const MinresBufs& mb, SolverWork& w, int nrhs, int n,
                            int nblk, double rtol, int maxiter, hipStream_t st) {
    dim3 grid(nblk, nrhs), blk(RL_SOLVER_THREADS);
    const size_t red = RL_SOLVER_THREADS * sizeof(double);
    RL_TRY(ski_mvm_int(s, mb.v, mb.q, nrhs, st));
    RL_LAUNCH(k_minres_a, grid, blk, red, st, mb, n, w.part[0]);
    RL_LAUNCH(k_minres_b, grid, blk, red, st, mb, n, (const double*)w.part[0], w.part[1]);
    RL_LAUNCH(k_minres_c, grid, blk, red, st, mb, n, (const double*)w.part[0],
              (const double*)w.part[1], w.part[2]);
    RL_LAUNCH(k_minres_test, dim3(1), blk, 0, st, mb, (const double*)w.part[2], nblk, nrhs,
              rtol, maxiter);
    return RL_OK;
}

// one round of the two-kernel MINRES (rl_solver.h): operator product on the
// unnormalised Lanczos vector, P, B; identical arguments every round
static bool g_is_v2(const rl_gridop* g) { return g->v2; }

// Polynomial rounds of a small MINRES solve (rl_solver.h): are they possible for
// this handle and batch, and if so build (once) the row blocks -- at most 1024 rows,
// each inside ONE output -- and make sure the form is verified at rank RL_LR_RS.
// Default where the grid is eligible for the polynomial form anyway (1-D, >= 2048
// points; RUNLMC_NO_POLY_ROUND=1 switches the rounds off, RUNLMC_POLY_ROUND=1 also
// admits grids of 96 .. 2047 points).  Measured at C2: a round is 17 + 12 us in two
// kernels against 42 us in five, the NLL + gradient step 6.1 against 7.5 ms
// (eps = 1: 5.3 against 6.8) including the 0.54 ms verification per step.
static int poly_round_prepare(rl_ski* s, int nrhs, int max_blk, bool* ok) {
    *ok = false;
    rl_gridop* g = s->g;
    if (!s->extra.empty() || s->W4_base == nullptr || s->h_base.empty() ||
        !g->lr_round_try || s->poly_nblk < 0 || s->kn.no_poly_round || g->kn.no_poly_round)
        return RL_OK;
    if (s->poly_nblk == 0) {
        const int D = g->D, m = g->m, n = s->n;
        std::vector<int> tab, ob(D + 1, 0);
        int i = 0;
        for (int d = 0; d < D; ++d) {
            ob[d] = (int)tab.size() / RL_PT;
            const int start = i;
            while (i < n && s->h_base[i] < (d + 1) * m) ++i;
            const int cnt = i - start;
            // (1024 rows per block; 512 / 256 measured slower at C2: 8.1 / 11.1 against
            // 7.0 ms per step -- every workgroup repeats the mix of its system)
            const int nb = (cnt + 1023) / 1024;
            for (int b = 0; b < nb; ++b) {
                const int per = (cnt + nb - 1) / nb;
                const int r0 = start + b * per, r1 = std::min(start + (b + 1) * per, i);
                tab.push_back(r0);
                tab.push_back(r1);
                tab.push_back(d);
                // grid points (within the output) the block's rows touch
                const int gfirst = s->h_base[r0] - d * m;
                tab.push_back(gfirst);
                tab.push_back(s->h_base[r1 - 1] - d * m + 4 - gfirst);
            }
        }
        ob[D] = (int)tab.size() / RL_PT;
        if (i != n || tab.empty()) {
            s->poly_nblk = -1;
            return RL_OK;
        }
        RL_TRY(upload_raw((void**)&s->poly_tab, tab.data(), tab.size() * sizeof(int)));
        RL_TRY(upload_raw((void**)&s->poly_ob, ob.data(), ob.size() * sizeof(int)));
        s->poly_nblk = (int)tab.size() / RL_PT;
    }
    if (s->poly_nblk > std::max(max_blk, RL_SOLVER_THREADS)) return RL_OK;
    RL_TRY(lr_ensure(g));
    if (!g->lr_ok || g->lr_r != RL_LR_RS) return RL_OK;
    const size_t need = (size_t)nrhs * s->poly_nblk * RL_LR_RS;
    if (s->poly_part_cap < need) {
        if (s->poly_part) RL_HIP(hipFree(s->poly_part));
        s->poly_part = nullptr;
        s->poly_part_cap = 0;
        RL_HIP(hipMalloc((void**)&s->poly_part, need * sizeof(double)));
        s->poly_part_cap = need;
    }
    *ok = true;
    return RL_OK;
}

static int minres2_round(rl_ski* s, const Minres2Bufs& mb, int nrhs, int n, int nblk, int round,
                         double rtol, int maxiter, hipStream_t st) {
    const int par = (round - 1) & 1;
    dim3 grid(nblk, nrhs), blk(RL_SOLVER_THREADS);
    const size_t red = RL_SOLVER_THREADS * sizeof(double);
    const double* yin = mb.tri[1 - par];      // y_{r-1}: the operator's input this round
    if (mb.poly_part != nullptr) {
        // the operator lives inside P and B (rl_solver.h: polynomial rounds)
        trace_once("minres round: polynomial form inside P and B, no grid vectors");
        const size_t lds = (2 * RL_SOLVER_THREADS +
                            std::max(mb.poly_D * RL_LR_RS + 11 * RL_LR_RS + RL_PG, RL_SOLVER_THREADS)) *
                           sizeof(double);
        RL_LAUNCH(k_minres2_p, grid, blk, lds, st, mb, n, par);
        RL_LAUNCH(k_minres2_b, grid, blk, lds, st, mb, n, par, rtol, maxiter);
        return RL_OK;
    }
    if (mb.W_indptr != nullptr && mb.fuse_wt) {
        trace_once(g_is_v2(s->g) ? "minres round: W^T in k2_cols_fwd, W in P"
                                 : "minres round: W^T in k_cols_fwd, W in P");
        // W^T fused into the column transforms (gathered while loading), W into
        // P: three grid kernels, P, B
        rl_gridop* g = s->g;
        Gather gs;
        gs.indptr = s->WT_indptr;
        gs.indices = s->WT_indices;
        gs.vals = s->WT_data;
        gs.src = yin;
        gs.n = n;
        gs.nnz = s->nnzWT;
        gs.lo = s->WT_lo;
        MixParams mp{g->Q, g->nfac, g->spec, g->facA, g->facW, g->facQ, g->kappa,
                     g->mixtab_ok ? g->mixtab : nullptr,
                     g->mixtab_ok ? g->mixtab + (size_t)g->D * g->L : nullptr};
        if (g->v2)
            RL_TRY(mvm_chunk_v2(g, mp, nullptr, s->G2, nrhs, ((size_t)nrhs + 1) / 2, st, &gs,
                                mb.giter));
        else
            RL_TRY(mvm_chunk_v1(g, mp, nullptr, s->G2, nrhs, ((size_t)nrhs + 1) / 2, st, &gs,
                                mb.giter));
    } else if (mb.W_indptr != nullptr) {
        // W product fused into P: only W^T and the grid product run here
        RL_TRY(ski_wt_int(s, yin, s->G1, nrhs, st, mb.giter));
        RL_TRY(rl_gridop_mvm(s->g, s->G1, s->G2, nrhs, st));
    } else {
        // row-polynomial operator: the previous round's B finishes inside this projection
        // (y_{r-1} = y' - coef y_{r-2} formed and stored while the tile is staged; rl_rowpoly.h)
        if (mb.fuse_b && round >= 2) s->rp_fuse = RpFuse{mb.tri[par], mb.coef, mb.nrmB};
        if (mb.fuse_p) {
            // ... and this round's P inside the expansion (rl_rowpoly.h RpPFuse): its scalar
            // head between the coefficient map and the expansion, no k_minres2_p
            s->rp_pfuse = RpPFuse{mb.pc, mb.tri[par], mb.w[par], mb.w[1 - par], mb.x,
                                  mb.partA[1 - par], mb.partC};
            const Minres2Bufs mbc = mb;
            s->rp_mid = [mbc, par](hipStream_t q) {
                RL_LAUNCH(k_minres2_ph, dim3(mbc.fuse_p), dim3(RL_SOLVER_THREADS), red, q, mbc, par);
            };
        }
        const int rc = ski_mvm_int(s, yin, mb.q, nrhs, st, mb.giter, mb.fuse_p || mb.eps_runs == 0);
        s->rp_fuse = RpFuse{nullptr, nullptr, nullptr};
        s->rp_pfuse = RpPFuse{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        s->rp_mid = nullptr;
        if (rc != RL_OK) return rc;
        if (mb.fuse_p) {
            RL_LAUNCH(k_minres2_bh, dim3(nrhs), blk, red, st, mb, mb.np, par, rtol, maxiter);
            // (P inside the W product: no projection carries B's vector work)
            if (!mb.fuse_b) RL_LAUNCH(k_minres2_bv, grid, blk, red, st, mb, n, par);
            return RL_OK;
        }
    }
    RL_LAUNCH(k_minres2_p, grid, blk, red, st, mb, n, par);
    if (mb.fuse_b)
        RL_LAUNCH(k_minres2_bh, dim3(nrhs), blk, red, st, mb, nblk, par, rtol, maxiter);
    else
        RL_LAUNCH(k_minres2_b, grid, blk, red, st, mb, n, par, rtol, maxiter);
    return RL_OK;
}

static int cg_iteration(rl_ski* s, SolverWork& w, double* X, int nrhs, int n, int nblk,
                        int first, int maxiter, hipStream_t st) {
    dim3 grid(nblk, nrhs), blk(RL_SOLVER_THREADS);
    const size_t red = RL_SOLVER_THREADS * sizeof(double);
    dim3 grid1((nrhs + 63) / 64), blk1(64);
    double *r = w.vec[0], *p = w.vec[1], *q = w.vec[2];
    RL_LAUNCH(k_cg_head, grid1, blk1, 0, st, w.S[0], w.I, (const double*)w.part[1], nblk, nrhs,
              first, maxiter);
    RL_LAUNCH(k_cg_p, grid, blk, 0, st, p, (const double*)r, n, (const double*)w.S[0],
              (const int*)w.I);
    RL_TRY(ski_mvm_int(s, p, q, nrhs, st));
    RL_LAUNCH(k_dot_partial, grid, blk, red, st, (const double*)p, (const double*)q, n,
              w.part[0]);
    RL_LAUNCH(k_cg_update, grid, blk, red, st, X, r, (const double*)p, (const double*)q, n,
              (const double*)w.S[0], w.I, (const double*)w.part[0], w.part[1]);
    RL_LAUNCH(k_count_iter, grid1, blk1, 0, st, w.I, nrhs);
    return RL_OK;
}

static int solve_batch_impl(rl_ski* s, const double* B, double* X, int nrhs, int method,
                            double tol, int check_every, int maxiter, int* iters_out,
                            double* resid_out, int* istop_out, double* lanczos_out,
                            int lanczos_cap, void* stream);


extern "C" int rl_solve_batch(rl_ski* s, const double* B, double* X, int nrhs, int method,
                              double tol, int check_every, int maxiter, int* iters_out,
                              double* resid_out, int* istop_out, void* stream) {
    return solve_batch_impl(s, B, X, nrhs, method, tol, check_every, maxiter, iters_out,
                            resid_out, istop_out, nullptr, 0, stream);
}

extern "C" int rl_solve_batch_lanczos(rl_ski* s, const double* B, double* X, int nrhs,
                                      int method, double tol, int check_every, int maxiter,
                                      int* iters_out, double* resid_out, int* istop_out,
                                      double* lanczos_out, int lanczos_cap, void* stream) {
    if (lanczos_out != nullptr && lanczos_cap < 1)
        return fail(RL_EINVAL, "rl_solve_batch_lanczos: lanczos_cap < 1");
    if (method != RL_MINRES && method != RL_MINRES_RULE)
        return fail(RL_EINVAL, "rl_solve_batch_lanczos: RL_MINRES or RL_MINRES_RULE only");
    return solve_batch_impl(s, B, X, nrhs, method, tol, check_every, maxiter, iters_out,
                            resid_out, istop_out, lanczos_out, lanczos_cap, stream);
}

static int solve_batch_impl(rl_ski* s, const double* B, double* X, int nrhs, int method,
                            double tol, int check_every, int maxiter, int* iters_out,
                            double* resid_out, int* istop_out, double* lanczos_out,
                            int lanczos_cap, void* stream) {
    if (!s || !B || !X) return fail(RL_EINVAL, "rl_solve_batch: NULL argument");
    if (nrhs < 0) return fail(RL_EINVAL, "rl_solve_batch: nrhs < 0");
    if (method != RL_MINRES && method != RL_CG && method != RL_MINRES_RULE)
        return fail(RL_EINVAL, "rl_solve_batch: unknown method");
    // RL_MINRES_RULE: MINRES whose own stopping tests are off -- a system ends on the
    // reference's explicit residual rule or at maxiter (the kernels read rtol < 0)
    const bool rule_only = method == RL_MINRES_RULE;
    if (rule_only) {
        if (check_every <= 0)
            return fail(RL_EINVAL, "rl_solve_batch: RL_MINRES_RULE needs check_every > 0");
        method = RL_MINRES;
    }
    if (!(tol > 0.0)) return fail(RL_EINVAL, "rl_solve_batch: tol must be > 0");
    if (check_every < 0) return fail(RL_EINVAL, "rl_solve_batch: check_every < 0");
    if (nrhs == 0) return RL_OK;
    RL_HIP(hipSetDevice(s->g->device));
    // the iteration is captured into a hipGraph, which needs a capturable
    // stream: run on the handle's own stream, ordered after the caller's
    RL_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (!s->solver_stream)
        RL_HIP(hipStreamCreateWithFlags(&s->solver_stream, hipStreamNonBlocking));
    hipStream_t st = s->solver_stream;
    const int n = s->n;
    if (maxiter <= 0) maxiter = n;
    const double rtol = rule_only ? -1.0 : (tol < 1e-10 ? tol : 1e-10);
    const int rows_per_blk = 1024;       // (512 / 320 / 256 measured slower at C2)
    int nblk = (n + rows_per_blk - 1) / rows_per_blk;
    // <= RL_SOLVER_THREADS: the partial sums are read one per thread.  (64 until round 4; at C5
    // 120 blocks per system measured 0.460-0.467 against 0.489 ms per 17-system round and
    // 2.99-3.02 against 3.04-3.06 per 129-system round: 17 x 64 workgroups are 4.25 per CU)
    int max_blk = 120;
    if (s->kn.solver_maxblk > 0) max_blk = std::max(1, std::min(s->kn.solver_maxblk, RL_SOLVER_THREADS));
    nblk = std::max(1, std::min(nblk, max_blk));
    // small single-term MINRES solves of a smooth kernel: polynomial rounds, whose
    // row blocks follow the outputs (every other kernel of the solve takes the same
    // NUMBER of blocks; its partial sums do not care where the block borders are)
    bool poly_round = false;
    if (method == RL_MINRES && !s->kn.minres_v1 &&
        (size_t)n * nrhs < ((size_t)1 << 22) && !s->kn.no_fuse_w)
        RL_TRY(poly_round_prepare(s, nrhs, max_blk, &poly_round));
    if (poly_round) nblk = s->poly_nblk;
    dim3 grid(nblk, nrhs), blk(RL_SOLVER_THREADS);
    const size_t red = RL_SOLVER_THREADS * sizeof(double);

    // iterations per graph replay: a divisor of check_every, at most 10
    int per_graph = 10;
    if (check_every > 0) {
        per_graph = 1;
        for (int d = 1; d <= 10; ++d)
            if (check_every % d == 0) per_graph = d;
    }
    const bool use_graph = !s->kn.no_graph;

    SolverWork w;
    SolverWorkGuard guard(s, &w);
    // (vec[5]: v of the four-kernel MINRES only; vec[8], vec[9]: right-hand sides
    // and solutions in the handle's internal row order)
    unsigned need = method == RL_MINRES ? 0x5bu : 0x0fu;      // two-kernel MINRES: 0, 1, 3, 4, 6
    if (method == RL_MINRES && s->kn.minres_v1) need = 0x7fu;
    if (s->permuted) need |= 0x300u;
    RL_TRY(solver_alloc(s, w, need, nrhs, n, nblk, st));
    // everything the operator product allocates lazily must exist before capture
    RL_TRY(ski_reserve(s, nrhs));
    RL_TRY(gridop_prepare(s->g, nrhs));
    for (const SkiTerm& t : s->extra) RL_TRY(gridop_prepare(t.g, nrhs));
    if (rp_ok(s, nrhs)) RL_TRY(rp_prepare(s, nrhs));       // (row-polynomial rounds: F, runs, partial sums)
    int active = nrhs;
    int done = 0;          // iterations issued so far

    // iterate in the handle's internal row order (data sorted by grid position)
    const double* Bi = B;
    double* Xi = X;
    if (s->permuted) {
        permute_rows(s, B, w.vec[8], nrhs, 0, st);
        Bi = w.vec[8];
        Xi = w.vec[9];
    }
    RL_LAUNCH(k_dot_partial, grid, blk, red, st, Bi, Bi, n, w.part[0]);
    if (method == RL_MINRES && !s->kn.minres_v1) {
        // two vector kernels per round (rl_solver.h: Minres2Bufs)
        Minres2Bufs mb;
        mb.tri[0] = w.vec[0]; mb.tri[1] = w.vec[1];
        mb.w[0] = w.vec[3]; mb.w[1] = w.vec[4];
        mb.q = w.vec[6];
        mb.x = Xi;
        mb.S[0] = w.S[0]; mb.S[1] = w.S[1];
        mb.I = w.I;
        mb.giter = w.count + 1;
        mb.partA[0] = w.part[0]; mb.partA[1] = w.part[3];
        mb.partB = w.part[1];
        mb.partC = w.part[2];
        mb.lanczos = nullptr;
        mb.lanczos_cap = 0;
        // the W product rides inside P when the problem is small enough to be
        // launch-bound (a big one amortises the CSR over 8 vectors in k_spmv<8>)
        const bool fuse_w = s->extra.empty() && (size_t)n * nrhs < ((size_t)1 << 22) &&
                            !s->kn.no_fuse_w;
        mb.W_indptr = fuse_w ? s->W_indptr : nullptr;
        // ... and W^T inside the first grid kernel when that is a k2_cols_fwd and
        // the batch is one chunk (the operator input is then the rotating buffer
        // itself: no copy of the new Lanczos vector)
        // (only while the rows of W^T are short -- about as many data points as grid
        // points: a row of 30 entries, as on the weather workload, is a serial chain
        // of gathers inside the transform kernel: 1.45 vs 1.23 s per fit)
        const bool short_rows = (size_t)s->nnzWT <= (size_t)8 * s->ngrid;
        const bool fuse_wt = fuse_w && short_rows && s->g->Q >= 1 && !s->g->wide &&
                             ((size_t)nrhs + 1) / 2 <= s->g->chunk_pairs &&
                             !s->kn.no_fuse_wt;
        mb.fuse_wt = fuse_wt ? 1 : 0;
        mb.W_indices = s->W_indices;
        mb.W_data = s->W_data;
        mb.W_nnz = s->nnz;
        mb.W4_base = s->W4_base;
        mb.W4_w = s->W4_w;
        mb.g = s->G2;
        mb.eps = s->has_noise ? s->noise_diag : nullptr;
        mb.ngrid = s->ngrid;
        mb.poly_part = nullptr;
        mb.giter2 = w.count + 3;
        if (poly_round && fuse_w) {
            rl_gridop* g = s->g;
            mb.poly_tab = s->poly_tab;
            mb.poly_ob = s->poly_ob;
            mb.poly_part = s->poly_part;
            mb.poly_M = g->lr_M;
            mb.poly_beta = g->lr_beta;
            mb.poly_D = g->D;
            mb.poly_m = g->m;
        }
        // single-term operator, W as its own kernel, noise in a few constant runs
        // (one per output): the noise term moves into P
        mb.eps_runs = 0;
        if (!fuse_w && s->extra.empty() && s->has_noise && !s->eps_end.empty()) {
            mb.eps_runs = (int)s->eps_end.size();
            for (int k = 0; k < mb.eps_runs; ++k) {
                mb.eps_end[k] = s->eps_end[k];
                mb.eps_val[k] = s->eps_val[k];
            }
        }
        if (lanczos_out != nullptr) {
            const size_t bytes = (size_t)nrhs * lanczos_cap * 2 * sizeof(double);
            if (s->lanczos_bytes < bytes) {
                if (s->lanczos_buf) RL_HIP(hipFree(s->lanczos_buf));
                s->lanczos_buf = nullptr;
                s->lanczos_bytes = 0;
                RL_HIP(hipMalloc((void**)&s->lanczos_buf, bytes));
                s->lanczos_bytes = bytes;
            }
            w.lanczos = s->lanczos_buf;
            RL_HIP(hipMemsetAsync(w.lanczos, 0, bytes, st));
            mb.lanczos = w.lanczos;
            mb.lanczos_cap = lanczos_cap;
        }
        // row-polynomial operator (every round of this solve takes it: the same test as the
        // product's): B's vector work rides in the next round's projection
        mb.fuse_b = 0;
        mb.coef = nullptr;
        mb.nrmB = nullptr;
        mb.nrm_n = 0;
        mb.fuse_p = 0;
        mb.pc = nullptr;
        mb.np = 0;
        if (mb.W_indptr == nullptr && mb.poly_part == nullptr && rp_ok(s, nrhs) && rp_ready(s, nrhs) &&
            !(s->kn.rp_fly & 2) && !s->kn.no_rp_fuse) {
            const size_t need = (size_t)nrhs * (s->rp_nruns + 1);
            if (s->rp_nrm_cap < need) {
                if (s->rp_nrm) RL_HIP(hipFree(s->rp_nrm));
                s->rp_nrm = nullptr;
                s->rp_nrm_cap = 0;
                RL_HIP(hipMalloc((void**)&s->rp_nrm, need * sizeof(double)));
                s->rp_nrm_cap = need;
            }
            mb.fuse_b = 1;
            mb.coef = s->rp_nrm;
            mb.nrmB = s->rp_nrm + nrhs;
            mb.nrm_n = s->rp_nruns;
            trace_once("minres round: B inside the row-polynomial projection (k_minres2_bh)");
            // (the fused expansion keeps 8 doubles of LDS per system: past 1024 systems -- 64 KB --
            // the launch would fail, so such batches keep P as its own kernel)
            size_t pfuse_lds = (size_t)nrhs * 8 * sizeof(double);
#if defined(RL_EMU)
            pfuse_lds += 256 * sizeof(double);
#endif
            if (s->kn.rp_pfuse && pfuse_lds <= (size_t)64 * 1024) {
                // ... and P inside the expansion (k_minres2_ph + k_rp_expand<.., true>)
                const int np = (n + 255) / 256;
                const size_t needp = (size_t)nrhs * (RL_RP_PCW + 3 * (size_t)np);
                if (s->rp_pp_cap < needp) {
                    if (s->rp_pp) RL_HIP(hipFree(s->rp_pp));
                    s->rp_pp = nullptr;
                    s->rp_pp_cap = 0;
                    RL_HIP(hipMalloc((void**)&s->rp_pp, needp * sizeof(double)));
                    s->rp_pp_cap = needp;
                }
                RL_HIP(hipMemsetAsync(s->rp_pp, 0, needp * sizeof(double), st));
                mb.fuse_p = nrhs;                 // (the head kernel's grid)
                mb.pc = s->rp_pp;
                mb.np = np;
                mb.partA[0] = s->rp_pp + (size_t)nrhs * RL_RP_PCW;
                mb.partA[1] = mb.partA[0] + (size_t)nrhs * np;
                mb.partC = mb.partA[1] + (size_t)nrhs * np;
                trace_once("minres round: P inside the row-polynomial expansion (k_minres2_ph)");
            }
        } else if (mb.W_indptr == nullptr && mb.poly_part == nullptr && s->kn.w_pfuse &&
                   s->extra.empty() && w_staged_p_ok(s, nrhs) &&
                   // (an operator wholly in the polynomial form has better rounds: row-polynomial
                   // ones, or the W kernel that expands the coefficients itself)
                   !(s->g->lr_ok && !s->g->lr_dirty) && !(rp_ok(s, nrhs) && rp_ready(s, nrhs))) {
            // interpolation products around a grid product (filter / transform forms): P inside
            // the staged W product (k_minres2_ph + k_spmv_w_staged_p), B = k_minres2_bh + k_minres2_bv
            const int np = (n + RL_THREADS - 1) / RL_THREADS;
            const size_t needp = (size_t)nrhs * (RL_RP_PCW + 1 + 3 * (size_t)np);
            if (s->rp_pp_cap < needp) {
                if (s->rp_pp) RL_HIP(hipFree(s->rp_pp));
                s->rp_pp = nullptr;
                s->rp_pp_cap = 0;
                RL_HIP(hipMalloc((void**)&s->rp_pp, needp * sizeof(double)));
                s->rp_pp_cap = needp;
            }
            RL_HIP(hipMemsetAsync(s->rp_pp, 0, needp * sizeof(double), st));
            mb.fuse_p = nrhs;
            mb.pc = s->rp_pp;
            mb.coef = s->rp_pp + (size_t)nrhs * RL_RP_PCW;
            mb.np = np;
            mb.partA[0] = mb.coef + nrhs;
            mb.partA[1] = mb.partA[0] + (size_t)nrhs * np;
            mb.partC = mb.partA[1] + (size_t)nrhs * np;
            mb.nrmB = mb.partB;               // (k_minres2_bv's partial norms, one per solver block)
            mb.nrm_n = nblk;
            trace_once("minres round: P inside the staged W product (k_minres2_ph, k_spmv_w_staged_p)");
        }
        RL_LAUNCH(k_minres2_init, grid, blk, 0, st, Bi, n, (const double*)w.part[0], mb);
        if (mb.poly_part != nullptr)        // projection of W^T y_0 for the first round's P
            RL_LAUNCH(k_poly_project_rows, grid, blk, 3 * RL_SOLVER_THREADS * sizeof(double), st,
                      Bi, n, mb);
        RL_TRY(active_count(w, nrhs, st, &active));
        // x lags one round behind: after `done` rounds it holds iterate done - 1.
        // The first round runs eagerly so that graph replays of per_graph rounds
        // land on done = 1 + j * per_graph, i.e. on the reference's check points
        if (active > 0) {
            RL_TRY(minres2_round(s, mb, nrhs, n, nblk, 1, rtol, maxiter, st));
            done = 1;
        }
        // graph replays start at round 2 + j * per2: the same parity every time
        // as long as per2 is even
        int per2 = 0;
        for (int d = 2; d <= 10; d += 2)
            if (check_every == 0 || check_every % d == 0) per2 = d;
        if (use_graph && active > 0 && per2 > 0) {
            RL_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = RL_OK;
            for (int k = 0; k < per2 && rc == RL_OK; ++k)
                rc = minres2_round(s, mb, nrhs, n, nblk, 2 + k, rtol, maxiter, st);
            hipError_t e = hipStreamEndCapture(st, &guard.graph);
            if (rc != RL_OK) return rc;
            RL_HIP(e);
            RL_HIP(hipGraphInstantiate(&guard.exec, guard.graph, nullptr, nullptr, 0));
        }
        // The host does not wait for the count of a replay before it launches the next
        // one: the count of replay j is read while replay j + 1 runs.  Submitting a
        // replay (per2 rounds of five to seven kernels) takes the host 1.5-2 ms, during
        // which the GPU sat idle when every replay ended in a synchronisation (C5, 129
        // systems: 3.82 ms per round against the kernels' 3.6); the price is one replay
        // of rounds for systems that have all stopped at the end of a solve -- P and B
        // return at once for them, the operator's kernels run (14 ms at C5) -- or next
        // to nothing in the polynomial rounds, which have no operator kernel.
        const bool lagged = guard.exec != nullptr;
        if (lagged && !s->pin_count) {
            RL_HIP(hipHostMalloc((void**)&s->pin_count, 2 * sizeof(int), hipHostMallocDefault));
            for (hipEvent_t& e : s->count_ev)
                RL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        bool pending[2] = {false, false};
        // (whatever way this block is left -- loop exit or an error return --, no copy
        // into the handle's pinned words stays in flight)
        struct Drain {
            hipEvent_t* ev;
            bool* pending;
            ~Drain() {
                for (int i = 0; i < 2; ++i)
                    if (pending[i]) (void)hipEventSynchronize(ev[i]);
            }
        } drain{s->count_ev, pending};
        int slot = 0;
        while (done <= maxiter && active > 0) {
            if (guard.exec) {
                RL_HIP(hipGraphLaunch(guard.exec, st));
                done += per2;
            } else {
                RL_TRY(minres2_round(s, mb, nrhs, n, nblk, done + 1, rtol, maxiter, st));
                done += 1;
            }
            const bool check = check_every > 0 && (done - 1) % check_every == 0;
            if (check)
                RL_TRY(residual_check(s, w, Bi, Xi, mb.q, nrhs, n, nblk, tol, 1, st));
            if (lagged && !check) {
                RL_LAUNCH(k_count_active, dim3(1), dim3(64), 0, st, w.I, nrhs, w.count);
                RL_HIP(hipMemcpyAsync(&s->pin_count[slot], w.count, sizeof(int),
                                      hipMemcpyDeviceToHost, st));
                RL_HIP(hipEventRecord(s->count_ev[slot], st));
                pending[slot] = true;
                const int other = slot ^ 1;
                if (pending[other]) {
                    RL_HIP(hipEventSynchronize(s->count_ev[other]));
                    active = s->pin_count[other];
                    pending[other] = false;
                }
                slot = other;
            } else if (check || guard.exec || (done - 1) % 10 == 0) {
                RL_TRY(active_count(w, nrhs, st, &active));        // (synchronises: nothing pending)
                pending[0] = pending[1] = false;
            }
        }
        RL_TRY(residual_check(s, w, Bi, Xi, mb.q, nrhs, n, nblk, tol, 0, st));
    } else if (method == RL_MINRES) {
        MinresBufs mb;
        mb.tri[0] = w.vec[0]; mb.tri[1] = w.vec[1]; mb.tri[2] = w.vec[2];
        mb.w[0] = w.vec[3]; mb.w[1] = w.vec[4];
        mb.v = w.vec[5];
        mb.q = w.vec[6];
        mb.x = Xi;
        mb.S[0] = w.S[0]; mb.S[1] = w.S[1];
        mb.I = w.I;
        mb.giter = w.count + 1;
        mb.lanczos = nullptr;
        mb.lanczos_cap = 0;
        if (lanczos_out != nullptr) {
            const size_t bytes = (size_t)nrhs * lanczos_cap * 2 * sizeof(double);
            if (s->lanczos_bytes < bytes) {
                if (s->lanczos_buf) RL_HIP(hipFree(s->lanczos_buf));
                s->lanczos_buf = nullptr;
                s->lanczos_bytes = 0;
                RL_HIP(hipMalloc((void**)&s->lanczos_buf, bytes));
                s->lanczos_bytes = bytes;
            }
            w.lanczos = s->lanczos_buf;
            RL_HIP(hipMemsetAsync(w.lanczos, 0, bytes, st));
            mb.lanczos = w.lanczos;
            mb.lanczos_cap = lanczos_cap;
        }
        RL_LAUNCH(k_minres_init, grid, blk, 0, st, Bi, n, (const double*)w.part[0], mb);
        RL_TRY(active_count(w, nrhs, st, &active));
        if (use_graph && active > 0) {
            RL_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = RL_OK;
            for (int k = 0; k < per_graph && rc == RL_OK; ++k)
                rc = minres_iteration(s, mb, w, nrhs, n, nblk, rtol, maxiter, st);
            hipError_t e = hipStreamEndCapture(st, &guard.graph);
            if (rc != RL_OK) return rc;
            RL_HIP(e);
            RL_HIP(hipGraphInstantiate(&guard.exec, guard.graph, nullptr, nullptr, 0));
        }
        while (done < maxiter && active > 0) {
            if (guard.exec) {
                RL_HIP(hipGraphLaunch(guard.exec, st));
                done += per_graph;
            } else {
                RL_TRY(minres_iteration(s, mb, w, nrhs, n, nblk, rtol, maxiter, st));
                done += 1;
            }
            const bool check = check_every > 0 && done % check_every == 0;
            // the free slot of the rotating triple is scratch for the check
            if (check)
                RL_TRY(residual_check(s, w, Bi, Xi, mb.q, nrhs, n, nblk, tol, 1, st));
            if (check || guard.exec || done % 10 == 0)
                RL_TRY(active_count(w, nrhs, st, &active));
        }
        RL_TRY(residual_check(s, w, Bi, Xi, mb.q, nrhs, n, nblk, tol, 0, st));
    } else {
        double *r = w.vec[0], *p = w.vec[1], *scratch = w.vec[3];
        RL_LAUNCH(k_cg_init, grid, blk, 0, st, Bi, n, (const double*)w.part[0], Xi, r, p, w.S[0],
                  w.I, rtol);
        RL_TRY(active_count(w, nrhs, st, &active));
        // first iteration eagerly (its head skips the rho update), the rest from a graph
        if (active > 0) {
            RL_TRY(cg_iteration(s, w, Xi, nrhs, n, nblk, 1, maxiter, st));
            done = 1;
            if (check_every == 1)
                RL_TRY(residual_check(s, w, Bi, Xi, scratch, nrhs, n, nblk, tol, 1, st));
            RL_TRY(active_count(w, nrhs, st, &active));
        }
        if (use_graph && active > 0 && per_graph > 1) {
            RL_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = cg_iteration(s, w, Xi, nrhs, n, nblk, 0, maxiter, st);
            hipError_t e = hipStreamEndCapture(st, &guard.graph);
            if (rc != RL_OK) return rc;
            RL_HIP(e);
            RL_HIP(hipGraphInstantiate(&guard.exec, guard.graph, nullptr, nullptr, 0));
        }
        while (done <= maxiter && active > 0) {
            if (guard.exec)
                RL_HIP(hipGraphLaunch(guard.exec, st));
            else
                RL_TRY(cg_iteration(s, w, Xi, nrhs, n, nblk, 0, maxiter, st));
            done += 1;
            const bool check = check_every > 0 && done % check_every == 0;
            if (check)
                RL_TRY(residual_check(s, w, Bi, Xi, scratch, nrhs, n, nblk, tol, 1, st));
            if (check || done % 10 == 0 || done > maxiter)
                RL_TRY(active_count(w, nrhs, st, &active));
        }
        RL_TRY(residual_check(s, w, Bi, Xi, scratch, nrhs, n, nblk, tol, 0, st));
    }
    if (s->permuted) permute_rows(s, Xi, X, nrhs, 1, st);
    RL_HIP(hipGetLastError());
    RL_HIP(hipStreamSynchronize(st));
    std::vector<int> hI((size_t)nrhs * I_NFIELDS);
    std::vector<double> hR((size_t)nrhs);
    RL_HIP(hipMemcpy(hI.data(), w.I, hI.size() * sizeof(int), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(hR.data(), w.resid, hR.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (lanczos_out != nullptr && w.lanczos != nullptr)
        RL_HIP(hipMemcpy(lanczos_out, w.lanczos,
                         (size_t)nrhs * lanczos_cap * 2 * sizeof(double),
                         hipMemcpyDeviceToHost));
    for (int r = 0; r < nrhs; ++r) {
        if (iters_out) iters_out[r] = hI[(size_t)r * I_NFIELDS + I_ITN];
        if (istop_out) istop_out[r] = hI[(size_t)r * I_NFIELDS + I_ISTOP];
        if (resid_out) resid_out[r] = hR[r];
    }
    return RL_OK;
}

// ---------------------------------------------------------------------------
// Direct solves through the polynomial form (rl_direct.h): K~ = F M F^T + E
// ---------------------------------------------------------------------------
#define RL_DZ_NBLK 120          // partial sums per system of the residual norms

// sum_k a[k] b[k] with four running sums (the compiler keeps them in two vector registers:
// a plain reduction loop is not vectorised without -ffast-math)
static inline double dz_dot(const double* a, const double* b, int n) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 3 < n; k += 4) {
        s0 += a[k] * b[k];
        s1 += a[k + 1] * b[k + 1];
        s2 += a[k + 2] * b[k + 2];
        s3 += a[k + 3] * b[k + 3];
    }
    for (; k < n; ++k) s0 += a[k] * b[k];
    return (s0 + s1) + (s2 + s3);
}
// host threads for the independent parts of the factorisation (a few hundred columns each)
static int dz_threads() {
    static int n = 0;
    if (n == 0) {
        // (up to 8 on a small host, a quarter of a large one's logical processors up to 32: the
        // MI355X box has 256 of them and eight ranks to share them)
        unsigned hc = std::thread::hardware_concurrency();
        n = (int)std::max(1u, hc <= 32u ? std::min(hc ? hc : 1u, 8u) : std::min(hc / 4u, 32u));
    }
    return n;
}
template <class F>
static void dz_parallel(int count, int min_per_thread, F body) {
    const int nt = std::max(1, std::min(dz_threads(), count / std::max(1, min_per_thread)));
    if (nt <= 1) {
        body(0, 1);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(body, t, nt);
    body(0, nt);
    for (std::thread& th : pool) th.join();
}
// in-place Cholesky factor (lower, row-major n x n; the strict upper part is left alone);
// false when a pivot is not positive.  Row by row, every entry ONE dot product of two finished
// row prefixes -- in panels of 64 columns: a panel's diagonal rows on the calling thread, then its
// columns of all rows below spread over threads (the same dot products in the same order
// whatever the thread count; 2.4 GFLOP at n = 1920, most of a parameter update's host time
// when done on one thread).
static bool dz_chol(std::vector<double>& a, int n) {
    const int PW = 64;
    for (int k0 = 0; k0 < n; k0 += PW) {
        const int k1 = std::min(n, k0 + PW);
        for (int i = k0; i < k1; ++i) {
            double* ai = a.data() + (size_t)i * n;
            for (int j = k0; j <= i; ++j) {
                const double* aj = a.data() + (size_t)j * n;
                const double s = ai[j] - dz_dot(ai, aj, j);
                if (j < i) {
                    ai[j] = s / aj[j];
                } else {
                    if (!(s > 0.0) || !std::isfinite(s)) return false;
                    ai[i] = std::sqrt(s);
                }
            }
        }
        dz_parallel(n - k1, 16, [&](int first, int step) {
            for (int i = k1 + first; i < n; i += step) {
                double* ai = a.data() + (size_t)i * n;
                for (int j = k0; j < k1; ++j) {
                    const double* aj = a.data() + (size_t)j * n;
                    ai[j] = (ai[j] - dz_dot(ai, aj, j)) / aj[j];
                }
            }
        });
    }
    return true;
}
// xt = (L^-1)^T, i.e. xt[c][i] = (L^-1)[i][c] (L lower, row-major; row c of xt is the solution
// of L x = e_c, zero before position c): the n columns are independent -- spread over threads
static void dz_tri_inverse_t(const std::vector<double>& L, int n, std::vector<double>& xt) {
    xt.assign((size_t)n * n, 0.0);
    dz_parallel(n, 32, [&](int first, int step) {
        for (int c = first; c < n; c += step) {
            double* x = xt.data() + (size_t)c * n;
            x[c] = 1.0 / L[(size_t)c * n + c];
            for (int i = c + 1; i < n; ++i) {
                const double* li = L.data() + (size_t)i * n;
                x[i] = -dz_dot(li + c, x + c, i - c) / li[i];
            }
        }
    });
}
// x = L^-1 (lower, row-major; x's strict upper part zero) -- the small per-output factors
static void dz_tri_inverse(const std::vector<double>& L, int n, std::vector<double>& x) {
    std::vector<double> xt;
    dz_tri_inverse_t(L, n, xt);
    x.assign((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c)
        for (int i = c; i < n; ++i) x[(size_t)i * n + c] = xt[(size_t)c * n + i];
}

// (F / part: another table and another partial-sum buffer than the handle's own -- the
// high-rank preconditioner projects on 48 columns of ITS table at a time)
template <int R>
static void rp_project_plain(rl_ski* s, const double* Xp, int nvec, hipStream_t st,
                             const double* Ftab = nullptr, double* partbuf = nullptr) {
    rl_gridop* g = s->g;
    const double* F_ = Ftab ? Ftab : (const double*)s->rp_F;
    double* part_ = partbuf ? partbuf : s->rp_part;
    const double* beta_ = (const double*)g->lr_beta;       // (read by the table-free variants only)
    constexpr int NT = (R + 15) / 16;
    const RpFuse nofz{nullptr, nullptr, nullptr};
    if (nvec <= RL_RP_VG + 1 && (nvec <= RL_RP_VG || nvec % RL_RP_VG == 1) && !s->kn.no_rp_small) {
        const size_t lds1 = (((size_t)16 * NT + RL_RP_VG) * RL_RP_LD + RL_RP_TILE) * sizeof(double);
        RL_LAUNCH((k_rp_project1<R, false>), dim3(s->rp_nruns), dim3(256), lds1, st, Xp, s->n, nvec,
                  F_, (const int*)s->rp_runs, part_, (int*)nullptr,
                  (const int*)s->W4_base, (const double*)s->W4_w, g->m, beta_, nofz);
        return;
    }
    const size_t lds = (((size_t)16 * NT + 2 * RL_RP_VG) * RL_RP_LD + RL_RP_TILE) * sizeof(double);
    const int vblk = RL_RP_NG(R) * RL_RP_VG;
    RL_LAUNCH((k_rp_project<R, false>), dim3(8 * ((s->rp_nruns + 7) / 8) * ((nvec + vblk - 1) / vblk)),
              dim3(256), lds, st, Xp, s->n, nvec, F_, (const int*)s->rp_runs,
              s->rp_nruns, part_, (int*)nullptr, (const int*)s->W4_base, (const double*)s->W4_w,
              g->m, beta_, nofz);
}
// Yp = F zhat + diag (.) X2
// (Ftab: expand from THAT table -- a block of the high-rank preconditioner's -- instead of the
// handle's own / the rows computed on the fly)
template <int R>
static void rp_expand_plain(rl_ski* s, const double* zhat, double* Yp, int nvec, const double* diag,
                            const double* X2, hipStream_t st, const double* Ftab = nullptr) {
    rl_gridop* g = s->g;
    const RpPFuse nopf{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (Ftab != nullptr) {
        RL_LAUNCH((k_rp_expand<R, false, false, true>), dim3((s->n + 255) / 256), dim3(256), 0, st, zhat,
                  Ftab, s->n, nvec, g->D, (const int*)s->rp_out_end, Yp, diag, X2,
                  s->kn.rp_stagger, (const int*)s->W4_base, (const double*)s->W4_w, g->m,
                  (const double*)g->lr_beta, nopf);
        return;
    }
    if (s->kn.rp_fly & 1)
        RL_LAUNCH((k_rp_expand<R, true, false, true>), dim3((s->n + 255) / 256), dim3(256), 0, st, zhat,
                  (const double*)s->rp_F, s->n, nvec, g->D, (const int*)s->rp_out_end, Yp, diag, X2,
                  s->kn.rp_stagger, (const int*)s->W4_base, (const double*)s->W4_w, g->m,
                  (const double*)g->lr_beta, nopf);
    else
        RL_LAUNCH((k_rp_expand<R, false, false, true>), dim3((s->n + 255) / 256), dim3(256), 0, st, zhat,
                  (const double*)s->rp_F, s->n, nvec, g->D, (const int*)s->rp_out_end, Yp, diag, X2,
                  s->kn.rp_stagger, (const int*)s->W4_base, (const double*)s->W4_w, g->m,
                  (const double*)g->lr_beta, nopf);
}
#define RL_DZ_RANKS(CALL)                                                               \
    switch (s->g->lr_r) {                                                                \
        case 24: CALL(24); break;                                                        \
        case 32: CALL(32); break;                                                        \
        case 36: CALL(36); break;                                                        \
        case 40: CALL(40); break;                                                        \
        case 48: CALL(48); break;                                                        \
        default: return fail(RL_EINVAL, "direct solve: bad basis size");                 \
    }

// The host's part of a factorisation (rl_direct.h): from the per-output Gram matrices U_d = F_d^T F_d on
// the unnormalised basis, the normalisation nu, the noise levels, the couplings B_q and the rows'
// coefficient matrices C_q at basis size R -- the scaled solve map Zs (D R x D R, symmetric), log det of
// F M F^T + E and the extreme Cholesky pivots of S.  false (with a reason) when a factor breaks down.
static bool dz_host_map(int D, int R, int Q, const double* U, const double* nu, const std::vector<double>& eps,
                        const std::vector<int>& rows, const double* hB, const double* hC,
                        std::vector<double>& Zs, double* logdet_out, double* pmin_out, double* pmax_out,
                        const char** why, std::vector<double>* Zh_t = nullptr) {
    const int Dr = D * R;
    // G_d = nu nu^T (.) U_d / eps_d = L_d L_d^T; Li_d = L_d^-1
    std::vector<std::vector<double>> L(D), Li(D);
    for (int d = 0; d < D; ++d) {
        L[d].assign((size_t)R * R, 0.0);
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < R; ++j) {
                const double u = 0.5 * (U[((size_t)d * R + i) * R + j] + U[((size_t)d * R + j) * R + i]);
                L[d][(size_t)i * R + j] = nu[i] * nu[j] * u / eps[d];
            }
        if (!dz_chol(L[d], R)) { *why = "an output has too few (or degenerate) rows for the basis"; return false; }
        for (int i = 0; i < R; ++i)
            for (int j = i + 1; j < R; ++j) L[d][(size_t)i * R + j] = 0.0;
        dz_tri_inverse(L[d], R, Li[d]);
    }
    // S = I + L^T M L, M_ab = sum_q B_q[a][b] C_q  (C symmetrised).  The D (D + 1) / 2 blocks
    // (a, b <= a) are independent: dealt to the threads in turn (not a thread per a: output D - 1
    // has D blocks, output 0 one); the inner loops run along rows (k ascending per entry, as a
    // dot product over k would).
    std::vector<double> S((size_t)Dr * Dr, 0.0);
    std::vector<std::pair<int, int>> blocks;
    for (int a = 0; a < D; ++a)
        for (int b = 0; b <= a; ++b) blocks.emplace_back(a, b);
    dz_parallel((int)blocks.size(), 1, [&](int first, int step) {
      std::vector<double> Mab((size_t)R * R), T1((size_t)R * R), A((size_t)R * R);
      for (size_t pb = first; pb < blocks.size(); pb += step) {
            const int a = blocks[pb].first, b = blocks[pb].second;
            std::fill(Mab.begin(), Mab.end(), 0.0);
            bool any = false;
            for (int q = 0; q < Q; ++q) {
                const double bq = 0.5 * (hB[((size_t)q * D + a) * D + b] + hB[((size_t)q * D + b) * D + a]);
                if (bq == 0.0) continue;
                any = true;
                const double* C = hC + (size_t)q * R * R;
                for (int i = 0; i < R; ++i)
                    for (int j = 0; j < R; ++j)
                        Mab[(size_t)i * R + j] += bq * 0.5 * (C[(size_t)i * R + j] + C[(size_t)j * R + i]);
            }
            if (!any) continue;
            // T1 = M_ab L_b   (L_b lower: row k of L_b has columns <= k)
            std::fill(T1.begin(), T1.end(), 0.0);
            for (int i = 0; i < R; ++i) {
                double* t = T1.data() + (size_t)i * R;
                for (int k = 0; k < R; ++k) {
                    const double mk = Mab[(size_t)i * R + k];
                    const double* l = L[b].data() + (size_t)k * R;
                    for (int j = 0; j <= k; ++j) t[j] += mk * l[j];
                }
            }
            // A_ab = L_a^T T1
            std::fill(A.begin(), A.end(), 0.0);
            for (int k = 0; k < R; ++k) {
                const double* t = T1.data() + (size_t)k * R;
                for (int i = 0; i <= k; ++i) {
                    const double lk = L[a][(size_t)k * R + i];
                    double* o = A.data() + (size_t)i * R;
                    for (int j = 0; j < R; ++j) o[j] += lk * t[j];
                }
            }
            for (int i = 0; i < R; ++i)
                for (int j = 0; j < R; ++j) {
                    const double acc = A[(size_t)i * R + j];
                    S[((size_t)a * R + i) * Dr + (size_t)b * R + j] = acc;
                    S[((size_t)b * R + j) * Dr + (size_t)a * R + i] = acc;
                }
      }
    });
    for (int i = 0; i < Dr; ++i)
        for (int j = 0; j < i; ++j) {
            const double v = 0.5 * (S[(size_t)i * Dr + j] + S[(size_t)j * Dr + i]);
            S[(size_t)i * Dr + j] = v;
            S[(size_t)j * Dr + i] = v;
        }
    for (int i = 0; i < Dr; ++i) S[(size_t)i * Dr + i] += 1.0;
    if (!dz_chol(S, Dr)) { *why = "I + L^T M L is not positive definite (M is not positive semi-definite to roundoff)"; return false; }
    double logdet = 0.0, pmin = 1e300, pmax = 0.0;
    for (int i = 0; i < Dr; ++i) {
        const double p = S[(size_t)i * Dr + i];
        logdet += 2.0 * std::log(p);
        pmin = std::min(pmin, p);
        pmax = std::max(pmax, p);
    }
    for (int d = 0; d < D; ++d) logdet += rows[d] * std::log(eps[d]);
    // Y = I - S^-1 = I - X^T X, X = chol(S)^-1: with Xt = X^T stored by rows (row c = column c
    // of X, zero before position c), S^-1[i][j] = Xt[i] . Xt[j] over positions >= max(i, j)
    std::vector<double> Xt;
    dz_tri_inverse_t(S, Dr, Xt);
    std::vector<double> Y((size_t)Dr * Dr, 0.0);
    dz_parallel(Dr, 32, [&](int first, int step) {
        for (int i = first; i < Dr; i += step) {
            const double* xi = Xt.data() + (size_t)i * Dr;
            for (int j = 0; j <= i; ++j) {
                const double* xj = Xt.data() + (size_t)j * Dr;
                const double v = (i == j ? 1.0 : 0.0) - dz_dot(xi + i, xj + i, Dr - i);
                Y[(size_t)i * Dr + j] = v;
            }
        }
    });
    for (int i = 0; i < Dr; ++i)
        for (int j = 0; j < i; ++j) Y[(size_t)j * Dr + i] = Y[(size_t)i * Dr + j];
    // Z_ab = Li_a^T Y_ab Li_b, scaled:  Zs = -(nu_i / eps_a) Z (nu_j / eps_b)
    Zs.assign((size_t)Dr * Dr, 0.0);
    dz_parallel((int)blocks.size(), 1, [&](int first, int step) {
      std::vector<double> T1((size_t)R * R), A((size_t)R * R);
      for (size_t pb = first; pb < blocks.size(); pb += step) {
            const int a = blocks[pb].first, b = blocks[pb].second;
            // T1 = Y_ab Li_b  (Li_b lower)
            std::fill(T1.begin(), T1.end(), 0.0);
            for (int i = 0; i < R; ++i) {
                double* t = T1.data() + (size_t)i * R;
                const double* y = Y.data() + ((size_t)a * R + i) * Dr + (size_t)b * R;
                for (int k = 0; k < R; ++k) {
                    const double yk = y[k];
                    const double* l = Li[b].data() + (size_t)k * R;
                    for (int j = 0; j <= k; ++j) t[j] += yk * l[j];
                }
            }
            std::fill(A.begin(), A.end(), 0.0);
            for (int k = 0; k < R; ++k) {
                const double* t = T1.data() + (size_t)k * R;
                for (int i = 0; i <= k; ++i) {
                    const double lk = Li[a][(size_t)k * R + i];
                    double* o = A.data() + (size_t)i * R;
                    for (int j = 0; j < R; ++j) o[j] += lk * t[j];
                }
            }
            for (int i = 0; i < R; ++i)
                for (int j = 0; j < R; ++j) {
                    const double v = -(nu[i] / eps[a]) * A[(size_t)i * R + j] * (nu[j] / eps[b]);
                    Zs[((size_t)a * R + i) * Dr + (size_t)b * R + j] = v;
                    Zs[((size_t)b * R + j) * Dr + (size_t)a * R + i] = v;
                }
      }
    });
    // The square-root factor's map (rl_ski_precond_sample), on request: with Q = E^-1/2 F L^-T
    // (orthonormal columns) the factorised matrix is E^1/2 (I + Q (S - I) Q^T) E^1/2 and, S = C C^T,
    //     I + Q (S - I) Q^T = B B^T,   B = I + Q (C - I) Q^T
    // so  E^1/2 B w = sqrt(eps) w + F_q [Zh F_q^T (w / sqrt(eps))],  Zh = nu L^-T (C - I) L^-1 nu
    // (block lower triangular, NOT symmetric: stored transposed, the layout the map kernels read).
    if (Zh_t != nullptr) {
        Zh_t->assign((size_t)Dr * Dr, 0.0);
        dz_parallel((int)blocks.size(), 1, [&](int first, int step) {
          std::vector<double> T1((size_t)R * R), A((size_t)R * R);
          for (size_t pb = first; pb < blocks.size(); pb += step) {
                const int a = blocks[pb].first, b = blocks[pb].second;       // a >= b
                // T1 = (C - I)_ab Li_b   (the diagonal block of C is lower triangular: S's strict
                // upper part still holds the symmetric matrix's entries -- masked here)
                std::fill(T1.begin(), T1.end(), 0.0);
                for (int i = 0; i < R; ++i) {
                    double* t = T1.data() + (size_t)i * R;
                    const double* c = S.data() + ((size_t)a * R + i) * Dr + (size_t)b * R;
                    const int kmax = a == b ? i : R - 1;
                    for (int k = 0; k <= kmax; ++k) {
                        const double ck = c[k] - (a == b && k == i ? 1.0 : 0.0);
                        const double* l = Li[b].data() + (size_t)k * R;
                        for (int j = 0; j <= k; ++j) t[j] += ck * l[j];
                    }
                }
                std::fill(A.begin(), A.end(), 0.0);
                for (int k = 0; k < R; ++k) {
                    const double* t = T1.data() + (size_t)k * R;
                    for (int i = 0; i <= k; ++i) {
                        const double lk = Li[a][(size_t)k * R + i];
                        double* o = A.data() + (size_t)i * R;
                        for (int j = 0; j < R; ++j) o[j] += lk * t[j];
                    }
                }
                for (int i = 0; i < R; ++i)
                    for (int j = 0; j < R; ++j)
                        (*Zh_t)[((size_t)b * R + j) * Dr + (size_t)a * R + i] = nu[i] * A[(size_t)i * R + j] * nu[j];
          }
        });
    }
    *logdet_out = logdet;
    *pmin_out = pmin;
    *pmax_out = pmax;
    return true;
}

// ---------------------------------------------------------------------------
// The larger preconditioner.  An operator NONE of whose rows is in the polynomial form (filter and
// transform kernels only: Matern) has no basis size of its own, and its rows' spectra fall off
// slowly: the 48 functions every handle generates leave conjugate gradients 1358 iterations at
// C5 (8 s; profiles/r06/pcg_probe_c5_matern.txt).  More functions cut that several times over --
// 220 iterations with 96, 77 with 144, 38 with 192 (0.39 s; profiles/r06/pcg_probe_c5_matern_hi*.txt)
// -- so such an operator of enough rows gets a basis of its own, of hz_rank() functions:
//   * orthonormal polynomials by the same recurrence as lr_make_basis, F = W Phi as an [R][n]
//     table (k_rp_build), the Gram matrices of its columns per output: once per handle;
//   * per parameter update C_q = Phi^T T_q Phi from ONE product of the row with the functions
//     (rl_gridop_mvm_top) and an R x R block of dot products, then dz_host_map at size D * R;
//   * per application the 48-function kernels on one block of 48 columns of the table at a time
//     (rp_project_plain<48> / rp_expand_plain<48> with a table argument) around the dense map
//     (k_hz_sums -> k_hz_map -> k_hz_collect, rl_direct.h).
// The factorisation is a PRECONDITIONER (rl_solve_pcg): nothing here is exact, and nothing else in
// the library reads this basis.  ("96" in names and comments below: the smallest such basis.)
// A factorisation uses the first blocks of the table that hold all but 1e-5 of every row: hz_try.
// ---------------------------------------------------------------------------
static_assert(RL_HZ_BLK == RL_RP_RMAX, "the blocks run the rank-48 kernels");
// basis size of this handle: the largest number of whole blocks of 48, at most RUNLMC_PRECOND_HI_RANK
// (default RL_HZ_RDEF) and at least two, that
//   * k_hz_map's slices of the coefficient rows are sized for (D * R <= 2048),
//   * the grid carries (the recurrence keeps its functions orthonormal well past m = 8 R; hz_basis
//     checks), and
//   * pays: the host's part of a parameter update grows as (D R)^3 -- measured on the MI355X box's
//     host at C5 (D = 10): 47 ms at R = 96, 90 ms at 144, 0.24 s at 192 on 8 threads -- against
//     conjugate-gradient iterations of ~7 us per thousand rows saved (C5: 1358 iterations with the
//     48 functions, 220 with 96, 77 with 144, 38 with 192).
// 0: none.
#define RL_HZ_RDEF 192
static int hz_rank(const rl_ski* s) {
    const int D = s->g->D;
    const int lim = RL_HZ_FS * RL_HZ_FMAX / D / RL_HZ_BLK * RL_HZ_BLK;
    for (int R = std::min(s->kn.precond_hi_rank / RL_HZ_BLK * RL_HZ_BLK, lim); R >= 2 * RL_HZ_BLK; R -= RL_HZ_BLK) {
        const double f = (double)D * R / 960.0;
        if (s->g->m >= 8 * R && (double)s->n >= 1e5 * f * f * f) return R;
    }
    return 0;
}
template <class T>
static int hz_grow(T** p, size_t count) {
    if (*p) RL_HIP(hipFree(*p));
    *p = nullptr;
    RL_HIP(hipMalloc((void**)p, count * sizeof(T)));
    return RL_OK;
}

// the square-root factor's map of the current factorisation on the device (empty: not asked for)
static int dz_upload_sample_map(rl_ski* s, const std::vector<double>& Zh) {
    s->dz_Zh_valid = false;
    if (Zh.empty()) return RL_OK;
    if (s->dz_Zh_cap < Zh.size()) {
        s->dz_Zh_cap = 0;
        RL_TRY(hz_grow(&s->dz_Zh, Zh.size()));
        s->dz_Zh_cap = Zh.size();
    }
    RL_HIP(hipMemcpy(s->dz_Zh, Zh.data(), Zh.size() * sizeof(double), hipMemcpyHostToDevice));
    s->dz_Zh_valid = true;
    return RL_OK;
}

// buffers of an application to nvec vectors (never inside a capture: rl_solve_pcg calls it first)
static int hz_reserve(rl_ski* s, int nvec) {
    const size_t cap = (size_t)std::max(nvec, RL_HZ_BLK);
    if (s->hz_vec_cap >= cap && s->hz_part != nullptr) return RL_OK;
    s->hz_vec_cap = 0;
    const size_t nb = (size_t)s->hz_R / RL_HZ_BLK;
    RL_TRY(hz_grow(&s->hz_part, nb * s->rp_nruns * cap * RL_HZ_BLK));
    RL_TRY(hz_grow(&s->hz_zhat, nb * (cap + 16) * s->g->D * RL_HZ_BLK));
    RL_TRY(hz_grow(&s->hz_tmp, cap * s->n));      // (the block-by-block expansion's)
    RL_TRY(hz_grow(&s->hz_S, cap * s->g->D * s->hz_R));
    RL_TRY(hz_grow(&s->hz_P, (size_t)RL_HZ_FS * cap * s->g->D * s->hz_R));
    s->hz_vec_cap = cap;
    return RL_OK;
}

// once per handle: the basis on the grid, its recurrence, F and the Gram matrices of F's columns
// per output.  hz_why != nullptr afterwards: the handle has no such basis.
static int hz_basis(rl_ski* s) {
    rl_gridop* g = s->g;
    if (s->hz_basis_tried) return RL_OK;
    s->hz_basis_tried = true;
    s->hz_why = "basis not built";
    const int m = g->m, n = s->n, D = g->D, R = hz_rank(s), NB = R / RL_HZ_BLK;
    if (R == 0) { s->hz_why = "no room for a larger basis at this many outputs"; return RL_OK; }
    s->hz_R = R;
    const int rows = (R + D - 1) / D * D;               // (products take whole vectors of D blocks)
    std::vector<double> phi((size_t)rows * m, 0.0), beta(R), nu(R);
    {
        std::vector<long double> prev(m, 0.0L), cur(m), nxt(m);
        const long double p0 = 1.0L / sqrtl((long double)m);
        for (int i = 0; i < m; ++i) cur[i] = p0;
        long double bj = 0.0L, nuj = p0;
        for (int j = 0; j < R; ++j) {
            for (int i = 0; i < m; ++i) phi[(size_t)j * m + i] = (double)cur[i];
            beta[j] = (double)(bj * bj);
            nu[j] = (double)nuj;
            long double nrm = 0.0L;
            for (int i = 0; i < m; ++i) {
                const long double si = -1.0L + 2.0L * i / (m - 1);
                nxt[i] = si * cur[i] - bj * prev[i];
                nrm += nxt[i] * nxt[i];
            }
            bj = sqrtl(nrm);
            if (!(bj > 0.0L)) { s->hz_why = "the larger basis is degenerate on this grid"; return RL_OK; }
            nuj /= bj;
            for (int i = 0; i < m; ++i) {
                prev[i] = cur[i];
                cur[i] = nxt[i] / bj;
            }
        }
    }
    // (the recurrence loses orthogonality when the degree nears 2 sqrt(m): the last function
    // against a few of the others says whether it did)
    {
        const int probe[] = {0, 1, R / 4, R / 2 - 1, R / 2, R - 3, R - 2};
        const double* last = phi.data() + (size_t)(R - 1) * m;
        double worst = std::fabs(dz_dot(last, last, m) - 1.0);
        for (int j : probe) worst = std::max(worst, std::fabs(dz_dot(last, phi.data() + (size_t)j * m, m)));
        if (!(worst < 1e-9)) { s->hz_why = "the larger basis is not orthonormal on this grid (too few points)"; return RL_OK; }
    }
    // (the table is R n doubles -- 1.5 GB at C5, 15 GB at n = 10^7: a handle that cannot spare
    // twice that keeps the 48 functions instead of failing its solves on an allocation)
    {
        size_t free_b = 0, total_b = 0;
        RL_HIP(hipMemGetInfo(&free_b, &total_b));
        const double need = 2.0 * (double)R * n * sizeof(double) + 2.0 * (double)rows * m * sizeof(double) + 1e9;
        if ((double)free_b < need) { s->hz_why = "not enough free device memory for the larger basis' table"; return RL_OK; }
    }
    RL_TRY(upload(&s->hz_phi, phi));
    RL_TRY(upload(&s->hz_beta, beta));
    s->hz_hnu = nu;
    RL_TRY(hz_grow(&s->hz_tphi, (size_t)rows * m));
    RL_TRY(hz_grow(&s->hz_C, (size_t)R * R));
    RL_TRY(hz_grow(&s->hz_F, (size_t)R * n));
    {
        std::vector<double> ones((size_t)n, 1.0);
        RL_TRY(upload(&s->hz_ones, ones));
    }
    hipStream_t st = nullptr;
    RL_LAUNCH(k_rp_build, dim3((n + 255) / 256), dim3(256), 0, st, (const int*)s->W4_base,
              (const double*)s->W4_w, n, m, R, (const double*)s->hz_beta, s->hz_F);
    RL_HIP(hipGetLastError());
    RL_TRY(hz_reserve(s, RL_HZ_BLK));
    // U_d[(rb, j)][(cb, v)] = sum over the runs c of output d of part[c][v][j]: half cb of the
    // table's columns as a batch of 48 vectors, projected on half rb
    s->hz_U.assign((size_t)D * R * R, 0.0);
    std::vector<double> part((size_t)s->rp_nruns * RL_HZ_BLK * RL_HZ_BLK);
    for (int cb = 0; cb < NB; ++cb)
        for (int rb = 0; rb < NB; ++rb) {
            rp_project_plain<RL_HZ_BLK>(s, s->hz_F + (size_t)cb * RL_HZ_BLK * n, RL_HZ_BLK, st,
                                        s->hz_F + (size_t)rb * RL_HZ_BLK * n, s->hz_part);
            RL_HIP(hipGetLastError());
            RL_HIP(hipMemcpy(part.data(), s->hz_part, part.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int d = 0; d < D; ++d)
                for (int c = s->h_run_ptr[d]; c < s->h_run_ptr[d + 1]; ++c)
                    for (int v = 0; v < RL_HZ_BLK; ++v)
                        for (int j = 0; j < RL_HZ_BLK; ++j)
                            s->hz_U[((size_t)d * R + rb * RL_HZ_BLK + j) * R + cb * RL_HZ_BLK + v] +=
                                part[((size_t)c * RL_HZ_BLK + v) * RL_HZ_BLK + j];
        }
    s->hz_why = nullptr;
    return RL_OK;
}

// Is the larger preconditioner this operator's?  (Not every row in the polynomial form, rows
// enough -- RUNLMC_PRECOND_HI_MIN, 10^5: below, a conjugate-gradient iteration costs tens of
// microseconds and the 48 functions' cheaper update wins -- and a basis size that pays, hz_rank.)
static bool hz_wanted(const rl_ski* s) {
    const rl_gridop* g = s->g;
    if (s->kn.no_precond_hi || g->lr_ok || (g->lr_np > 0 && s->kn.no_precond_hi_mixed)) return false;
    return hz_rank(s) > 0 && (double)s->n >= (double)s->kn.precond_hi_min;
}

// the factorisation at the current parameters: *ok = false with a reason leaves the handle to the
// 48-function one
static int hz_try(rl_ski* s, const std::vector<double>& eps, const std::vector<int>& rows, bool* ok,
                  const char** why) {
    rl_gridop* g = s->g;
    *ok = false;
    RL_TRY(hz_basis(s));
    if (s->hz_why != nullptr) { *why = s->hz_why; return RL_OK; }
    const int D = g->D, m = g->m, Q = g->Q, RT = s->hz_R;
    int R = RT;
    const int nv = (RT + D - 1) / D;
    hipStream_t st = nullptr;
    std::vector<double> hCT((size_t)Q * RT * RT), one((size_t)RT * RT);
    std::vector<double> capb((size_t)Q * (RT / RL_HZ_BLK), 0.0);      // captured by the first 48 (k + 1) functions
    const unsigned nb = ((RT + RL_XD_A - 1) / RL_XD_A) * ((RT + RL_XD_B - 1) / RL_XD_B);
    for (int q = 0; q < Q; ++q) {
        RL_TRY(rl_gridop_mvm_top(g, q, s->hz_phi, s->hz_tphi, nv, st));
        RL_LAUNCH(k_cross_dots_tiled, dim3(nb, 1), dim3(RL_SOLVER_THREADS), RL_SOLVER_THREADS * sizeof(double),
                  st, (const double*)s->hz_phi, (const double*)s->hz_tphi, RT, m, s->hz_C);
        RL_HIP(hipGetLastError());
        RL_HIP(hipMemcpy(one.data(), s->hz_C, one.size() * sizeof(double), hipMemcpyDeviceToHost));
        const double t0 = g->h_tops.size() > (size_t)q * m ? g->h_tops[(size_t)q * m] : 0.0;
        double tr = 0.0;
        double* dst = hCT.data() + (size_t)q * RT * RT;
        for (int i = 0; i < RT; ++i) {
            tr += one[(size_t)i * RT + i];
            if ((i + 1) % RL_HZ_BLK == 0) capb[(size_t)q * (RT / RL_HZ_BLK) + i / RL_HZ_BLK] = t0 > 0.0 ? tr / (t0 * m) : 0.0;
            for (int j = 0; j < RT; ++j) {
                dst[(size_t)i * RT + j] = 0.5 * (one[(size_t)i * RT + j] + one[(size_t)j * RT + i]);
                if (!std::isfinite(dst[(size_t)i * RT + j])) { *why = "a top row's projection is not finite"; return RL_OK; }
            }
        }
        if (!(t0 > 0.0 && tr / (t0 * m) >= 0.8)) {
            *why = "the larger subspace holds less than 0.8 of a row's spectrum either";
            return RL_OK;
        }
    }
    if (getenv("RUNLMC_TRACE") != nullptr && !s->hz_traced) {
        s->hz_traced = true;
        for (int q = 0; q < Q; ++q) {
            fprintf(stderr, "[runlmc] larger basis: row %d (form %d) holds", q, (int)g->top_form.size() > q ? g->top_form[q] : -1);
            for (int k = 0; k < RT / RL_HZ_BLK; ++k) fprintf(stderr, " %.6f", capb[(size_t)q * (RT / RL_HZ_BLK) + k]);
            fprintf(stderr, " of its trace in the first 48, 96 ... functions\n");
        }
    }
    // How many of the functions this factorisation uses.  Iterations fall with the basis, the
    // host's part of the update grows as (D R)^3: where that part is small (D R <= 960: under
    // 50 ms) all of them; else the fewest blocks that leave at most 1e-5 of every row's trace
    // outside (the tail is what conjugate gradients have to resolve), else all.  Measured --
    // C5 'mix' (Matern gamma = 1 next to four smooth rows: < 5e-7 outside 96 functions): 11
    // iterations / 0.13 s a step with 96, 6 / 0.13 with 144, 4 / 0.16 with 192 (59 with the
    // operator's own 36); C5 matern (gamma up to 10: 1.8e-4 outside 96, 2.3e-5 outside 192): 220 /
    // 77 / 38 iterations, 1.6 / 0.76 / 0.50 s; a D = 3 fit with a gamma = 20 row (1.4e-3 outside 96):
    // 98 / 50 / 23 iterations, 73 / 58 / 50 ms a step.
    if ((double)D * RT > 960.0)
        for (int k = 2; k < RT / RL_HZ_BLK; ++k) {
            double worst = 1.0;
            for (int q = 0; q < Q; ++q) worst = std::min(worst, capb[(size_t)q * (RT / RL_HZ_BLK) + k - 1]);
            if (worst >= 1.0 - 1e-5) { R = k * RL_HZ_BLK; break; }
        }
    if (s->kn.precond_hi_use > 0) R = std::max(2 * RL_HZ_BLK, std::min(RT, s->kn.precond_hi_use / RL_HZ_BLK * RL_HZ_BLK));
    std::vector<double> hC((size_t)Q * R * R);
    for (int q = 0; q < Q; ++q)
        for (int i = 0; i < R; ++i)
            std::memcpy(hC.data() + ((size_t)q * R + i) * R, hCT.data() + ((size_t)q * RT + i) * RT, (size_t)R * sizeof(double));
    std::vector<double> Zs;
    double logdet = 0.0, pmin = 1.0, pmax = 1.0;
    std::vector<double> Usub;
    const double* U = s->hz_U.data();
    if (R != RT) {                              // the leading R x R corner of every output's block
        Usub.resize((size_t)D * R * R);
        for (int d = 0; d < D; ++d)
            for (int i = 0; i < R; ++i)
                std::memcpy(Usub.data() + ((size_t)d * R + i) * R, s->hz_U.data() + ((size_t)d * RT + i) * RT,
                            (size_t)R * sizeof(double));
        U = Usub.data();
    }
    std::vector<double> Zh;
    if (!dz_host_map(D, R, Q, U, s->hz_hnu.data(), eps, rows, g->lr_hB.data(), hC.data(), Zs,
                     &logdet, &pmin, &pmax, why, s->dz_want_sample ? &Zh : nullptr))
        return RL_OK;
    RL_TRY(dz_upload_sample_map(s, Zh));
    for (double v : Zs)
        if (!std::isfinite(v)) { *why = "the solve map is not finite"; return RL_OK; }
    if (s->dz_Zt_cap < Zs.size()) {
        s->dz_Zt_cap = 0;
        RL_TRY(hz_grow(&s->dz_Zt, Zs.size()));
        s->dz_Zt_cap = Zs.size();
    }
    RL_HIP(hipMemcpy(s->dz_Zt, Zs.data(), Zs.size() * sizeof(double), hipMemcpyHostToDevice));
    s->dz_logdet = logdet;
    s->dz_cond = (pmax / pmin) * (pmax / pmin);
    s->hz_Ruse = R;
    *ok = true;
    return RL_OK;
}

// out = P^-1 in through the 96-function factorisation (dz_apply's other branch)
// (map / diag: another dense map and another diagonal than the solve's -- the square-root factor)
static int hz_apply(rl_ski* s, const double* in, double* out, int nvec, hipStream_t st,
                    const double* map = nullptr, const double* diag = nullptr) {
    if (map == nullptr) map = s->dz_Zt;
    if (diag == nullptr) diag = s->dz_inv;
    rl_gridop* g = s->g;
    const int D = g->D, n = s->n, Dr = D * s->hz_Ruse, NB = s->hz_Ruse / RL_HZ_BLK;
    RL_TRY(hz_reserve(s, nvec));
    for (int k = 0; k < NB; ++k)
        rp_project_plain<RL_HZ_BLK>(s, in, nvec, st, s->hz_F + (size_t)k * RL_HZ_BLK * n,
                                    s->hz_part + (size_t)k * s->rp_nruns * nvec * RL_HZ_BLK);
    const dim3 egrid((Dr + 255) / 256, nvec);
    RL_LAUNCH(k_hz_sums, egrid, dim3(256), 0, st, (const double*)s->hz_part, (const int*)s->rp_run_ptr,
              s->rp_nruns, nvec, D, NB, s->hz_S);
    const int per = (Dr + RL_HZ_FS - 1) / RL_HZ_FS;
    RL_LAUNCH(k_hz_map, dim3((Dr + 255) / 256, (nvec + RL_HZ_VB - 1) / RL_HZ_VB, RL_HZ_FS), dim3(256),
              (size_t)per * RL_HZ_VB * sizeof(double), st, (const double*)s->hz_S, map,
              nvec, Dr, s->hz_P);
    if (!s->kn.precond_hi_passes && NB >= 3) {
        // ONE pass over the rows for all blocks, on the matrix cores (k_hz_expand_mm): the vectors
        // cross the fabric once -- in, out -- and the table once.  C5, 129 vectors: 1.83 ms for four
        // blocks against 4 x 0.64 ms block by block (solve 0.336 against 0.355 s); for TWO blocks
        // 1.45 against 2 x 0.63 ms -- those stay on the rank-48 kernel below.
        const int nvp = (nvec + 15) / 16 * 16, nvt = nvp / 16;
        RL_LAUNCH(k_hz_collect, dim3((Dr + 255) / 256, nvp), dim3(256), 0, st, (const double*)s->hz_P, nvec, D, NB,
                  RL_HZ_FS, s->hz_zhat, nvp);
        const dim3 xgrid((n + 127) / 128, (nvt + 8) / 9), xgrid2((n + 127) / 128, (nvt + 1) / 2);
        if (nvt <= 2)
            RL_LAUNCH((k_hz_expand_mm<2>), xgrid2, dim3(256), 0, st, (const double*)s->hz_zhat, (const double*)s->hz_F,
                      n, nvec, nvp, D, s->hz_Ruse, (const int*)s->rp_out_end, out, diag, in);
        else
            RL_LAUNCH((k_hz_expand_mm<9>), xgrid, dim3(256), 0, st, (const double*)s->hz_zhat, (const double*)s->hz_F,
                      n, nvec, nvp, D, s->hz_Ruse, (const int*)s->rp_out_end, out, diag, in);
        RL_HIP(hipGetLastError());
        return RL_OK;
    }
    RL_LAUNCH(k_hz_collect, egrid, dim3(256), 0, st, (const double*)s->hz_P, nvec, D, NB, RL_HZ_FS, s->hz_zhat, 0);
    // (a pass over the rows per block, each adding to the one before through the noise term's
    // operand with a diagonal of ones: not in place -- the kernel's pointers are declared not
    // to alias -- but alternating between `out` and one more buffer so that the last lands in out)
    const double* src = in;
    for (int k = 0; k < NB; ++k) {
        double* dst = (NB - 1 - k) % 2 == 0 ? out : s->hz_tmp;
        rp_expand_plain<RL_HZ_BLK>(s, s->hz_zhat + (size_t)k * nvec * D * RL_HZ_BLK, dst, nvec,
                                   k == 0 ? diag : (const double*)s->hz_ones, src, st, s->hz_F + (size_t)k * RL_HZ_BLK * n);
        src = dst;
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// May this handle's operator be inverted through its polynomial form?  Runs the pending
// verification of the forms (whatever the batch gate says: the decision is the operator's,
// not a batch's).  `why` receives the reason when not.
static int dz_available(rl_ski* s, bool* ok, const char** why) {
    rl_gridop* g = s->g;
    *ok = false;
    *why = "";
    if (!s->extra.empty()) { *why = "kernels on several grids"; return RL_OK; }
    if (g->wide) { *why = "more than 16 outputs"; return RL_OK; }
    if (s->W4_base == nullptr || s->h_base.empty() || s->ngrid != g->D * g->m) {
        *why = "W is not a cubic interpolant of a 1-D grid"; return RL_OK;
    }
    if (!g->lr_try || g->kn.no_rp || (s->kn.rp_fly & 2)) { *why = "polynomial form switched off or grid not eligible"; return RL_OK; }
    if (s->n >= (1 << 28)) { *why = "n >= 2^28"; return RL_OK; }
    if (g->Q < 1) { *why = "no parameters set"; return RL_OK; }
    if (!s->has_noise || (int)s->h_noise.size() != s->n) { *why = "no noise set"; return RL_OK; }
    RL_HIP(hipSetDevice(g->device));
    RL_TRY(lr_ensure(g));
    // (a row outside the polynomial form -- filter or transform kernels -- does not end it here: its
    // projection on the subspace still makes a preconditioner, dz_ensure decides)
    if ((int)g->lr_hB.size() < g->Q * g->D * g->D) { *why = "no host copy of the couplings"; return RL_OK; }
    *ok = true;
    return RL_OK;
}

// builds (or rebuilds, after a parameter / noise update) the factorisation; *ok = false with
// a reason when the operator has no such form or the factorisation breaks down
static int dz_ensure(rl_ski* s, bool* ok, const char** why) {
    rl_gridop* g = s->g;
    RL_TRY(dz_available(s, ok, why));
    if (!*ok) { s->dz_valid = false; return RL_OK; }
    // basis size: the operator's own when some row is in the polynomial form, else the largest
    const bool exact = g->lr_ok;
    const int R = exact || g->lr_np > 0 ? g->lr_r : RL_LR_RMAX;
    const int D = g->D, n = s->n, Dr = D * R, Q = g->Q;
    if (s->dz_valid && s->dz_param_ver == g->param_ver && s->dz_noise_ver == s->noise_ver &&
        (s->dz_R == R || s->dz_hz))
        return RL_OK;
    if (s->dz_fail_why != nullptr && s->dz_fail_param_ver == g->param_ver && s->dz_fail_noise_ver == s->noise_ver) {
        *ok = false;                       // (decided for these parameters already)
        *why = s->dz_fail_why;
        return RL_OK;
    }
    // an operator that had no factorisation at its last parameter sets sits out a growing number
    // of updates before the next attempt (1, 3, 7 ... 31: the projections of its rows cost a
    // millisecond an update, and a fit's kernels do not become smooth from one step to the next)
    if (!exact && s->dz_fail_streak > 0 && s->dz_fail_why != nullptr) {
        if (s->dz_fail_skip > 0) {
            --s->dz_fail_skip;
            *ok = false;
            *why = s->dz_fail_why;
            s->dz_fail_param_ver = g->param_ver;
            s->dz_fail_noise_ver = s->noise_ver;
            return RL_OK;
        }
    }
    struct FailNote {                      // every "not available" below is remembered with its versions
        rl_ski* s; bool* ok; const char** why;
        ~FailNote() {
            if (!*ok) {
                s->dz_fail_why = *why;
                s->dz_fail_param_ver = s->g->param_ver;
                s->dz_fail_noise_ver = s->noise_ver;
                s->dz_fail_streak = std::min(s->dz_fail_streak + 1, 5);
                s->dz_fail_skip = (1 << s->dz_fail_streak) - 1;
            } else {
                s->dz_fail_why = nullptr;
                s->dz_fail_streak = 0;
                s->dz_fail_skip = 0;
            }
        }
    } note{s, ok, why};
    // The host's part is ~1.5 (D r)^3 multiply-adds per parameter update (2.5 ms at D r = 240, 8 ms at
    // 360, ~0.1 s at 768): past D r = 576 a SMALL system's Krylov solve is cheaper than its
    // factorisation (a round of a 10^4-row system is 20 us), so such operators keep the Krylov path
    s->dz_hz = false;
    if (Dr > 576 && (double)n < 1e5 * ((double)Dr / 576.0) * ((double)Dr / 576.0) * ((double)Dr / 576.0)) {
        s->dz_valid = false;
        *ok = false;
        *why = "D * rank is large for this few rows: the factorisation would cost more than the Krylov solve";
        return RL_OK;
    }
    s->dz_valid = false;
    *ok = false;
    hipStream_t st = nullptr;
    // C_q of every top row at this basis size: the accepted form's, or (rows on the filter /
    // transform kernels) the row's projection on the subspace -- then the factorisation is the
    // inverse of F M_r F^T + E, a PRECONDITIONER for K~ (rl_solve_pcg), not K~^-1.  (First of all:
    // on a handle without polynomial rows this is what creates the basis and sets its size.)
    std::vector<double> hCx;
    const double* hC = g->lr_hC.data();
    const char* cap_why = nullptr;
    if (!exact) {
        if (s->kn.no_precond_approx) { *why = "not every top row is in the polynomial form"; return RL_OK; }
        std::vector<char> ex;
        std::vector<double> cap;
        RL_TRY(lr_all_coeffs(g, R, &hCx, &ex, &cap));
        for (int q = 0; q < Q; ++q)
            if (!(cap[q] >= 0.8))
                cap_why = "not every top row is in the polynomial form, and the subspace holds less than 0.8 of a row's spectrum (no preconditioner either)";
        // (... which the 96-function basis below may still hold)
        if (cap_why != nullptr && !hz_wanted(s)) { *why = cap_why; return RL_OK; }
        hC = hCx.data();
    } else if ((int)g->lr_hC.size() < Q * R * R) {
        *why = "no host copy of the coefficient maps";
        return RL_OK;
    }
    RL_TRY(rp_prepare(s, std::max(R, 1)));
    if (s->rp_F == nullptr || s->rp_R != R) { *why = "no table of F"; return RL_OK; }
    // per-output noise, rows per output, 1 / eps per row on the device: per NOISE update (a
    // parameter update alone does not walk the n rows again)
    if (s->dz_eps_ver != s->noise_ver || (int)s->dz_eps.size() != D) {
        s->dz_eps.assign(D, 0.0);
        s->dz_eps_why = nullptr;
        for (int d = 0, a = 0; d < D && !s->dz_eps_why; ++d) {
            const int b = s->h_out_end[d];
            if (b > a) {
                s->dz_eps[d] = s->h_noise[a];
                for (int i = a; i < b; ++i)
                    if (s->h_noise[i] != s->dz_eps[d]) { s->dz_eps_why = "noise is not constant per output"; break; }
            } else {
                s->dz_eps[d] = 1.0;
            }
            if (!s->dz_eps_why && (!(s->dz_eps[d] > 0.0) || !std::isfinite(s->dz_eps[d])))
                s->dz_eps_why = "noise is not positive";
            a = b;
        }
        if (!s->dz_eps_why) {
            std::vector<double> inv((size_t)n);
            for (int i = 0; i < n; ++i) inv[i] = 1.0 / s->h_noise[i];
            if (!s->dz_inv) RL_HIP(hipMalloc((void**)&s->dz_inv, (size_t)n * sizeof(double)));
            RL_HIP(hipMemcpy(s->dz_inv, inv.data(), inv.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        s->dz_eps_ver = s->noise_ver;
    }
    if (s->dz_eps_why) { *why = s->dz_eps_why; return RL_OK; }
    const std::vector<double>& eps = s->dz_eps;
    std::vector<int> rows(D, 0);
    for (int d = 0, a = 0; d < D; ++d) {
        rows[d] = s->h_out_end[d] - a;
        a = s->h_out_end[d];
    }
    // an operator without a polynomial row, of enough rows: the 96-function preconditioner (hz_*
    // above); whatever it declines falls back to the 48 functions
    if (!exact && hz_wanted(s)) {
        bool hz_ok = false;
        const char* hz_why = "";
        RL_TRY(hz_try(s, eps, rows, &hz_ok, &hz_why));
        if (hz_ok) {
            s->dz_param_ver = g->param_ver;
            s->dz_noise_ver = s->noise_ver;
            s->dz_R = s->hz_Ruse;
            s->dz_exact = false;
            s->dz_hz = true;
            s->dz_valid = true;
            *ok = true;
            return RL_OK;
        }
        if (cap_why != nullptr) { *why = cap_why; return RL_OK; }
    }
    // Gram matrices of F on the unnormalised basis, once per (handle, rank): the columns of F
    // ARE a batch of R vectors (degree-major table)
    if (s->dz_U_R != R) {
#define RL_DZ_PROJ(R_) rp_project_plain<R_>(s, s->rp_F, R_, st)
        RL_DZ_RANKS(RL_DZ_PROJ);
#undef RL_DZ_PROJ
        RL_HIP(hipGetLastError());
        std::vector<double> part((size_t)s->rp_nruns * R * R);
        RL_HIP(hipMemcpy(part.data(), s->rp_part, part.size() * sizeof(double), hipMemcpyDeviceToHost));
        s->dz_U.assign((size_t)D * R * R, 0.0);
        for (int d = 0; d < D; ++d)
            for (int c = s->h_run_ptr[d]; c < s->h_run_ptr[d + 1]; ++c)
                for (int e = 0; e < R * R; ++e)
                    s->dz_U[(size_t)d * R * R + e] += part[(size_t)c * R * R + e];
        s->dz_U_R = R;
    }
    if ((int)g->lr_hnu.size() < R) { *why = "no host copy of the basis normalisation"; return RL_OK; }
    std::vector<double> Zs;
    double logdet = 0.0, pmin = 1.0, pmax = 1.0;
    std::vector<double> Zh;
    if (!dz_host_map(D, R, Q, s->dz_U.data(), g->lr_hnu.data(), eps, rows, g->lr_hB.data(), hC, Zs, &logdet,
                     &pmin, &pmax, why, s->dz_want_sample ? &Zh : nullptr))
        return RL_OK;
    RL_TRY(dz_upload_sample_map(s, Zh));
    for (double v : Zs)
        if (!std::isfinite(v)) { *why = "the solve map is not finite"; return RL_OK; }
    if (s->dz_Zt_cap < Zs.size()) {
        if (s->dz_Zt) RL_HIP(hipFree(s->dz_Zt));
        s->dz_Zt = nullptr;
        s->dz_Zt_cap = 0;
        RL_HIP(hipMalloc((void**)&s->dz_Zt, Zs.size() * sizeof(double)));
        s->dz_Zt_cap = Zs.size();
    }
    RL_HIP(hipMemcpy(s->dz_Zt, Zs.data(), Zs.size() * sizeof(double), hipMemcpyHostToDevice));
    s->dz_logdet = logdet;
    s->dz_cond = (pmax / pmin) * (pmax / pmin);
    s->dz_param_ver = g->param_ver;
    s->dz_noise_ver = s->noise_ver;
    s->dz_R = R;
    s->dz_exact = exact;
    s->dz_valid = true;
    *ok = true;
    return RL_OK;
}

// out = K~^-1 in (to roundoff), both in internal row order; in and out may not alias
static int dz_apply(rl_ski* s, const double* in, double* out, int nvec, hipStream_t st,
                    const double* map = nullptr, const double* diag = nullptr) {
    rl_gridop* g = s->g;
    if (s->dz_hz) return hz_apply(s, in, out, nvec, st, map, diag);
    if (map == nullptr) map = s->dz_Zt;
    if (diag == nullptr) diag = s->dz_inv;
    const int R = g->lr_r, D = g->D;
#define RL_DZ_PROJ(R_) rp_project_plain<R_>(s, in, nvec, st)
    RL_DZ_RANKS(RL_DZ_PROJ);
#undef RL_DZ_PROJ
    RL_LAUNCH(k_dz_mix, dim3((nvec + RL_DZ_VB - 1) / RL_DZ_VB), dim3(256),
              (size_t)RL_DZ_VB * D * R * sizeof(double), st, (const double*)s->rp_part,
              (const int*)s->rp_run_ptr, nvec, D, R, map, g->lr_zhat);
#define RL_DZ_EXP(R_) rp_expand_plain<R_>(s, g->lr_zhat, out, nvec, diag, in, st)
    RL_DZ_RANKS(RL_DZ_EXP);
#undef RL_DZ_EXP
    RL_HIP(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_ski_factor(rl_ski* s, int* available, double* logdet, double* cond) {
    if (!s) return fail(RL_EINVAL, "rl_ski_factor: NULL handle");
    bool ok = false;
    const char* why = "";
    RL_TRY(dz_ensure(s, &ok, &why));
    // (2: the factorisation inverts the operator's projection on the polynomial subspace -- a
    // preconditioner for rl_solve_pcg; its log det is not the operator's)
    // (3: the same on the 96-function basis of an operator without a polynomial row, hz_* above)
    if (available) *available = ok ? (s->dz_exact ? 1 : s->dz_hz ? 3 : 2) : 0;
    if (logdet) *logdet = ok && s->dz_exact ? s->dz_logdet : 0.0;
    if (cond) *cond = ok ? s->dz_cond : 0.0;
    if (!ok) (void)fail(RL_OK, std::string("direct solve not available: ") + why);
    return RL_OK;
}

extern "C" int rl_ski_project(rl_ski* s, const double* X, int nvec, double* out, int* rank,
                              void* stream) {
    if (!s || !X || !out) return fail(RL_EINVAL, "rl_ski_project: NULL argument");
    if (nvec < 0) return fail(RL_EINVAL, "rl_ski_project: nvec < 0");
    rl_gridop* g = s->g;
    bool ok = false;
    const char* why = "";
    RL_TRY(dz_ensure(s, &ok, &why));
    if (!ok) return fail(RL_ELIMIT, std::string("rl_ski_project: not available for this operator: ") + why);
    if (!s->dz_exact) return fail(RL_ELIMIT, "rl_ski_project: not every top row is in the polynomial form");
    if (rank) *rank = g->lr_r;
    if (nvec == 0) return RL_OK;
    RL_HIP(hipSetDevice(g->device));
    hipStream_t st = (hipStream_t)stream;
    RL_TRY(rp_prepare(s, std::max(nvec, g->lr_r)));
    const double* Xi = X;
    if (s->permuted) {
        RL_TRY(ski_reserve_perm(s, nvec));
        permute_rows(s, X, s->P1, nvec, 0, st);
        Xi = s->P1;
    }
#define RL_DZ_PROJ(R_) rp_project_plain<R_>(s, Xi, nvec, st)
    RL_DZ_RANKS(RL_DZ_PROJ);
#undef RL_DZ_PROJ
    RL_LAUNCH(k_dz_coeffs, dim3(nvec), dim3(256), 0, st, (const double*)s->rp_part,
              (const int*)s->rp_run_ptr, nvec, g->D, g->lr_r, (const double*)g->lr_nu, out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_solve_direct(rl_ski* s, const double* B, double* X, int nrhs, double tol,
                               int max_refine, int* iters_out, double* resid_out, int* istop_out,
                               void* stream) {
    if (!s || !B || !X) return fail(RL_EINVAL, "rl_solve_direct: NULL argument");
    if (nrhs < 0) return fail(RL_EINVAL, "rl_solve_direct: nrhs < 0");
    if (!(tol > 0.0)) return fail(RL_EINVAL, "rl_solve_direct: tol must be > 0");
    if (max_refine < 0) return fail(RL_EINVAL, "rl_solve_direct: max_refine < 0");
    if (nrhs == 0) return RL_OK;
    rl_gridop* g = s->g;
    RL_HIP(hipSetDevice(g->device));
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
    const char* why = "";
    RL_TRY(dz_ensure(s, &ok, &why));
    if (!ok) return fail(RL_ELIMIT, std::string("rl_solve_direct: not available for this operator: ") + why);
    if (!s->dz_exact)
        return fail(RL_ELIMIT, "rl_solve_direct: not every top row is in the polynomial form -- the "
                               "factorisation is a preconditioner here (rl_solve_pcg)");
    const int n = s->n, R = g->lr_r;
    // everything the kernels below allocate lazily
    RL_TRY(rp_prepare(s, std::max(nrhs, R)));
    RL_TRY(ski_reserve(s, nrhs));
    RL_TRY(gridop_prepare(g, nrhs));
    if (rp_ok(s, nrhs)) RL_TRY(rp_prepare(s, nrhs));
    RL_TRY(ski_reserve_perm(s, nrhs));
    const size_t ve = (size_t)nrhs * n;
    if (s->dz_vec_cap < ve) {
        if (s->dz_res) RL_HIP(hipFree(s->dz_res));
        if (s->dz_cor) RL_HIP(hipFree(s->dz_cor));
        s->dz_res = s->dz_cor = nullptr;
        s->dz_vec_cap = 0;
        RL_HIP(hipMalloc((void**)&s->dz_res, ve * sizeof(double)));
        RL_HIP(hipMalloc((void**)&s->dz_cor, ve * sizeof(double)));
        s->dz_vec_cap = ve;
    }
    if (s->dz_rhs_cap < (size_t)nrhs) {
        if (s->dz_part) RL_HIP(hipFree(s->dz_part));
        if (s->dz_go) RL_HIP(hipFree(s->dz_go));
        s->dz_part = nullptr;
        s->dz_go = nullptr;
        s->dz_rhs_cap = 0;
        RL_HIP(hipMalloc((void**)&s->dz_part, (size_t)nrhs * (2 * RL_DZ_NBLK + 1) * sizeof(double)));
        RL_HIP(hipMalloc((void**)&s->dz_go, (size_t)nrhs * sizeof(int)));
        s->dz_rhs_cap = (size_t)nrhs;
    }
    const double* Bi = B;
    double* Xi = X;
    if (s->permuted) {
        permute_rows(s, B, s->P1, nrhs, 0, st);
        Bi = s->P1;
        Xi = s->P2;
    }
    const int nblk = std::max(1, std::min(RL_DZ_NBLK, (n + 1023) / 1024));
    double* norms = s->dz_part + (size_t)nrhs * RL_DZ_NBLK;
    std::vector<double> res((size_t)nrhs, 0.0), best((size_t)nrhs, 1e300);
    std::vector<int> go((size_t)nrhs, 1), its((size_t)nrhs, 1), stop((size_t)nrhs, 0);
    trace_once("solve: direct, through the polynomial form (k_rp_project / k_dz_mix / k_rp_expand)");
    RL_TRY(dz_apply(s, Bi, Xi, nrhs, st));
    for (int it = 0;; ++it) {
        // r = b - K~ x and its norm; the reference's rule ends a system (iterative.py:36-42,54-58)
        RL_TRY(ski_mvm_int(s, Xi, s->dz_res, nrhs, st));
        RL_LAUNCH(k_dz_resid, dim3(nblk, nrhs), dim3(256), 256 * sizeof(double), st, Bi, s->dz_res, n,
                  s->dz_part);
        RL_LAUNCH(k_dz_norms, dim3((nrhs + 63) / 64), dim3(64), 0, st, (const double*)s->dz_part, nblk,
                  nrhs, norms);
        RL_HIP(hipGetLastError());
        RL_HIP(hipMemcpyAsync(res.data(), norms, (size_t)nrhs * sizeof(double), hipMemcpyDeviceToHost, st));
        RL_HIP(hipStreamSynchronize(st));
        int active = 0;
        for (int v = 0; v < nrhs; ++v) {
            if (!go[v]) continue;
            if (res[v] < tol) {
                go[v] = 0;
                stop[v] = RL_ISTOP_RESIDUAL;
            } else if (!std::isfinite(res[v]) || it >= max_refine) {
                go[v] = 0;
                stop[v] = RL_ISTOP_DIRECT_STALL;
            } else {
                ++active;
            }
        }
        if (active == 0) break;
        RL_HIP(hipMemcpyAsync(s->dz_go, go.data(), (size_t)nrhs * sizeof(int), hipMemcpyHostToDevice, st));
        RL_TRY(dz_apply(s, s->dz_res, s->dz_cor, nrhs, st));
        RL_LAUNCH(k_dz_axpy, dim3(nblk, nrhs), dim3(256), 0, st, Xi, (const double*)s->dz_cor, n,
                  (const int*)s->dz_go);
        for (int v = 0; v < nrhs; ++v) its[v] += go[v];
    }
    if (s->permuted) permute_rows(s, Xi, X, nrhs, 1, st);
    RL_HIP(hipGetLastError());
    RL_HIP(hipStreamSynchronize(st));
    for (int v = 0; v < nrhs; ++v) {
        if (iters_out) iters_out[v] = its[v];
        if (resid_out) resid_out[v] = res[v];
        if (istop_out) istop_out[v] = stop[v];
    }
    return RL_OK;
}

// ---------------------------------------------------------------------------
// Stochastic Lanczos quadrature on the host: r^T log(K~) r from a system's Lanczos tridiagonal
// ---------------------------------------------------------------------------
// Eigenvalues of the symmetric tridiagonal (diagonal d[0..n), off-diagonals e[0..n-1)) into d
// and the FIRST components of its normalised eigenvectors into z -- all a Gauss quadrature
// needs (Golub & Welsch): the implicit QL iteration with Wilkinson shifts, its plane rotations
// applied to one row of the eigenvector matrix instead of all n (O(n^2) in all; LAPACK's
// drivers return the whole matrix, O(n^3) or an MRRR pass that gives up on the strongly
// graded tridiagonals of a converged Lanczos run).  false: an eigenvalue did not settle.
static bool slq_ql_first_row(std::vector<double>& d, std::vector<double>& e, std::vector<double>& z) {
    const int n = (int)d.size();
    e.resize(n);
    e[n - 1] = 0.0;
    z.assign(n, 0.0);
    z[0] = 1.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m = l;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= eps * dd) break;
            }
            if (m == l) break;
            if (iter++ == 300) return false;
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = std::hypot(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + std::copysign(r, g));
            double s = 1.0, c = 1.0, p = 0.0;
            int i = m - 1;
            for (; i >= l; --i) {
                double f = s * e[i];
                const double b = c * e[i];
                r = std::hypot(f, g);
                e[i + 1] = r;
                if (r == 0.0) {
                    d[i + 1] -= p;
                    e[m] = 0.0;
                    break;
                }
                s = f / r;
                c = g / r;
                g = d[i + 1] - p;
                r = (d[i] - g) * s + 2.0 * c * b;
                p = s * r;
                d[i + 1] = g + p;
                g = c * r - b;
                f = z[i + 1];
                z[i + 1] = s * z[i] + c * f;
                z[i] = c * z[i] - s * f;
            }
            if (r == 0.0 && i >= l) continue;
            d[l] -= p;
            e[l] = g;
            e[m] = 0.0;
        } while (m != l);
    }
    return true;
}

// The loop of rl_solve_pcg.  lanczos / sqnorms (host [nrhs][cap][2], [nrhs]; both or neither): the run
// also RECORDS the recurrence's scalars and leaves the Lanczos matrix of the preconditioned operator
// P^-1/2 K~ P^-1/2 per system -- diagonal and off-diagonal pairs as rl_solve_batch_lanczos leaves
// MINRES's -- with r0^T P^-1 r0; such a run does not restart from explicit residuals (a restart
// starts another Krylov space) and ends at cap iterations at the latest.
static int pcg_core(rl_ski* s, const double* B, double* X, int nrhs, double tol, int maxiter,
                    int* iters_out, double* resid_out, int* istop_out, void* stream,
                    double* lanczos, int cap, double* sqnorms) {
    rl_gridop* g = s->g;
    const bool record = lanczos != nullptr;
    RL_HIP(hipSetDevice(g->device));
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
    const char* why = "";
    RL_TRY(dz_ensure(s, &ok, &why));
    if (!ok) return fail(RL_ELIMIT, std::string("rl_solve_pcg: no preconditioner for this operator: ") + why);
    const int n = s->n, R = g->lr_r;
    if (maxiter <= 0) maxiter = n;
    if (record) {
        maxiter = std::min(maxiter, cap);
        const size_t need = (size_t)cap * nrhs * 2;
        if (s->dz_rec_cap < need) {
            s->dz_rec_cap = 0;
            RL_TRY(hz_grow(&s->dz_rec, need));
            s->dz_rec_cap = need;
        }
    }
    RL_TRY(rp_prepare(s, std::max(nrhs, R)));
    RL_TRY(ski_reserve(s, nrhs));
    RL_TRY(gridop_prepare(g, nrhs));
    if (rp_ok(s, nrhs)) RL_TRY(rp_prepare(s, nrhs));
    RL_TRY(ski_reserve_perm(s, nrhs));
    if (s->dz_hz) RL_TRY(hz_reserve(s, nrhs));
    const size_t ve = (size_t)nrhs * n;
    if (s->dz_vec_cap < ve) {
        if (s->dz_res) RL_HIP(hipFree(s->dz_res));
        if (s->dz_cor) RL_HIP(hipFree(s->dz_cor));
        s->dz_res = s->dz_cor = nullptr;
        s->dz_vec_cap = 0;
        RL_HIP(hipMalloc((void**)&s->dz_res, ve * sizeof(double)));
        RL_HIP(hipMalloc((void**)&s->dz_cor, ve * sizeof(double)));
        s->dz_vec_cap = ve;
    }
    if (s->dz_pq_cap < ve) {
        if (s->dz_p) RL_HIP(hipFree(s->dz_p));
        if (s->dz_q) RL_HIP(hipFree(s->dz_q));
        s->dz_p = s->dz_q = nullptr;
        s->dz_pq_cap = 0;
        RL_HIP(hipMalloc((void**)&s->dz_p, ve * sizeof(double)));
        RL_HIP(hipMalloc((void**)&s->dz_q, ve * sizeof(double)));
        s->dz_pq_cap = ve;
    }
    if (s->dz_rhs_cap < (size_t)nrhs || s->dz_scal == nullptr) {
        if (s->dz_part) RL_HIP(hipFree(s->dz_part));
        if (s->dz_go) RL_HIP(hipFree(s->dz_go));
        if (s->dz_scal) RL_HIP(hipFree(s->dz_scal));
        s->dz_part = nullptr;
        s->dz_go = nullptr;
        s->dz_scal = nullptr;
        s->dz_rhs_cap = 0;
        RL_HIP(hipMalloc((void**)&s->dz_part, (size_t)nrhs * (2 * RL_DZ_NBLK + 1) * sizeof(double)));
        RL_HIP(hipMalloc((void**)&s->dz_go, (size_t)nrhs * sizeof(int)));
        RL_HIP(hipMalloc((void**)&s->dz_scal, (size_t)nrhs * 2 * sizeof(double)));
        s->dz_rhs_cap = (size_t)nrhs;
    }
    const double* Bi = B;
    double* Xi = X;
    if (s->permuted) {
        permute_rows(s, B, s->P1, nrhs, 0, st);
        Bi = s->P1;
        Xi = s->P2;
    }
    const int nblk = std::max(1, std::min(RL_DZ_NBLK, (n + 1023) / 1024));
    const dim3 vgrid(nblk, nrhs), vblk(256), hgrid((nrhs + 63) / 64), hblk(64);
    const size_t red = 256 * sizeof(double);
    double* part2 = s->dz_part + (size_t)nrhs * RL_DZ_NBLK;           // (r.r while p.q is still read)
    double* norms = s->dz_part + (size_t)2 * nrhs * RL_DZ_NBLK;
    double *r = s->dz_res, *z = s->dz_cor, *p = s->dz_p, *q = s->dz_q;
    std::vector<double> res((size_t)nrhs, 0.0);
    std::vector<int> go((size_t)nrhs, 1), its((size_t)nrhs, 0), stop((size_t)nrhs, 0);
    if (s->dz_hz) trace_once("solve: conjugate gradients preconditioned by the Woodbury inverse on 96 polynomials per output (rl_solve_pcg)");
    else trace_once("solve: conjugate gradients preconditioned by the polynomial subspace's Woodbury inverse (rl_solve_pcg)");
    RL_HIP(hipMemsetAsync(Xi, 0, ve * sizeof(double), st));
    RL_HIP(hipMemcpyAsync(r, Bi, ve * sizeof(double), hipMemcpyDeviceToDevice, st));
    RL_HIP(hipMemsetAsync(s->dz_scal, 0, (size_t)nrhs * 2 * sizeof(double), st));
    // zero right-hand sides end at once
    RL_LAUNCH(k_dot_partial, vgrid, vblk, red, st, (const double*)r, (const double*)r, n, s->dz_part);
    RL_LAUNCH(k_dz_norms, hgrid, hblk, 0, st, (const double*)s->dz_part, nblk, nrhs, norms);
    RL_HIP(hipMemcpyAsync(res.data(), norms, (size_t)nrhs * sizeof(double), hipMemcpyDeviceToHost, st));
    RL_HIP(hipStreamSynchronize(st));
    int active = 0;
    for (int v = 0; v < nrhs; ++v) {
        if (res[v] == 0.0) { go[v] = 0; stop[v] = RL_ISTOP_ZERO_RHS; }
        else if (res[v] < tol) { go[v] = 0; stop[v] = RL_ISTOP_RESIDUAL; }
        else ++active;
    }
    int restarts = 0;
    for (int k = 0; active > 0; ++k) {
        RL_HIP(hipMemcpyAsync(s->dz_go, go.data(), (size_t)nrhs * sizeof(int), hipMemcpyHostToDevice, st));
        RL_TRY(dz_apply(s, r, z, nrhs, st));                              // z = M r
        RL_LAUNCH(k_dot_partial, vgrid, vblk, red, st, (const double*)r, (const double*)z, n, s->dz_part);
        RL_LAUNCH(k_pcg_head, hgrid, hblk, 0, st, (const double*)s->dz_part, nblk, nrhs, s->dz_scal,
                  (const int*)s->dz_go);
        RL_LAUNCH(k_pcg_p, vgrid, vblk, 0, st, p, (const double*)z, n, (const double*)s->dz_scal,
                  (const int*)s->dz_go, k == 0 ? 1 : 0);
        RL_TRY(ski_mvm_int(s, p, q, nrhs, st));                            // q = K~ p
        RL_LAUNCH(k_dot_partial, vgrid, vblk, red, st, (const double*)p, (const double*)q, n, s->dz_part);
        RL_LAUNCH(k_pcg_update, vgrid, vblk, red, st, Xi, r, (const double*)p, (const double*)q, n,
                  (const double*)s->dz_scal, (const double*)s->dz_part, part2, (const int*)s->dz_go,
                  record ? s->dz_rec + (size_t)k * nrhs * 2 : (double*)nullptr);
        RL_LAUNCH(k_dz_norms, hgrid, hblk, 0, st, (const double*)part2, nblk, nrhs, norms);
        RL_HIP(hipGetLastError());
        RL_HIP(hipMemcpyAsync(res.data(), norms, (size_t)nrhs * sizeof(double), hipMemcpyDeviceToHost, st));
        RL_HIP(hipStreamSynchronize(st));
        active = 0;
        for (int v = 0; v < nrhs; ++v) {
            if (!go[v]) continue;
            ++its[v];
            if (!std::isfinite(res[v])) { go[v] = 0; stop[v] = RL_ISTOP_DIRECT_STALL; }
            else if (res[v] < tol) { go[v] = 0; stop[v] = RL_ISTOP_RESIDUAL; }
            else if (its[v] >= maxiter) { go[v] = 0; stop[v] = 6; }
            else ++active;
        }
        if (active == 0 && restarts < 3 && !record) {
            // the recurrence's residuals met the rule: the EXPLICIT residuals decide (the
            // reference's own final check, iterative.py:54); a system whose explicit residual is
            // still above the tolerance goes on from it
            RL_TRY(ski_mvm_int(s, Xi, q, nrhs, st));
            RL_LAUNCH(k_dz_resid, vgrid, vblk, red, st, Bi, q, n, s->dz_part);
            RL_LAUNCH(k_dz_norms, hgrid, hblk, 0, st, (const double*)s->dz_part, nblk, nrhs, norms);
            RL_HIP(hipMemcpyAsync(res.data(), norms, (size_t)nrhs * sizeof(double), hipMemcpyDeviceToHost, st));
            RL_HIP(hipStreamSynchronize(st));
            for (int v = 0; v < nrhs; ++v)
                if (stop[v] == RL_ISTOP_RESIDUAL && res[v] >= tol && its[v] < maxiter) {
                    go[v] = 1;
                    stop[v] = 0;
                    ++active;
                }
            if (active > 0) {
                // restart from the explicit residuals (q holds b - K~ x of every system; frozen
                // systems' vectors are not touched again)
                RL_HIP(hipMemcpyAsync(r, q, ve * sizeof(double), hipMemcpyDeviceToDevice, st));
                ++restarts;
                k = -1;                       // (the next pass is a first iteration: p = z)
            }
        }
    }
    // final explicit residuals of every system (what resid_out reports)
    RL_TRY(ski_mvm_int(s, Xi, q, nrhs, st));
    RL_LAUNCH(k_dz_resid, vgrid, vblk, red, st, Bi, q, n, s->dz_part);
    RL_LAUNCH(k_dz_norms, hgrid, hblk, 0, st, (const double*)s->dz_part, nblk, nrhs, norms);
    RL_HIP(hipMemcpyAsync(res.data(), norms, (size_t)nrhs * sizeof(double), hipMemcpyDeviceToHost, st));
    if (s->permuted) permute_rows(s, Xi, X, nrhs, 1, st);
    RL_HIP(hipGetLastError());
    RL_HIP(hipStreamSynchronize(st));
    for (int v = 0; v < nrhs; ++v) {
        if (iters_out) iters_out[v] = its[v];
        if (resid_out) resid_out[v] = res[v];
        if (istop_out) istop_out[v] = stop[v];
    }
    if (record) {
        // conjugate gradients' scalars -> the Lanczos matrix (Saad, Iterative Methods, 6.7.3):
        //   alpha_j = rho_j / (p_j . q_j),  beta_j = rho_j / rho_{j-1};
        //   T[j][j] = 1 / alpha_j + beta_j / alpha_{j-1},  T[j][j+1] = sqrt(beta_{j+1}) / alpha_j
        int kmax = 0;
        for (int v = 0; v < nrhs; ++v) kmax = std::max(kmax, its[v]);
        std::vector<double> rec((size_t)std::max(kmax, 1) * nrhs * 2, 0.0);
        if (kmax > 0)
            RL_HIP(hipMemcpy(rec.data(), s->dz_rec, (size_t)kmax * nrhs * 2 * sizeof(double), hipMemcpyDeviceToHost));
        for (int v = 0; v < nrhs; ++v) {
            double* lz = lanczos + (size_t)v * cap * 2;
            std::fill(lz, lz + (size_t)cap * 2, 0.0);
            sqnorms[v] = its[v] > 0 ? rec[2 * (size_t)v] : 0.0;
            double alpha_prev = 0.0, rho_prev = 0.0;
            for (int j = 0; j < its[v]; ++j) {
                const double rho = rec[((size_t)j * nrhs + v) * 2], pq = rec[((size_t)j * nrhs + v) * 2 + 1];
                const double alpha = rho / pq;
                const double beta = j > 0 ? rho / rho_prev : 0.0;
                lz[2 * j] = 1.0 / alpha + (j > 0 ? beta / alpha_prev : 0.0);
                if (j > 0) lz[2 * (j - 1) + 1] = std::sqrt(beta) / alpha_prev;
                alpha_prev = alpha;
                rho_prev = rho;
            }
        }
    }
    return RL_OK;
}

extern "C" int rl_solve_pcg(rl_ski* s, const double* B, double* X, int nrhs, double tol, int maxiter,
                            int* iters_out, double* resid_out, int* istop_out, void* stream) {
    if (!s || !B || !X) return fail(RL_EINVAL, "rl_solve_pcg: NULL argument");
    if (nrhs < 0) return fail(RL_EINVAL, "rl_solve_pcg: nrhs < 0");
    if (!(tol > 0.0)) return fail(RL_EINVAL, "rl_solve_pcg: tol must be > 0");
    if (nrhs == 0) return RL_OK;
    return pcg_core(s, B, X, nrhs, tol, maxiter, iters_out, resid_out, istop_out, stream, nullptr, 0, nullptr);
}

extern "C" int rl_solve_pcg_lanczos(rl_ski* s, const double* B, double* X, int nrhs, double tol, int maxiter,
                                    int* iters_out, double* resid_out, int* istop_out, double* lanczos_out,
                                    int cap, double* sqnorms_out, void* stream) {
    if (!s || !B || !X || !lanczos_out || !sqnorms_out) return fail(RL_EINVAL, "rl_solve_pcg_lanczos: NULL argument");
    if (nrhs < 0 || cap < 1) return fail(RL_EINVAL, "rl_solve_pcg_lanczos: bad sizes");
    if (!(tol > 0.0)) return fail(RL_EINVAL, "rl_solve_pcg_lanczos: tol must be > 0");
    if (nrhs == 0) return RL_OK;
    return pcg_core(s, B, X, nrhs, tol, maxiter, iters_out, resid_out, istop_out, stream, lanczos_out, cap,
                    sqnorms_out);
}

// Rout[v] = E^1/2 B Win[v]: rows of Win with identity covariance (the reference's +-1 probes) become
// rows with the covariance P of the current (inexact) factorisation -- the right-hand sides whose
// preconditioned Lanczos quadrature estimates log det (P^-1 K~) without bias.  *logdet_p = log det P.
extern "C" int rl_ski_precond_sample(rl_ski* s, const double* Win, double* Rout, int nvec, double* logdet_p,
                                     void* stream) {
    if (!s || !Win || !Rout) return fail(RL_EINVAL, "rl_ski_precond_sample: NULL argument");
    if (nvec < 0) return fail(RL_EINVAL, "rl_ski_precond_sample: nvec < 0");
    rl_gridop* g = s->g;
    RL_HIP(hipSetDevice(g->device));
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
    const char* why = "";
    if (!s->dz_want_sample) {
        s->dz_want_sample = true;          // (from now on every factorisation of this handle keeps the map)
        s->dz_valid = false;
    }
    RL_TRY(dz_ensure(s, &ok, &why));
    if (!ok) return fail(RL_ELIMIT, std::string("rl_ski_precond_sample: no factorisation for this operator: ") + why);
    if (!s->dz_Zh_valid) return fail(RL_ELIMIT, "rl_ski_precond_sample: no square-root map");
    if (logdet_p) *logdet_p = s->dz_logdet;
    if (nvec == 0) return RL_OK;
    const int n = s->n;
    if (s->dz_isq_ver != s->noise_ver || s->dz_isq == nullptr) {
        std::vector<double> isq((size_t)n);
        for (int i = 0; i < n; ++i) isq[i] = 1.0 / std::sqrt(s->h_noise[i]);
        if (!s->dz_isq) RL_HIP(hipMalloc((void**)&s->dz_isq, (size_t)n * sizeof(double)));
        RL_HIP(hipMemcpy(s->dz_isq, isq.data(), isq.size() * sizeof(double), hipMemcpyHostToDevice));
        s->dz_isq_ver = s->noise_ver;
    }
    RL_TRY(rp_prepare(s, std::max(nvec, g->lr_r)));
    RL_TRY(ski_reserve_perm(s, nvec));
    if (s->dz_hz) RL_TRY(hz_reserve(s, nvec));
    RL_TRY(lr_reserve(g, nvec));
    const double* Wi = Win;
    double* Ri = Rout;
    if (s->permuted) {
        permute_rows(s, Win, s->P1, nvec, 0, st);
        Wi = s->P1;
        Ri = s->P2;
    }
    // in = w / sqrt(eps), in a buffer of its own (the apply functions do not alias in and out)
    const size_t ve = (size_t)nvec * n;
    if (s->dz_smp_cap < ve) {
        s->dz_smp_cap = 0;
        RL_TRY(hz_grow(&s->dz_smp, ve));
        s->dz_smp_cap = ve;
    }
    const int nblk = std::max(1, std::min(RL_DZ_NBLK, (n + 1023) / 1024));
    RL_LAUNCH(k_dz_scale, dim3(nblk, nvec), dim3(256), 0, st, Wi, (const double*)s->dz_isq, n, s->dz_smp);
    RL_TRY(dz_apply(s, s->dz_smp, Ri, nvec, st, (const double*)s->dz_Zh, (const double*)s->noise_diag));
    if (s->permuted) permute_rows(s, Ri, Rout, nvec, 1, st);
    RL_HIP(hipGetLastError());
    return RL_OK;
}


// ---------------------------------------------------------------------------
// Host helper: +-1 probes drawn as the reference draws them (int64) -> one byte per entry
// ---------------------------------------------------------------------------
extern "C" int rl_probes_to_int8(const long long* src, int nrows, long long row_stride, long long n,
                                 signed char* dst, int nthreads, int* all_pm1) {
    if (!src || !dst || !all_pm1) return fail(RL_EINVAL, "rl_probes_to_int8: NULL argument");
    if (nrows < 0 || n < 0) return fail(RL_EINVAL, "rl_probes_to_int8: negative size");
    const int nt = (int)std::max<long long>(1, std::min<long long>(nthreads, (long long)nrows * n / 65536 + 1));
    std::vector<int> bad((size_t)nt, 0);
    // (ONE pass: every entry is read once, checked and narrowed; rows of the matrix are
    // `row_stride` entries apart -- a rank's share of a round-robin deal is a strided view)
    auto work = [&](int t) {
        const long long total = (long long)nrows * n, lo = total * t / nt, hi = total * (t + 1) / nt;
        int b = 0;
        long long r = n > 0 ? lo / n : 0, c = n > 0 ? lo - r * n : 0;
        for (long long e = lo; e < hi;) {
            const long long* sr = src + r * row_stride;
            signed char* dr = dst + r * n;
            const long long cend = std::min(n, c + (hi - e));
            for (long long k = c; k < cend; ++k) {
                const long long v = sr[k];
                b |= (v != 1 && v != -1);
                dr[k] = (signed char)v;
            }
            e += cend - c;
            c = 0;
            ++r;
        }
        bad[t] = b;
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (std::thread& th : pool) th.join();
    }
    int any = 0;
    for (int b : bad) any |= b;
    *all_pm1 = any ? 0 : 1;
    return RL_OK;
}

extern "C" int rl_slq_log_quadrature(const double* lanczos, int nrhs, int cap, const int* iters,
                                     const double* sqnorms, double* out, int nthreads) {
    if (!lanczos || !iters || !sqnorms || !out) return fail(RL_EINVAL, "rl_slq_log_quadrature: NULL argument");
    if (nrhs < 0 || cap < 1) return fail(RL_EINVAL, "rl_slq_log_quadrature: bad sizes");
    std::vector<int> bad((size_t)std::max(nrhs, 1), 0);
    auto work = [&](int first, int step) {
        std::vector<double> d, e, z;
        for (int v = first; v < nrhs; v += step) {
            const int k = std::min(iters[v], cap);
            out[v] = 0.0;
            if (k < 1) continue;
            const double* lz = lanczos + (size_t)v * cap * 2;
            d.resize(k);
            e.assign(k, 0.0);
            for (int j = 0; j < k; ++j) d[j] = lz[2 * j];
            for (int j = 0; j + 1 < k; ++j) e[j] = lz[2 * j + 1];
            if (!slq_ql_first_row(d, e, z)) {
                // (the caller's fallback takes this system: NaN marks it)
                bad[v] = 1;
                out[v] = std::nan("");
                continue;
            }
            double acc = 0.0;
            for (int j = 0; j < k; ++j)
                if (d[j] > 0.0) acc += z[j] * z[j] * std::log(d[j]);
            out[v] = sqnorms[v] * acc;
        }
    };
    const int nt = std::max(1, std::min(nthreads, nrhs));
    if (nt == 1) {
        work(0, 1);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work, t, nt);
        for (std::thread& th : pool) th.join();
    }
    return RL_OK;
}

#if defined(RL_TIMING) && !defined(RL_EMU)
// experiment builds: the phase stamps of the last launches (see rl_device.h)
extern "C" int rl_debug_poke(int slot, long long value) {
    RL_HIP(hipDeviceSynchronize());
    RL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rl_timing_buf), &value, sizeof(long long),
                             (size_t)slot * sizeof(long long)));
    return RL_OK;
}
extern "C" int rl_debug_timing(long long* out, int count) {
    RL_HIP(hipDeviceSynchronize());
    RL_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(rl_timing_buf),
                               (size_t)std::min(count, 256) * sizeof(long long)));
    return RL_OK;
}
#endif

// ---------------------------------------------------------------------------
// gradient partial sums
// ---------------------------------------------------------------------------
extern "C" int rl_cross_dots(const double* U, const double* V, int nvec, int D, int m,
                             double* out, void* stream) {
    if (!U || !V || !out) return fail(RL_EINVAL, "rl_cross_dots: NULL argument");
    if (nvec < 0 || D < 1 || m < 1) return fail(RL_EINVAL, "rl_cross_dots: bad sizes");
    if (nvec == 0) return RL_OK;
    if (D >= 4 && m >= 1024) {
        const unsigned nb = ((D + RL_XD_A - 1) / RL_XD_A) * ((D + RL_XD_B - 1) / RL_XD_B);
        RL_LAUNCH(k_cross_dots_tiled, dim3(nb, nvec), dim3(RL_SOLVER_THREADS),
                  RL_SOLVER_THREADS * sizeof(double), (hipStream_t)stream, U, V, D, m, out);
    } else {
        RL_LAUNCH(k_cross_dots, dim3(D * D, nvec), dim3(RL_SOLVER_THREADS),
                  RL_SOLVER_THREADS * sizeof(double), (hipStream_t)stream, U, V, D, m, out);
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_segment_dots(const double* U, const double* V, const int* offsets, int nvec,
                               int n, int D, double* out, void* stream) {
    if (!U || !V || !offsets || !out) return fail(RL_EINVAL, "rl_segment_dots: NULL argument");
    if (nvec < 0 || D < 1 || n < 1) return fail(RL_EINVAL, "rl_segment_dots: bad sizes");
    if (nvec == 0) return RL_OK;
    RL_LAUNCH(k_segment_dots, dim3(D, nvec), dim3(RL_SOLVER_THREADS),
              RL_SOLVER_THREADS * sizeof(double), (hipStream_t)stream, U, V, offsets, n, D,
              out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

