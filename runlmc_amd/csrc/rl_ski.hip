// SKI operator K~ = sum_t W_t K_t W_t^T + diag(eps): handles, interpolation products, the
// row-polynomial form of a polynomial-form operator, products in the caller's and in the
// internal row order (one of the three translation units of librunlmc_hip.so, rl_host.h).
#include "rl_host.h"


// Small batches want every (row, vector) on its own thread (latency-bound);
// large ones want the CSR entries reused across a block of vectors
// (bandwidth-bound: the structure is 12 bytes per non-zero per pass).
static void launch_spmv(const int* indptr, const int* indices, const double* vals, int nrows,
                        int ncols, int nvec, const double* X, double* Y, const double* diag,
                        const double* X2, hipStream_t st, int accumulate = 0,
                        int* bump = nullptr, int avg_nnz = 0) {
    const unsigned gx = (nrows + RL_THREADS - 1) / RL_THREADS;
    const bool blocked = (size_t)nrows * nvec >= ((size_t)1 << 22);
    // long rows of a small product: eight lanes per row (k_spmv_wide)
    if (!blocked && avg_nnz > 12 && diag == nullptr && !accumulate) {
        const unsigned gw = (unsigned)(((size_t)nrows * 8 + RL_THREADS - 1) / RL_THREADS);
        RL_LAUNCH(k_spmv_wide<8>, dim3(gw, nvec), dim3(RL_THREADS), RL_THREADS * sizeof(double), st,
                  indptr, indices, vals, nrows, ncols, X, Y, bump);
        return;
    }
    if (blocked) {
        RL_LAUNCH(k_spmv<8>, dim3(gx, (nvec + 7) / 8), dim3(RL_THREADS), 0, st, indptr, indices,
                  vals, nrows, ncols, nvec, X, Y, diag, X2, accumulate, bump);
    } else {
        RL_LAUNCH(k_spmv<1>, dim3(gx, nvec), dim3(RL_THREADS), 0, st, indptr, indices, vals,
                  nrows, ncols, nvec, X, Y, diag, X2, accumulate, bump);
    }
}

static int check_csr(const int* indptr, const int* indices, int nrows, int ncols,
                     const char* what) {
    if (!indptr || indptr[0] != 0) return fail(RL_EINVAL, std::string(what) + ": bad indptr[0]");
    for (int i = 0; i < nrows; ++i)
        if (indptr[i + 1] < indptr[i])
            return fail(RL_EINVAL, std::string(what) + ": indptr not monotone");
    const int nnz = indptr[nrows];
    if (nnz > 0 && !indices) return fail(RL_EINVAL, std::string(what) + ": indices is NULL");
    for (int k = 0; k < nnz; ++k)
        if (indices[k] < 0 || indices[k] >= ncols)
            return fail(RL_EINVAL, std::string(what) + ": column index out of range");
    return RL_OK;
}

int upload_raw(void** dev, const void* host, size_t bytes) {
    RL_HIP(hipMalloc(dev, std::max<size_t>(bytes, 16)));
    if (bytes) RL_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return RL_OK;
}

// Upload one term's CSR pair, rows of W (and column indices of WT) renumbered
// by `perm` (perm[i] = caller's row of internal row i) when given.
static int upload_term(SkiTerm* t, int n, int ngrid, const int* W_indptr, const int* W_indices,
                       const double* W_data, const int* WT_indptr, const int* WT_indices,
                       const double* WT_data, const std::vector<int>* perm) {
    const size_t nnz = W_indptr[n];
    std::vector<int> wp_ptr, wp_idx, wtp_idx;
    std::vector<double> wp_val, wtp_val;
    if (perm) {
        std::vector<int> inv(n);
        for (int i = 0; i < n; ++i) inv[(*perm)[i]] = i;
        wp_ptr.assign(n + 1, 0);
        wp_idx.reserve(nnz);
        wp_val.reserve(nnz);
        for (int i = 0; i < n; ++i) {
            const int r = (*perm)[i];
            for (int k = W_indptr[r]; k < W_indptr[r + 1]; ++k) {
                wp_idx.push_back(W_indices[k]);
                wp_val.push_back(W_data[k]);
            }
            wp_ptr[i + 1] = (int)wp_idx.size();
        }
        wtp_idx.resize(nnz);
        wtp_val.resize(nnz);
        std::vector<std::pair<int, double>> row;
        for (int r = 0; r < ngrid; ++r) {
            row.clear();
            for (int k = WT_indptr[r]; k < WT_indptr[r + 1]; ++k)
                row.emplace_back(inv[WT_indices[k]], WT_data[k]);
            std::sort(row.begin(), row.end());
            for (size_t j = 0; j < row.size(); ++j) {
                wtp_idx[WT_indptr[r] + j] = row[j].first;
                wtp_val[WT_indptr[r] + j] = row[j].second;
            }
        }
        W_indptr = wp_ptr.data(); W_indices = wp_idx.data(); W_data = wp_val.data();
        WT_indices = wtp_idx.data(); WT_data = wtp_val.data();
    }
    t->ngrid = ngrid;
    // interpolation structure (see SkiTerm)
    std::vector<int> base, lo, sp_ptr, sp_idx;
    std::vector<double> w4, sp_val;
    {
        bool ok = ngrid >= 4;
        base.resize(n);
        w4.assign((size_t)4 * n, 0.0);
        int prev = 0;
        for (int i = 0; i < n && ok; ++i) {
            const int k0 = W_indptr[i], cnt = W_indptr[i + 1] - k0;
            if (cnt > 4) { ok = false; break; }
            for (int j = 1; j < cnt; ++j)
                if (W_indices[k0 + j] != W_indices[k0] + j) ok = false;
            int b = cnt ? W_indices[k0] : prev, shift = 0;
            if (b > ngrid - 4) { shift = b - (ngrid - 4); b = ngrid - 4; }
            if (shift + cnt > 4 || b < prev) { ok = false; break; }
            prev = b;
            base[i] = b;
            for (int j = 0; j < cnt; ++j) w4[(size_t)4 * i + shift + j] = W_data[k0 + j];
        }
        if (ok) {
            // W^T: grid row r is touched by the rows with base in [r - 3, r]
            lo.resize(ngrid);
            sp_ptr.assign(ngrid + 1, 0);
            int a = 0, e = 0;
            for (int r = 0; r < ngrid; ++r) {
                while (a < n && base[a] < r - 3) ++a;
                while (e < n && base[e] <= r) ++e;
                lo[r] = a;
                sp_ptr[r + 1] = sp_ptr[r] + (e - a);
                for (int i = a; i < e; ++i) {
                    sp_idx.push_back(i);
                    sp_val.push_back(w4[(size_t)4 * i + (r - base[i])]);
                }
            }
            t->h_base = base;
            RL_TRY(upload_raw((void**)&t->W4_base, base.data(), (size_t)n * sizeof(int)));
            RL_TRY(upload_raw((void**)&t->W4_w, w4.data(), (size_t)4 * n * sizeof(double)));
            RL_TRY(upload_raw((void**)&t->WT_lo, lo.data(), (size_t)ngrid * sizeof(int)));
            WT_indptr = sp_ptr.data(); WT_indices = sp_idx.data(); WT_data = sp_val.data();
            for (int r0 = 0; r0 < n; r0 += RL_THREADS)
                t->w_xmax = std::max(t->w_xmax,
                                     base[std::min(n, r0 + RL_THREADS) - 1] + 4 - base[r0]);
            for (int r0 = 0; r0 < ngrid; r0 += RL_THREADS) {
                const int rl = std::min(ngrid, r0 + RL_THREADS) - 1;
                const int c1 = lo[rl] + (sp_ptr[rl + 1] - sp_ptr[rl]);
                t->wt_xmax = std::max(t->wt_xmax, c1 - lo[r0]);
                t->wt_emax = std::max(t->wt_emax, sp_ptr[rl + 1] - sp_ptr[r0]);
            }
        }
    }
    const size_t nnzT = WT_indptr[ngrid];
    t->nnzWT = (int)nnzT;
    RL_TRY(upload_raw((void**)&t->W_indptr, W_indptr, (size_t)(n + 1) * sizeof(int)));
    RL_TRY(upload_raw((void**)&t->W_indices, W_indices, nnz * sizeof(int)));
    RL_TRY(upload_raw((void**)&t->W_data, W_data, nnz * sizeof(double)));
    RL_TRY(upload_raw((void**)&t->WT_indptr, WT_indptr, (size_t)(ngrid + 1) * sizeof(int)));
    RL_TRY(upload_raw((void**)&t->WT_indices, WT_indices, nnzT * sizeof(int)));
    RL_TRY(upload_raw((void**)&t->WT_data, WT_data, nnzT * sizeof(double)));
    return RL_OK;
}

extern "C" int rl_ski_create(rl_gridop* g, int n, const int* W_indptr, const int* W_indices,
                             const double* W_data, const int* WT_indptr, const int* WT_indices,
                             const double* WT_data, rl_ski** out) {
    if (!out) return fail(RL_EINVAL, "out is NULL");
    *out = nullptr;
    if (!g) return fail(RL_EINVAL, "gridop is NULL");
    if (n < 1) return fail(RL_EINVAL, "rl_ski_create: n < 1");
    const int ngrid = g->D * g->m;
    RL_TRY(check_csr(W_indptr, W_indices, n, ngrid, "W"));
    RL_TRY(check_csr(WT_indptr, WT_indices, ngrid, n, "WT"));
    if (W_indptr[n] != WT_indptr[ngrid])
        return fail(RL_EINVAL, "rl_ski_create: W and WT have different nnz");
    RL_HIP(hipSetDevice(g->device));
    rl_ski* s = new rl_ski;
    HandleGuard<rl_ski, rl_ski_destroy> guard(s);
    s->kn = read_knobs();
    s->g = g;
    s->device = g->device;
    s->n = n;
    s->ngrid = ngrid;
    // sort the data points by the first grid point they touch
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    auto key = [&](int i) {
        return W_indptr[i + 1] > W_indptr[i] ? W_indices[W_indptr[i]] : 0x7fffffff;
    };
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return key(a) < key(b); });
    bool identity = true;
    for (int i = 0; i < n && identity; ++i) identity = perm[i] == i;
    if (s->kn.no_sort) identity = true;
    if (!identity) {
        s->permuted = true;
        s->h_perm = perm;
        RL_TRY(upload_raw((void**)&s->perm, perm.data(), (size_t)n * sizeof(int)));
    }
    SkiTerm t0;
    RL_TRY(upload_term(&t0, n, ngrid, W_indptr, W_indices, W_data, WT_indptr, WT_indices,
                       WT_data, s->permuted ? &s->h_perm : nullptr));
    s->W_indptr = t0.W_indptr; s->W_indices = t0.W_indices; s->W_data = t0.W_data;
    s->WT_indptr = t0.WT_indptr; s->WT_indices = t0.WT_indices; s->WT_data = t0.WT_data;
    s->W4_base = t0.W4_base; s->W4_w = t0.W4_w; s->WT_lo = t0.WT_lo; s->nnzWT = t0.nnzWT;
    s->h_base.swap(t0.h_base);
    s->wt_xmax = t0.wt_xmax; s->wt_emax = t0.wt_emax; s->w_xmax = t0.w_xmax;
    s->max_ngrid = ngrid;
    s->nnz = W_indptr[n];
    RL_HIP(hipMalloc((void**)&s->noise_diag, (size_t)n * sizeof(double)));
    RL_HIP(hipMemset(s->noise_diag, 0, (size_t)n * sizeof(double)));
    *out = guard.release();
    return RL_OK;
}

extern "C" int rl_ski_add_term(rl_ski* s, rl_gridop* g, const int* W_indptr,
                               const int* W_indices, const double* W_data,
                               const int* WT_indptr, const int* WT_indices,
                               const double* WT_data) {
    if (!s || !g) return fail(RL_EINVAL, "rl_ski_add_term: NULL handle");
    if (g->D != s->g->D || g->device != s->g->device)
        return fail(RL_EINVAL, "rl_ski_add_term: grid operator has another D or device");
    const int ngrid = g->D * g->m;
    RL_TRY(check_csr(W_indptr, W_indices, s->n, ngrid, "W"));
    RL_TRY(check_csr(WT_indptr, WT_indices, ngrid, s->n, "WT"));
    if (W_indptr[s->n] != WT_indptr[ngrid])
        return fail(RL_EINVAL, "rl_ski_add_term: W and WT have different nnz");
    RL_HIP(hipSetDevice(g->device));
    SkiTerm t;
    t.g = g;
    // the handle's row order was fixed by its first term
    int rc = upload_term(&t, s->n, ngrid, W_indptr, W_indices, W_data, WT_indptr, WT_indices,
                         WT_data, s->permuted ? &s->h_perm : nullptr);
    s->extra.push_back(t);     // pushed even on failure so destroy frees what was uploaded
    if (rc != RL_OK) return rc;
    if (ngrid > s->max_ngrid) {
        s->max_ngrid = ngrid;
        s->cap = 0;            // grid-side temporaries must grow
    }
    return RL_OK;
}


extern "C" int rl_ski_destroy(rl_ski* s) {
    if (!s) return RL_OK;
    (void)hipSetDevice(s->device);
    for (SkiTerm& t : s->extra) {
        void* tp[] = {t.W_indptr, t.W_indices, t.W_data, t.WT_indptr, t.WT_indices, t.WT_data,
                      t.W4_base, t.W4_w, t.WT_lo};
        for (void* p : tp)
            if (p) (void)hipFree(p);
    }
    if (s->solver_stream) (void)hipStreamDestroy(s->solver_stream);
    if (s->pin_count) (void)hipHostFree(s->pin_count);
    for (hipEvent_t e : s->count_ev)
        if (e) (void)hipEventDestroy(e);
    if (s->ws_valid) free_work(s->ws);
    void* ptrs[] = {s->W_indptr, s->W_indices, s->W_data, s->WT_indptr, s->WT_indices,
                    s->WT_data, s->noise_diag, s->G1, s->G2, s->perm, s->P1, s->P2,
                    s->W4_base, s->W4_w, s->WT_lo, s->lanczos_buf, s->poly_tab, s->poly_ob,
                    s->poly_part, s->rp_F, s->rp_Fc, s->rp_runs, s->rp_run_ptr, s->rp_out_end, s->rp_part, s->rp_nrm, s->rp_pp,
                    s->rp_base_c, s->rp_w4_c, s->dz_Zt, s->dz_inv, s->dz_res, s->dz_cor, s->dz_part,
                    s->dz_go, s->dz_p, s->dz_q, s->dz_scal, s->hz_beta, s->hz_phi, s->hz_tphi, s->hz_C,
                    s->hz_F, s->hz_ones, s->hz_part, s->hz_zhat, s->hz_tmp, s->hz_S, s->hz_P, s->dz_Zh, s->dz_isq, s->dz_smp, s->dz_rec};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete s;
    return RL_OK;
}

extern "C" int rl_ski_set_noise(rl_ski* s, const double* noise, const int* lens) {
    if (!s || !noise || !lens) return fail(RL_EINVAL, "rl_ski_set_noise: NULL argument");
    std::vector<double> diag;
    diag.reserve(s->n);
    for (int d = 0; d < s->g->D; ++d) {
        if (lens[d] < 0) return fail(RL_EINVAL, "rl_ski_set_noise: negative length");
        diag.insert(diag.end(), (size_t)lens[d], noise[d]);
    }
    if ((int)diag.size() != s->n)
        return fail(RL_EINVAL, "rl_ski_set_noise: sum(lens) != n");
    s->caller_noise_same = true;
    if (s->permuted) {
        std::vector<double> sorted(diag.size());
        for (int i = 0; i < s->n; ++i) sorted[i] = diag[s->h_perm[i]];
        for (int i = 0; i < s->n && s->caller_noise_same; ++i)
            s->caller_noise_same = sorted[i] == diag[i];
        diag.swap(sorted);
    }
    // runs of equal values in internal order (one per output when every output's
    // rows are contiguous): what the solver's vector kernel takes instead of the array
    s->eps_end.clear();
    s->eps_val.clear();
    for (int i = 0; i < s->n; ++i) {
        if (i == 0 || diag[i] != diag[i - 1]) {
            if ((int)s->eps_val.size() == RL_MAX_D) {       // not per-output after all
                s->eps_end.clear();
                s->eps_val.clear();
                break;
            }
            if (i > 0) s->eps_end.push_back(i);
            s->eps_val.push_back(diag[i]);
        }
    }
    if (!s->eps_val.empty()) s->eps_end.push_back(s->n);
    RL_HIP(hipSetDevice(s->g->device));
    RL_HIP(hipMemcpy(s->noise_diag, diag.data(), diag.size() * sizeof(double),
                     hipMemcpyHostToDevice));
    s->has_noise = true;
    s->h_noise.swap(diag);
    ++s->noise_ver;
    return RL_OK;
}

int ski_reserve(rl_ski* s, int nvec) {
    if (nvec <= s->cap && s->G1) return RL_OK;
    if (s->G1) RL_HIP(hipFree(s->G1));
    if (s->G2) RL_HIP(hipFree(s->G2));
    s->G1 = s->G2 = nullptr;
    s->cap = 0;
    RL_HIP(hipMalloc((void**)&s->G1, (size_t)nvec * s->max_ngrid * sizeof(double)));
    RL_HIP(hipMalloc((void**)&s->G2, (size_t)nvec * s->max_ngrid * sizeof(double)));
    s->cap = nvec;
    return RL_OK;
}

int ski_reserve_perm(rl_ski* s, int nvec) {
    if (!s->permuted || nvec <= s->pcap) return RL_OK;
    if (s->P1) RL_HIP(hipFree(s->P1));
    if (s->P2) RL_HIP(hipFree(s->P2));
    s->P1 = s->P2 = nullptr;
    s->pcap = 0;
    RL_HIP(hipMalloc((void**)&s->P1, (size_t)nvec * s->n * sizeof(double)));
    RL_HIP(hipMalloc((void**)&s->P2, (size_t)nvec * s->n * sizeof(double)));
    s->pcap = nvec;
    return RL_OK;
}

// caller order <-> internal (sorted) order
void permute_rows(rl_ski* s, const double* X, double* Y, int nvec, int scatter,
                         hipStream_t st) {
    // (row-block count padded to a multiple of 8 for the XCD-aware order; the
    // kernel masks rows past n)
    dim3 grid((((s->n + RL_THREADS - 1) / RL_THREADS) + 7) / 8 * 8, nvec);
    RL_LAUNCH(k_permute_rows, grid, dim3(RL_THREADS), 0, st, X, Y, (const int*)s->perm, s->n,
              scatter);
}

// groups of 8 vectors a staged-SpMV workgroup walks with the same rows
// (measured at C5: DESIGN.md)
static int staged_vgroups() { return 4; }

// the three stages in INTERNAL row order
int ski_wt_int(rl_ski* s, const double* Xp, double* G, int nvec, hipStream_t st, int* bump) {
    // large batch, structured W^T whose workgroup ranges fit LDS: staged form
    // (measured at C5, 129 vectors: see DESIGN.md)
    constexpr int VB = 8;
    const size_t lds = ((size_t)VB * s->wt_xmax + s->wt_emax) * sizeof(double);
    if (s->WT_lo != nullptr && s->wt_xmax > 0 && lds <= 64 * 1024 &&
        s->wt_xmax <= 4 * RL_THREADS &&
        ((size_t)s->ngrid * nvec >= ((size_t)1 << 22) || s->kn.staged_wt) && !s->kn.no_staged_wt) {
        trace_once("W^T product: k_spmv_wt_staged");
        static unsigned long long seen = 0;
        if (first_on_device(&seen)) {
            (void)hipFuncSetAttribute((const void*)k_spmv_wt_staged<VB, 1>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_spmv_wt_staged<VB, 2>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_spmv_wt_staged<VB, 4>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        }
        const unsigned gx = (s->ngrid + RL_THREADS - 1) / RL_THREADS;
        const int vg = staged_vgroups();
        const dim3 grid(gx, ((nvec + VB - 1) / VB + vg - 1) / vg);
#define RL_WT_STAGED(XPT)                                                                       \
    RL_LAUNCH((k_spmv_wt_staged<VB, XPT>), grid, dim3(RL_THREADS), lds, st,                     \
              (const int*)s->WT_indptr, (const int*)s->WT_lo, (const double*)s->WT_data,        \
              s->ngrid, s->n, nvec, Xp, G, s->wt_xmax, vg, bump)
        if (s->wt_xmax <= RL_THREADS) RL_WT_STAGED(1);
        else if (s->wt_xmax <= 2 * RL_THREADS) RL_WT_STAGED(2);
        else RL_WT_STAGED(4);
#undef RL_WT_STAGED
        RL_HIP(hipGetLastError());
        return RL_OK;
    }
    launch_spmv(s->WT_indptr, s->WT_indices, s->WT_data, s->ngrid, s->n, nvec, Xp, G, nullptr,
                nullptr, st, 0, bump, s->ngrid > 0 ? s->nnzWT / s->ngrid : 0);
    RL_HIP(hipGetLastError());
    return RL_OK;
}
// does the W product of a batch take the staged form?
static bool w_staged_ok(const rl_ski* s, int nvec) {
    constexpr int VB = 8;
    const size_t lds = (size_t)VB * s->w_xmax * sizeof(double);
    return s->W4_base != nullptr && s->w_xmax > 0 && lds <= 64 * 1024 &&
           s->w_xmax <= 4 * RL_THREADS &&
           ((size_t)s->n * nvec >= ((size_t)1 << 22) || s->kn.staged_wt) && !s->kn.no_staged_wt;
}
// ... and with MINRES's P inside (k_spmv_w_staged_p: the partial sums' words next to the tile,
// row accesses by 32-bit byte offsets)?
static size_t w_staged_p_lds(const rl_ski* s) {
    constexpr int VB = 8;
    size_t lds = ((size_t)VB * s->w_xmax + (size_t)staged_vgroups() * VB * 8) * sizeof(double);
#if defined(RL_EMU)
    lds += 256 * sizeof(double);
#endif
    return lds;
}
bool w_staged_p_ok(const rl_ski* s, int nvec) {
    return w_staged_ok(s, nvec) && w_staged_p_lds(s) <= 64 * 1024 && s->n < (1 << 28);
}
static int ski_w_int(rl_ski* s, const double* G, double* Yp, int nvec, const double* diag,
                     const double* X2p, hipStream_t st) {
    // large batch, structured W: staged form (see ski_wt_int)
    constexpr int VB = 8;
    const size_t lds = (size_t)VB * s->w_xmax * sizeof(double);
    const RpPFuse pf = s->rp_pfuse;          // (by value: the solver clears the handle's copy)
    if (pf.pc != nullptr) {
        // MINRES's P inside the product (the solver checked w_staged_ok): no output vector
        if (!w_staged_p_ok(s, nvec))
            return fail(RL_EINVAL, "internal: MINRES update fused into a W product that does not run staged");
        trace_once("W product: k_spmv_w_staged_p (MINRES's P inside)");
        static unsigned long long seenp = 0;
        if (first_on_device(&seenp)) {
            (void)hipFuncSetAttribute((const void*)k_spmv_w_staged_p<VB, 1>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_spmv_w_staged_p<VB, 2>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_spmv_w_staged_p<VB, 4>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        }
        const unsigned gx = (s->n + RL_THREADS - 1) / RL_THREADS;
        const int vg = staged_vgroups();
        const dim3 grid(gx, ((nvec + VB - 1) / VB + vg - 1) / vg);
        const size_t ldsp = w_staged_p_lds(s);
#define RL_W_STAGED_P(XPT)                                                                      \
    RL_LAUNCH((k_spmv_w_staged_p<VB, XPT>), grid, dim3(RL_THREADS), ldsp, st,                   \
              (const int*)s->W4_base, (const double*)s->W4_w, s->n, s->ngrid, nvec, G, diag,    \
              X2p, s->w_xmax, vg, pf)
        if (s->w_xmax <= RL_THREADS) RL_W_STAGED_P(1);
        else if (s->w_xmax <= 2 * RL_THREADS) RL_W_STAGED_P(2);
        else RL_W_STAGED_P(4);
#undef RL_W_STAGED_P
        RL_HIP(hipGetLastError());
        return RL_OK;
    }
    if (w_staged_ok(s, nvec)) {
        trace_once("W product: k_spmv_w_staged");
        static unsigned long long seen = 0;
        if (first_on_device(&seen)) {
            (void)hipFuncSetAttribute((const void*)k_spmv_w_staged<VB, 1>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_spmv_w_staged<VB, 2>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            (void)hipFuncSetAttribute((const void*)k_spmv_w_staged<VB, 4>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        }
        const unsigned gx = (s->n + RL_THREADS - 1) / RL_THREADS;
        const int vg = staged_vgroups();
        const dim3 grid(gx, ((nvec + VB - 1) / VB + vg - 1) / vg);
#define RL_W_STAGED(XPT)                                                                        \
    RL_LAUNCH((k_spmv_w_staged<VB, XPT>), grid, dim3(RL_THREADS), lds, st,                      \
              (const int*)s->W4_base, (const double*)s->W4_w, s->n, s->ngrid, nvec, G, Yp, diag, \
              X2p, s->w_xmax, vg)
        if (s->w_xmax <= RL_THREADS) RL_W_STAGED(1);
        else if (s->w_xmax <= 2 * RL_THREADS) RL_W_STAGED(2);
        else RL_W_STAGED(4);
#undef RL_W_STAGED
        RL_HIP(hipGetLastError());
        return RL_OK;
    }
    launch_spmv(s->W_indptr, s->W_indices, s->W_data, s->n, s->ngrid, nvec, G, Yp, diag, X2p,
                st);
    RL_HIP(hipGetLastError());
    return RL_OK;
}
// can the W product of a batch take the grid values from the polynomial form's mixed
// coefficients (k_spmv_w_poly)?  Staged W on a 1-D grid the polynomial form is eligible
// for; whether the operator IS in that form at rank 24 or 32 the grid handle decides when
// the product runs (its verification may still be pending here): lr_launch
static bool ski_w_poly_ok(const rl_ski* s, int nvec) {
    constexpr int VB = 8;
    const rl_gridop* g = s->g;
    const size_t lds = ((size_t)VB * s->w_xmax + (size_t)VB * 2 * RL_LR_RMAX) * sizeof(double);
    return s->W4_base != nullptr && s->w_xmax > 0 && s->w_xmax <= 2 * RL_THREADS &&
           lds <= 64 * 1024 && ((size_t)s->n * nvec >= ((size_t)1 << 22) || s->kn.staged_wt) &&
           !s->kn.no_staged_wt &&
           (g->lr_try || g->lr_ok) && g->m >= 4 * RL_THREADS && s->ngrid == g->D * g->m &&
           !s->kn.no_w_poly;
}
static int ski_w_poly(rl_ski* s, double* Yp, int nvec, const double* diag, const double* X2p,
                      hipStream_t st) {
    constexpr int VB = 8;
    rl_gridop* g = s->g;
    const int R = g->lr_r;
    const size_t lds = ((size_t)VB * s->w_xmax + (size_t)VB * 2 * R) * sizeof(double);
    trace_once("W product: k_spmv_w_poly (grid values from the mixed coefficients)");
    const unsigned gx = (s->n + RL_THREADS - 1) / RL_THREADS;
    const int vg = staged_vgroups();
    const dim3 grid(gx, ((nvec + VB - 1) / VB + vg - 1) / vg);
#define RL_W_POLY(XPT, R_)                                                                      \
    RL_LAUNCH((k_spmv_w_poly<VB, XPT, R_>), grid, dim3(RL_THREADS), lds, st,                    \
              (const int*)s->W4_base, (const double*)s->W4_w, s->n, s->ngrid, nvec,             \
              (const double*)g->lr_zhat, (const double*)g->lr_beta, g->D, g->m, Yp, diag, X2p,  \
              s->w_xmax, vg)
    if (R == 24) {
        if (s->w_xmax <= RL_THREADS) RL_W_POLY(1, 24);
        else RL_W_POLY(2, 24);
    } else if (R == 32) {
        if (s->w_xmax <= RL_THREADS) RL_W_POLY(1, 32);
        else RL_W_POLY(2, 32);
    } else if (R == 36) {
        if (s->w_xmax <= RL_THREADS) RL_W_POLY(1, 36);
        else RL_W_POLY(2, 36);
    } else {
        return fail(RL_EINVAL, "k_spmv_w_poly: no instantiation for this rank");
    }
#undef RL_W_POLY
    RL_HIP(hipGetLastError());
    return RL_OK;
}
// ---------------------------------------------------------------------------
// Row-polynomial form of large solver rounds (rl_rowpoly.h)
// ---------------------------------------------------------------------------
// may this handle's operator run as F M F^T for a batch of nvec vectors?  (single term,
// structured W on a 1-D grid, every top row in the polynomial form, a large system)
bool rp_ok(const rl_ski* s, int nvec) {
    const rl_gridop* g = s->g;
    return s->extra.empty() && s->W4_base != nullptr && !s->h_base.empty() && !g->wide &&
           g->lr_try && !g->lr_dirty && g->lr_ok && s->ngrid == g->D * g->m &&
           s->n < (1 << 28) &&          // (k_rp_expand's row accesses carry 32-bit byte offsets)
           // (the batch gate of the structured forms also gates this one: rl_gridop_set_form_gate
           // with a huge value puts the whole operator back on the transform kernels)
           (size_t)nvec * g->D * g->m >= g->lr_min &&
           ((size_t)s->n * nvec >= ((size_t)1 << 22) || s->kn.staged_wt) &&
           // (F is read twice per product whatever the batch: at ranks above 32 a batch of a
           // few dozen vectors is level with the interpolation products or behind them --
           // C5, 17 vectors, rank 36: 0.615 against 0.587 ms per round; rank 24: 0.49 against 0.53)
           // (... except batches of at most 17, which take the small-batch projection:
           // rank 36, 17 vectors: see profiles/r04/rp_ab.txt)
           (g->lr_r <= 32 || nvec >= 48 || nvec <= RL_RP_VG + 1 || s->kn.staged_wt) &&
           !s->kn.no_staged_wt && !s->kn.no_rp;
}
// F for the operator's current rank, the runs of rows k_rp_project walks, the partial sums
// of nvec vectors.  Allocates: never inside a stream capture (the solver calls it before).
int rp_prepare(rl_ski* s, int nvec) {
    rl_gridop* g = s->g;
    const int R = g->lr_r, n = s->n, D = g->D, m = g->m;
    if (s->rp_R != R && (s->kn.rp_fly & 2)) {
        s->rp_R = R;                 // (no table: both kernels compute F from the entries)
    }
    if (s->rp_R != R) {
        if (s->rp_F) RL_HIP(hipFree(s->rp_F));
        s->rp_F = nullptr;
        s->rp_R = 0;
        RL_HIP(hipMalloc((void**)&s->rp_F, (size_t)R * n * sizeof(double)));
        RL_LAUNCH(k_rp_build, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t) nullptr,
                  (const int*)s->W4_base, (const double*)s->W4_w, n, m, R,
                  (const double*)g->lr_beta, s->rp_F);
        RL_HIP(hipGetLastError());
        RL_HIP(hipDeviceSynchronize());
        s->rp_R = R;
    }
    if (!s->rp_runs) {
        // rows of an output are contiguous in the sorted order; runs of whole tiles, about five
        // hundred of them (C5, rows per run 512 ... 2048: the projection 344-371 us at 129
        // vectors, 74-76 at 17; k_lr_mix, which sums the runs, 32 -> 18 us: the longest wins)
        std::vector<int> out_end(D, 0), run_ptr(D + 1, 0), runs;
        int len = ((n + 511) / 512 + RL_RP_TILE - 1) / RL_RP_TILE * RL_RP_TILE;
        len = std::max(RL_RP_TILE, std::min(len, 64 * RL_RP_TILE));
        if (s->kn.rp_runlen > 0) len = (s->kn.rp_runlen + RL_RP_TILE - 1) / RL_RP_TILE * RL_RP_TILE;
        int i = 0;
        for (int d = 0; d < D; ++d) {
            const int start = i;
            while (i < n && s->h_base[i] < (d + 1) * m) ++i;
            out_end[d] = i;
            run_ptr[d] = (int)runs.size() / 3;
            for (int r0 = start; r0 < i; r0 += len) {
                runs.push_back(r0);
                runs.push_back(std::min(r0 + len, i));
                runs.push_back(d);
            }
        }
        run_ptr[D] = (int)runs.size() / 3;
        if (i != n) return fail(RL_EINVAL, "row-polynomial form: rows are not sorted by output");
        if (runs.empty()) return fail(RL_EINVAL, "row-polynomial form: no rows");
        RL_TRY(upload_raw((void**)&s->rp_runs, runs.data(), runs.size() * sizeof(int)));
        RL_TRY(upload_raw((void**)&s->rp_run_ptr, run_ptr.data(), run_ptr.size() * sizeof(int)));
        RL_TRY(upload_raw((void**)&s->rp_out_end, out_end.data(), out_end.size() * sizeof(int)));
        s->rp_nruns = run_ptr[D];
        s->h_run_ptr = run_ptr;
        s->h_out_end = out_end;
    }
    const size_t need = (size_t)s->rp_nruns * nvec * R;
    if (s->rp_part_cap < need) {
        if (s->rp_part) RL_HIP(hipFree(s->rp_part));
        s->rp_part = nullptr;
        s->rp_part_cap = 0;
        RL_HIP(hipMalloc((void**)&s->rp_part, need * sizeof(double)));
        s->rp_part_cap = need;
    }
    RL_TRY(lr_reserve(g, nvec));             // (the mixed coefficients live on the grid handle)
    return RL_OK;
}
bool rp_ready(const rl_ski* s, int nvec) {
    return (s->rp_F != nullptr || (s->kn.rp_fly & 2)) && s->rp_R == s->g->lr_r && s->rp_runs != nullptr &&
           s->rp_part_cap >= (size_t)s->rp_nruns * nvec * s->rp_R &&
           s->g->lr_zhat_cap >= (size_t)nvec * s->g->D * RL_LR_RMAX;
}
template <int R, bool FLYP, bool FLYE>
static void rp_launch(rl_ski* s, const double* F, const int* base, const double* w4,
                      const double* Xp, double* Yp, int nvec, const double* diag, hipStream_t st,
                      int* bump) {
    rl_gridop* g = s->g;
    constexpr int NT = (R + 15) / 16;
    const size_t lds = (((size_t)16 * NT + 2 * RL_RP_VG) * RL_RP_LD + RL_RP_TILE) * sizeof(double);
    const int vblk = RL_RP_NG(R) * RL_RP_VG;
    const RpFuse fz = s->rp_fuse;            // (by value: the solver clears the handle's copy)
    if constexpr (!FLYP) {
        // at most one block of 16 vectors and a lone last one (a rank's share of an 8-way probe
        // split): the small-batch kernel, next tile's loads in flight during the current one
        if (nvec <= RL_RP_VG + 1 && (nvec <= RL_RP_VG || nvec % RL_RP_VG == 1) && !s->kn.no_rp_small) {
            const size_t lds1 = (((size_t)16 * NT + RL_RP_VG) * RL_RP_LD + RL_RP_TILE) * sizeof(double);
            // (F from the table: computed on the fly here it measured 74 against 72 us)
            if (fz.r2 != nullptr)
                RL_LAUNCH((k_rp_project1<R, false, true>), dim3(s->rp_nruns), dim3(256), lds1, st, Xp,
                          s->n, nvec, F, (const int*)s->rp_runs, s->rp_part, bump, base, w4, g->m,
                          (const double*)g->lr_beta, fz);
            else
                RL_LAUNCH((k_rp_project1<R, false>), dim3(s->rp_nruns), dim3(256), lds1, st, Xp, s->n,
                          nvec, F, (const int*)s->rp_runs, s->rp_part, bump, base, w4, g->m,
                          (const double*)g->lr_beta, RpFuse{nullptr, nullptr, nullptr});
            goto projected;
        }
        if (fz.r2 != nullptr) {
            const int vb = RL_RP_NG_FB(R) * RL_RP_VG;
            RL_LAUNCH((k_rp_project<R, false, true>),
                      dim3(8 * ((s->rp_nruns + 7) / 8) * ((nvec + vb - 1) / vb)), dim3(256), lds, st,
                      Xp, s->n, nvec, F, (const int*)s->rp_runs, s->rp_nruns, s->rp_part, bump, base,
                      w4, g->m, (const double*)g->lr_beta, fz);
            goto projected;
        }
    }
    RL_LAUNCH((k_rp_project<R, FLYP>), dim3(8 * ((s->rp_nruns + 7) / 8) * ((nvec + vblk - 1) / vblk)),
              dim3(256), lds, st, Xp, s->n, nvec, F, (const int*)s->rp_runs, s->rp_nruns, s->rp_part,
              bump, base, w4, g->m, (const double*)g->lr_beta, RpFuse{nullptr, nullptr, nullptr});
projected:
    int split3 = 0;
    const size_t mix_lds = lr_mix_lds(g->D, R, g->Q, &split3);
    RL_LAUNCH(k_lr_mix, dim3(nvec), dim3(RL_LR_MIXT), mix_lds, st,
              (const double*)s->rp_part, 0, nvec, g->D, R, g->Q, (const double*)g->lr_C,
              (const double*)g->lr_B, (const double*)g->lr_nu, g->lr_zhat,
              (const int*)s->rp_run_ptr, split3);
    if (s->rp_mid) s->rp_mid(st);
    const RpPFuse pf = s->rp_pfuse;          // (by value: the solver clears the handle's copy)
    if (pf.pc != nullptr) {
        // MINRES's P inside the expansion: no operator output is written
        size_t lds = (size_t)nvec * 8 * sizeof(double);
#if defined(RL_EMU)
        lds += 256 * sizeof(double);
#endif
        RL_LAUNCH((k_rp_expand<R, FLYE, true>), dim3((s->n + 255) / 256), dim3(256), lds, st,
                  (const double*)g->lr_zhat, F, s->n, nvec, g->D,
                  (const int*)s->rp_out_end, Yp, diag, Xp, s->kn.rp_stagger, base, w4, g->m,
                  (const double*)g->lr_beta, pf);
        return;
    }
    const RpPFuse nopf{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (diag != nullptr)
        RL_LAUNCH((k_rp_expand<R, FLYE, false, true>), dim3((s->n + 255) / 256), dim3(256), 0, st,
                  (const double*)g->lr_zhat, F, s->n, nvec, g->D,
                  (const int*)s->rp_out_end, Yp, diag, Xp, s->kn.rp_stagger, base, w4, g->m,
                  (const double*)g->lr_beta, nopf);
    else
        RL_LAUNCH((k_rp_expand<R, FLYE, false, false>), dim3((s->n + 255) / 256), dim3(256), 0, st,
                  (const double*)g->lr_zhat, F, s->n, nvec, g->D,
                  (const int*)s->rp_out_end, Yp, diag, Xp, s->kn.rp_stagger, base, w4, g->m,
                  (const double*)g->lr_beta, nopf);
}
// Do the rows of every output occupy the SAME index range in the caller's order as in the
// sorted one?  (True for W built output by output, multi_interpolant's block-diagonal layout,
// runlmc/approx/interpolation.py:161-176; false when the caller interleaves outputs.)  Only
// then do the sorted order's runs, output borders and per-output noise serve a batch in the
// caller's order; otherwise rl_ski_mvm permutes the batch (correct for any row order).
static bool caller_order_same(rl_ski* s) {
    if (s->caller_ranges_same < 0) {
        const int n = s->n, m = s->g->m;
        bool same = (int)s->h_base.size() == n && (int)s->h_perm.size() == n;
        for (int a = 0; a < n && same;) {
            const int d = s->h_base[a] / m;
            int b = a;
            while (b < n && s->h_base[b] / m == d) ++b;
            for (int i = a; i < b && same; ++i) same = s->h_perm[i] >= a && s->h_perm[i] < b;
            a = b;
        }
        s->caller_ranges_same = same ? 1 : 0;
    }
    return s->caller_ranges_same == 1 && s->caller_noise_same;
}
// caller_order: the batch is in the caller's row order (F / entries permuted accordingly)
static int ski_rp_mvm(rl_ski* s, bool caller_order, const double* Xp, double* Yp, int nvec,
                      const double* diag, hipStream_t st, int* bump) {
    trace_once("K~ product: row-polynomial form (k_rp_project / k_lr_mix / k_rp_expand)");
    const bool flyp = (s->kn.rp_fly & 2) != 0, flye = (s->kn.rp_fly & 1) != 0;
    const double* F = caller_order ? s->rp_Fc : s->rp_F;
    const int* base = caller_order ? s->rp_base_c : s->W4_base;
    const double* w4 = caller_order ? s->rp_w4_c : s->W4_w;
#define RL_RP_CASE(R_)                                                                         \
    case R_:                                                                                    \
        if (flyp) rp_launch<R_, true, true>(s, F, base, w4, Xp, Yp, nvec, diag, st, bump);     \
        else if (flye) rp_launch<R_, false, true>(s, F, base, w4, Xp, Yp, nvec, diag, st, bump); \
        else rp_launch<R_, false, false>(s, F, base, w4, Xp, Yp, nvec, diag, st, bump);         \
        break
    switch (s->g->lr_r) {
        RL_RP_CASE(24); RL_RP_CASE(32); RL_RP_CASE(36); RL_RP_CASE(40); RL_RP_CASE(48);
        default: return fail(RL_EINVAL, "row-polynomial form: bad basis size");
    }
#undef RL_RP_CASE
    RL_HIP(hipGetLastError());
    return RL_OK;
}
// Yp = K~ Xp, both in internal row order (what the solver iterates on)
// (noise = false: Yp = W K_UU W^T Xp only -- the caller adds eps (.) Xp itself)
int ski_mvm_int(rl_ski* s, const double* Xp, double* Yp, int nvec, hipStream_t st, int* bump,
                bool noise) {
    RL_TRY(ski_reserve(s, nvec));
    const double* diag = s->has_noise && noise ? s->noise_diag : nullptr;
    // (a pending form decision is taken here, so that the path does not depend on whether an
    // earlier product happened to trigger it)
    if (s->extra.empty() && !s->g->wide && !stream_capturing(st)) RL_TRY(lr_prepare(s->g, nvec));
    // every top row in the polynomial form, a large system: F M F^T, no interpolation
    // products, no grid vector (rl_rowpoly.h).  (Buffers: the solver prepares them before it
    // captures; a plain product outside a capture prepares them here.)
    if (rp_ok(s, nvec)) {
        if (!rp_ready(s, nvec) && !stream_capturing(st)) RL_TRY(rp_prepare(s, nvec));
        if (rp_ready(s, nvec)) {
            return ski_rp_mvm(s, false, Xp, Yp, nvec, diag, st, bump);
        }
    }
    // (the solver asked for its vector update inside the projection: no other path does it)
    if (s->rp_fuse.r2 != nullptr)
        return fail(RL_EINVAL, "internal: MINRES update fused into a projection that does not run");
    // (the solver asked for P inside the W product: the staged kernel, a single term, and the
    // grid vector written -- not the W kernel that expands the polynomial form itself)
    const bool wp = s->rp_pfuse.pc != nullptr;
    if (wp && (!s->extra.empty() || !w_staged_p_ok(s, nvec)))
        return fail(RL_EINVAL, "internal: MINRES update fused into a W product that does not run staged");
    RL_TRY(ski_wt_int(s, Xp, s->G1, nvec, st, bump));
    // a polynomial-form operator hands its mixed coefficients to the W kernel instead of
    // writing the grid vector (the grid handle says whether it took that path)
    s->g->expand_deferred = false;
    s->g->defer_expand = !wp && ski_w_poly_ok(s, nvec);
    const int rc = rl_gridop_mvm(s->g, s->G1, s->G2, nvec, st);
    s->g->defer_expand = false;
    if (rc != RL_OK) return rc;
    if (s->g->expand_deferred) {
        s->g->expand_deferred = false;
        RL_TRY(ski_w_poly(s, Yp, nvec, diag, Xp, st));
    } else {
        if (wp && s->rp_mid) s->rp_mid(st);       // (P's scalar head: k_minres2_ph)
        RL_TRY(ski_w_int(s, s->G2, Yp, nvec, diag, Xp, st));
    }
    for (const SkiTerm& t : s->extra) {       // Yp += W_t K_t W_t^T Xp
        launch_spmv(t.WT_indptr, t.WT_indices, t.WT_data, t.ngrid, s->n, nvec, Xp, s->G1,
                    nullptr, nullptr, st);
        RL_TRY(rl_gridop_mvm(t.g, s->G1, s->G2, nvec, st));
        launch_spmv(t.W_indptr, t.W_indices, t.W_data, s->n, t.ngrid, nvec, s->G2, Yp, nullptr,
                    nullptr, st, 1);
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// CSR pair of term `term` (0 = the handle's first term)
static int term_view(rl_ski* s, int term, SkiTerm* out) {
    if (term < 0 || term > (int)s->extra.size())
        return fail(RL_EINVAL, "SKI term index out of range");
    if (term == 0) {
        out->g = s->g;
        out->ngrid = s->ngrid;
        out->W_indptr = s->W_indptr; out->W_indices = s->W_indices; out->W_data = s->W_data;
        out->WT_indptr = s->WT_indptr; out->WT_indices = s->WT_indices;
        out->WT_data = s->WT_data;
    } else {
        *out = s->extra[term - 1];
    }
    return RL_OK;
}

extern "C" int rl_ski_apply_wt_term(rl_ski* s, int term, const double* X, double* G, int nvec,
                                    void* stream) {
    if (!s || !X || !G) return fail(RL_EINVAL, "rl_ski_apply_wt: NULL argument");
    if (nvec <= 0) return nvec == 0 ? RL_OK : fail(RL_EINVAL, "nvec < 0");
    SkiTerm t;
    RL_TRY(term_view(s, term, &t));
    RL_HIP(hipSetDevice(s->g->device));
    hipStream_t st = (hipStream_t)stream;
    if (s->permuted) {
        RL_TRY(ski_reserve_perm(s, nvec));
        permute_rows(s, X, s->P1, nvec, 0, st);
        X = s->P1;
    }
    launch_spmv(t.WT_indptr, t.WT_indices, t.WT_data, t.ngrid, s->n, nvec, X, G, nullptr,
                nullptr, st);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_ski_apply_w_term(rl_ski* s, int term, const double* G, double* Y, int nvec,
                                   void* stream) {
    if (!s || !G || !Y) return fail(RL_EINVAL, "rl_ski_apply_w: NULL argument");
    if (nvec <= 0) return nvec == 0 ? RL_OK : fail(RL_EINVAL, "nvec < 0");
    SkiTerm t;
    RL_TRY(term_view(s, term, &t));
    RL_HIP(hipSetDevice(s->g->device));
    hipStream_t st = (hipStream_t)stream;
    double* dst = Y;
    if (s->permuted) {
        RL_TRY(ski_reserve_perm(s, nvec));
        dst = s->P2;
    }
    launch_spmv(t.W_indptr, t.W_indices, t.W_data, s->n, t.ngrid, nvec, G, dst, nullptr, nullptr,
                st);
    if (s->permuted) permute_rows(s, s->P2, Y, nvec, 1, st);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_ski_apply_wt(rl_ski* s, const double* X, double* G, int nvec, void* stream) {
    return rl_ski_apply_wt_term(s, 0, X, G, nvec, stream);
}

extern "C" int rl_ski_apply_w(rl_ski* s, const double* G, double* Y, int nvec, void* stream) {
    return rl_ski_apply_w_term(s, 0, G, Y, nvec, stream);
}

extern "C" int rl_ski_mvm(rl_ski* s, const double* X, double* Y, int nvec, void* stream) {
    if (!s || !X || !Y) return fail(RL_EINVAL, "rl_ski_mvm: NULL argument");
    if (X == Y) return fail(RL_EINVAL, "rl_ski_mvm: X and Y may not alias");
    if (nvec <= 0) return nvec == 0 ? RL_OK : fail(RL_EINVAL, "nvec < 0");
    RL_HIP(hipSetDevice(s->g->device));
    hipStream_t st = (hipStream_t)stream;
    if (!s->permuted) return ski_mvm_int(s, X, Y, nvec, st);
    // Row-polynomial form in the CALLER's row order: F's columns permuted once per rank, no
    // row permutation of the batch (they were half of this product's time at C5).  The rows of
    // an output are contiguous in both orders, so runs, output borders and the noise array
    // (constant per output) serve both.
    if (s->extra.empty() && !s->g->wide && !stream_capturing(st) && caller_order_same(s)) {
        RL_TRY(lr_prepare(s->g, nvec));
        if (rp_ok(s, nvec)) {
            RL_TRY(rp_prepare(s, nvec));
            if (s->kn.rp_fly && (!s->rp_base_c || !s->rp_w4_c)) {
                // (guarded one by one: a failed second allocation must not leave the first
                // behind as a sign that both exist)
                if (!s->rp_base_c)
                    RL_HIP(hipMalloc((void**)&s->rp_base_c, (size_t)s->n * sizeof(int)));
                if (!s->rp_w4_c) {
                    const hipError_t e = hipMalloc((void**)&s->rp_w4_c, (size_t)4 * s->n * sizeof(double));
                    if (e != hipSuccess) {
                        s->rp_w4_c = nullptr;
                        (void)hipFree(s->rp_base_c);
                        s->rp_base_c = nullptr;
                        RL_HIP(e);
                    }
                }
                RL_LAUNCH(k_rp_permute_entries, dim3((s->n + 255) / 256), dim3(256), 0,
                          (hipStream_t) nullptr, (const int*)s->W4_base, (const double*)s->W4_w,
                          (const int*)s->perm, s->n, s->rp_base_c, s->rp_w4_c);
                RL_HIP(hipGetLastError());
                RL_HIP(hipDeviceSynchronize());
            }
            if (!(s->kn.rp_fly & 2) && s->rp_Fc_R != s->rp_R) {
                if (s->rp_Fc) RL_HIP(hipFree(s->rp_Fc));
                s->rp_Fc = nullptr;
                s->rp_Fc_R = 0;
                RL_HIP(hipMalloc((void**)&s->rp_Fc, (size_t)s->rp_R * s->n * sizeof(double)));
                permute_rows(s, s->rp_F, s->rp_Fc, s->rp_R, 1, (hipStream_t) nullptr);
                RL_HIP(hipGetLastError());
                RL_HIP(hipDeviceSynchronize());
                s->rp_Fc_R = s->rp_R;
            }
            return ski_rp_mvm(s, true, X, Y, nvec, s->has_noise ? s->noise_diag : nullptr, st,
                              nullptr);
        }
    }
    RL_TRY(ski_reserve_perm(s, nvec));
    permute_rows(s, X, s->P1, nvec, 0, st);
    RL_TRY(ski_mvm_int(s, s->P1, s->P2, nvec, st));
    permute_rows(s, s->P2, Y, nvec, 1, st);
    return RL_OK;
}

