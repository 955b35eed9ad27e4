// Host-side internals shared by the three translation units of the library (round 6 split of
// runlmc_hip.hip): rl_gridop.hip (grid operator, forms), rl_ski.hip (SKI operator products),
// rl_solve.hip (Krylov and direct solves, host helpers, gradient partial sums).  Kernels come
// from the rl_*.h headers; their non-template kernels have internal linkage, so that every
// translation unit instantiates only what it launches.
#pragma once
#include "../../include/runlmc_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "rl_kernels.h"
#include "rl_kernels2.h"
#include "rl_kernels3.h"
#include "rl_lowrank.h"
#include "rl_rowpoly.h"
#include "rl_filter.h"

#define RL_MAX_D 16

// ---------------------------------------------------------------------------
// errors (rl_gridop.hip owns the thread-local message)
// ---------------------------------------------------------------------------
int fail(int code, const std::string& msg);
#define RL_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(RL_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

#define RL_TRY(expr)              \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != RL_OK) return rc_; \
    } while (0)


// RUNLMC_TRACE=1: one line on stderr the first time each kernel variant is chosen
void trace_once(const char* what);
// true the first time a call site (its own bit mask `seen`) sees the current device
bool first_on_device(unsigned long long* seen);

// ---------------------------------------------------------------------------
// Environment switches.  Read ONCE, when a handle is created, into the handle:
// nothing a product or a solve does afterwards depends on the environment.
// What is left are (a) switches that turn a form or a fused path off for A/B
// measurements, (b) hooks the tests use to reach, at small sizes, the code paths
// that sizes select in production (chunked products, staged SpMVs, long-system
// loops).  Kernel-tuning knobs of earlier rounds (tile sizes, thread counts,
// stream counts, the round-1 kernels) are gone with the measurements that settled
// them.
// ---------------------------------------------------------------------------
struct RlKnobs {
    bool pow2_only = false;      // RUNLMC_POW2_ONLY: the reference's embedding length
    int chunk_mb = 0;            // RUNLMC_CHUNK_MB: intermediates per chunk of a batched product
    int two_streams = -1;        // RUNLMC_TWO_STREAMS=0/1 (default: by size)
    int affine = -1;             // RUNLMC_AFFINE=0/1: pair-affine order of the transform kernels
    int affine_kb = 0;           // RUNLMC_AFFINE_KB: L2-sized chunks of that many KB per XCD (experiment)
    int affine_max_kb = 2048;    // RUNLMC_AFFINE_MAX_KB: largest pair (KB of intermediates) that takes the order
    bool no_v1p = false;         // RUNLMC_NO_V1P: never the single-tile product
    int v1p_min = 0;             // RUNLMC_V1P_MIN
    bool no_lowrank = false;     // RUNLMC_NO_LOWRANK: no polynomial-subspace form
    bool no_filter = false;      // RUNLMC_NO_FILTER: no recursive-filter form
    bool sf_scan2 = false;       // RUNLMC_SF_SCAN2: the chunk chain reading the chunk states twice (k_sf_scan) on short grids too
    bool sf_carries1 = false;    // RUNLMC_SF_CARRIES1: the chunk states without the parity trick (k_sf_carries<2>)
    bool no_lr_bound = false;    // RUNLMC_NO_LR_BOUND: the polynomial verification without its operator-norm
                                 // bound (tests: what rounds 2-4 accepted)
    bool poly_round = false;     // RUNLMC_POLY_ROUND: polynomial rounds also on grids of 96..2047 points
    bool no_poly_round = false;  // RUNLMC_NO_POLY_ROUND
    long long lr_min = -1;       // RUNLMC_LR_MIN: batch gate of the structured forms
    bool staged_wt = false;      // RUNLMC_STAGED_WT: LDS-staged SpMVs whatever the size
    bool no_staged_wt = false;   // RUNLMC_NO_STAGED_WT
    bool no_w_poly = false;      // RUNLMC_NO_W_POLY: the W product reads the expanded grid vector
    bool no_rp = false;          // RUNLMC_NO_RP: no row-polynomial form of large solver rounds (rl_rowpoly.h)
    int rp_stagger = 7;          // RUNLMC_RP_STAGGER: k_rp_expand's workgroup b starts at vector (stagger b) mod nvec
    int rp_runlen = 0;           // RUNLMC_RP_RUNLEN: rows per run of k_rp_project (default: about n / 1024)
    bool no_rp_small = false;    // RUNLMC_NO_RP_SMALL: batches of <= 17 vectors through the general k_rp_project
    bool rp_pfuse = true;        // RUNLMC_NO_RP_PFUSE turns off MINRES's P inside the row-polynomial expansion
                                 // (k_minres2_ph + k_rp_expand<.., true>; rl_rowpoly.h RpPFuse): C5 round 2.30 ms
                                 // against 2.62 with P as its own kernel (profiles/r05/rp_pfuse_ab.txt, run 3)
    bool w_pfuse = true;         // RUNLMC_NO_W_PFUSE turns off MINRES's P inside the staged W product of rounds whose
                                 // operator is not in the row-polynomial form (k_spmv_w_staged_p, k_minres2_bv):
                                 // C5 matern round 3.71 ms against 4.18, mix 3.98 against 4.41
    bool no_rp_fuse = false;     // RUNLMC_NO_RP_FUSE: MINRES's B as its own kernel in row-polynomial rounds
    bool no_lr_small = false;    // RUNLMC_NO_LR_SMALL: small batches never take k_lr_small_*
    bool no_precond_approx = false;   // RUNLMC_NO_PRECOND_APPROX: no preconditioner for operators outside the polynomial form
    long long precond_hi_min = 100000;   // RUNLMC_PRECOND_HI_MIN: rows from which an operator without a polynomial row
                                      // gets the 96-function preconditioner
    int precond_hi_use = 0;           // RUNLMC_PRECOND_HI_USE: functions of it a factorisation uses (0: the library's rule)
    int precond_hi_rank = 192;         // RUNLMC_PRECOND_HI_RANK: its basis size (whole blocks of 48)
    bool no_precond_hi_mixed = false; // RUNLMC_NO_PRECOND_HI_MIXED: ... but not operators with SOME rows in the polynomial form
    bool precond_hi_passes = false;   // RUNLMC_PRECOND_HI_PASSES: its expansion block by block through the rank-48
                                      // kernel (as first built) instead of k_hz_expand_mm's one pass (A/B)
    bool no_precond_hi = false;       // RUNLMC_NO_PRECOND_HI: operators without a polynomial row keep the 48-function
                                      // preconditioner (not the 96-function one of rl_solve.hip: hz_*)
    int rp_fly = 1;              // RUNLMC_RP_FLY: bit 0: k_rp_expand computes F from the interpolation entries,
                                 // bit 1: k_rp_project too (otherwise from the table).  Measured (C5, per round):
                                 // expansion 257 -> 236 us at 129 vectors, 61 -> 42 at 17; projection level at rank
                                 // 24 (339 vs 347, 95 vs 92), behind at rank 36 (697 vs 610)
    int w_poly_rmax = 32;        // RUNLMC_W_POLY_RMAX: largest rank whose expansion the W kernel takes over
                                 // (36 measured: 791 us against 186 + 473 for expansion + staged W per C5 round)
    bool no_sort = false;        // RUNLMC_NO_SORT: caller's data order inside the SKI handle
    long long ws_cache_mb = -1;  // RUNLMC_WS_CACHE_MB
    int solver_maxblk = 0;       // RUNLMC_SOLVER_MAXBLK
    bool no_fuse_w = false;      // RUNLMC_NO_FUSE_W
    bool no_fuse_wt = false;     // RUNLMC_NO_FUSE_WT
    bool no_graph = false;       // RUNLMC_NO_GRAPH: solver rounds launched eagerly
    bool minres_v1 = false;      // RUNLMC_MINRES_V1 (emulator build only: the four-kernel
                                 // iteration the tests hold the two-kernel rounds against)
};
// Three switches are the user's: RUNLMC_POW2_ONLY (the reference's embedding length),
// RUNLMC_WS_CACHE_MB (memory the solver keeps between calls), RUNLMC_TRACE (prints which
// kernels ran).  Every other one is an A/B or test hook and is read ONLY under
// RUNLMC_DEBUG=1 -- without it the library runs its defaults whatever the environment says
// (and names every hook it ignored on stderr, each once).
RlKnobs read_knobs();

template <class T>
static int upload(T** dev, const std::vector<T>& host) {
    RL_HIP(hipMalloc((void**)dev, std::max<size_t>(host.size(), 1) * sizeof(T)));
    if (!host.empty())
        RL_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return RL_OK;
}

static const size_t kLdsSoft = 76 * 1024;    // two workgroups per CU
static const size_t kLdsHard = 156 * 1024;   // one workgroup per CU (160 KiB LDS)

// ---------------------------------------------------------------------------
// grid operator
// ---------------------------------------------------------------------------
struct rl_gridop {
    RlKnobs kn;                 // environment switches as they were when the handle was created
    int device = 0;
    int D = 0, m = 0, L = 0, N1 = 0, N2 = 0;
    Geom geo{0, 0, 0};   // m1 == 0: 1-D grid; else an m1 x m2 grid (2-D BTTB)
    int colsA = 0;   // columns per k_cols_* workgroup
    int rowsB = 0;   // rows per k_rows_mix workgroup
    int rowsS = 0;   // rows per k_rows_spec workgroup
    int code1 = 0, code2 = 0;   // fused_code(N1), fused_code(N2); both != 0 -> v2 kernels
    bool v2 = false;
    bool rows3 = false;         // row kernel = k3_rows_mix (plan2 = {RA, RB, 2})
    // polynomial-subspace form for smooth kernels (rl_lowrank.h)
    bool lr_try = false;        // eligible: 1-D grid, long enough, not switched off
    bool lr_ok = false;         // verified against the FFT path for the current parameters
    bool lr_round_try = false;  // the solver's polynomial rounds may use this grid
    // D > RL_MAX_D outputs ("wide" operator, round 4): the D x D mix does not fit the row
    // kernels' registers, so the Toeplitz blocks run through a child handle with ONE output
    // on nvec * D rows (T_q applied to every row: rl_gridop_mvm_top of the child, in
    // whatever form that top takes) and k_wide_mix applies the dense couplings,
    //   Y[v][a] (+)= sum_b B_q[a][b] (T_q X)[v][b].
    // (reference kronecker.py:39-46 has no limit on D; Q products + Q mix passes)
    bool wide = false;
    rl_gridop* child = nullptr;
    double* wide_B = nullptr;   // [max_tops][D][D]
    double* wide_Z = nullptr;   // T_q X of one top: nvec * D * m
    size_t wide_Z_cap = 0;
    bool defer_expand = false;  // ski_mvm_int: leave the expansion to the W kernel (k_spmv_w_poly) ...
    bool expand_deferred = false;   // ... done: lr_zhat holds the mixed coefficients of the batch
    int lr_rejects = 0;         // consecutive parameter sets with a top the verification rejected
    int lr_skip = 0;            // parameter updates the verification still sits out (back-off:
                                // 0, 1, 3 ... 31 updates after 1, 2, 3 ... rejections in a row)
    double* lr_stat = nullptr;  // dev verdict records of the verification; lr_C lives behind them
    bool lr_dirty = false;      // parameters changed since the last verification: the
                                // set-time work (lr_setup) runs when the first batch above
                                // the gate asks for it -- small-batch users never pay it
    std::vector<double> lr_A, lr_W, lr_kap;    // the parameters lr_setup will need
    std::vector<int> lr_Qi;
    bool lr_bypass = false;     // set while the FFT path is wanted (set-time verification)
    int lr_r = 0;               // basis size in use (24 / 32 / 48)
    int lr_rank_hint = 0;       // first rank the verification tries (rl_gridop_set_rank_hint)
    size_t lr_min = 0;          // batches below this many elements stay on the FFT path
    double* lr_beta = nullptr;  // dev [RL_LR_RMAX] recurrence coefficients
    double* lr_nu = nullptr;    // dev [RL_LR_RMAX] normalisation
    double* lr_phiJ = nullptr;  // dev [RL_LR_RMAX][m]
    double* lr_C = nullptr;     // dev [max_tops][r][r]
    double* lr_M = nullptr;     // dev [D][24][D][24]: the whole coefficient map (polynomial rounds)
    std::vector<double> lr_hC;  // host copy of lr_C for it
    double* lr_spart = nullptr; // dev [nvec][D][nseg][r]: partial sums of k_lr_small_project
    size_t lr_spart_cap = 0;
    double* lr_Cx = nullptr;    // dev [RMAX][RMAX]: scratch of lr_all_coeffs
    double* lr_Mf = nullptr;    // dev [D][r][D][r]: the coefficient map for k_lr_small_expand (any rank)
    size_t lr_Mf_cap = 0;
    bool lr_Mf_ok = false;      // ... built for the current parameters
    std::vector<double> lr_hB;  // host [Q][D][D]: the couplings lr_B holds (direct solves, rl_direct.h)
    std::vector<double> lr_hnu; // host copy of lr_nu
    unsigned long long param_ver = 0;   // bumped by every parameter update (what a factorisation was built for)
    double* lr_B = nullptr;     // dev [max_tops][D][D]
    double* lr_eye = nullptr;   // dev [D][D]
    double* lr_part = nullptr;  // projection partial sums
    size_t lr_part_cap = 0;
    double* lr_zhat = nullptr;  // mixed coefficients [rows][r]
    size_t lr_zhat_cap = 0;
    double* lr_scr = nullptr;   // set-time scratch: 3 vectors of D*m + partial maxima
    double* lr_pw = nullptr;    // the power iteration's vectors (lr_verify (iv))
    double* lr_sel = nullptr;   // ... and the selector couplings of its groups of tops
    std::vector<double> lr_vstat;   // host [max_tops][4]: the last verification's measurements per top
                                    // (trial ratio, tail ratio, ||E v||, ||T w||: rl_gridop_form_stats)
    double* lr_Cc = nullptr;    // dev [max_tops][r][r]: C of the polynomial tops only, contiguous
    double* lr_Bc = nullptr;    // dev [max_tops][D][D]: their couplings (operators that mix forms)
    int lr_np = 0;              // polynomial tops among the Q
    // per-top forms (decided by forms_setup at every parameter update):
    //   0 transform kernels, 1 polynomial-subspace form, 2 recursive filter (rl_filter.h)
    std::vector<double> h_tops;     // host copy of the top rows [Q][m]
    std::vector<int> top_form;
    bool st_ok = false;         // every top is 1 or 2 and at least one is 2: the operator runs as
                                // polynomial part + filter part, nothing through the transforms
    bool sf_try = false;        // eligible for the filter form: 1-D grid, not switched off
    int sf_n = 0;               // filter tops
    int sf_nfac = 0;            // rank-one factors that belong to them
    int sf_ns = 2;              // states per direction the operator's filters need (2 or 3)
    std::vector<int> sf_slot;   // per top: its place among the filter tops, or -1
    std::vector<int> sf_top_ns; // per top: 2 or 3
    SfTop* sf_tops = nullptr;   // dev [max_tops]
    double* sf_pwp = nullptr;   // dev [max_tops][2 G]: the chunk states' parity weights (k_sf_carries2)
    double* sf_blob = nullptr;  // dev: the filter part's block for k_sf_apply (rl_filter.h)
    double* sf_blob_top = nullptr;  // dev [max_tops][...]: the same for each top alone (B = I)
    double* sf_pw = nullptr;    // dev [max_tops][G + 1]
    double* sf_kappa = nullptr; // dev [max_tops][D]
    double* sf_facA = nullptr;  // dev [max_fac][D]
    double* sf_facAW = nullptr; // dev [max_fac][D]
    int* sf_facJ = nullptr;     // dev [max_fac]
    double* sf_E = nullptr;     // chunk states [nchunks][rows][NF][2][NS]
    size_t sf_E_cap = 0;
    double* sf_Cin = nullptr;   // incoming states [nchunks][nvec][nchan][2][NS]
    int* sf_next = nullptr;     // k_sf_apply's tile counter (zeroed by k_sf_scan)
    size_t sf_Cin_cap = 0;
    double* mixtab = nullptr;   // dev [D + nfac][L]: dc rows then gs rows (k_mix_tables)
    size_t mixtab_rows = 0;     // rows allocated
    bool mixtab_ok = false;     // tables match the current parameters
    int max_tops = 0;
    FftPlan plan1, plan2;
    cplx *tw1 = nullptr, *tw2 = nullptr, *twlo = nullptr, *twhi = nullptr;
    int* freq1 = nullptr;
    TwiddleL twl;
    std::vector<int> h_freq1, h_freq2;
    // parameters
    int Q = 0, nfac = 0;
    double* tops = nullptr;    // dev [max_tops][m]
    double* spec = nullptr;    // dev [max_tops][L]
    double* facA = nullptr;    // dev [max_fac][D]
    double* facW = nullptr;    // dev [max_fac]
    int* facQ = nullptr;       // dev [max_fac]
    double* kappa = nullptr;   // dev [max_tops][D]
    double* ones = nullptr;    // dev [D]  (identity mix for mvm_top)
    int max_fac = 0;
    // workspace: packed intermediates [pairs][D][L] complex
    cplx* T = nullptr;
    size_t T_pairs = 0;
    // a second chunk of intermediates and a side stream: consecutive chunks of a
    // large batched product run on two streams, so that one chunk's kernels fill
    // the compute units another chunk's tail leaves idle
    cplx* T2[3] = {nullptr, nullptr, nullptr};     // workspaces of the side streams
    size_t T2_pairs = 0;
    int nside = 0;              // side streams prepared (product runs on 1 + nside streams)
    cplx* Tcur = nullptr;       // workspace of the chunk being launched
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    size_t chunk_pairs = 1;
    bool affine = false;        // pair-affine order of the three transform kernels (rl_kernels2.h)
    // single-tile product (k1_product): grids short enough that all D transforms
    // of a pair fit one LDS tile
    bool v1p = false;
    FftPlan planL;
    cplx* twL = nullptr;       // exp(-2 pi i k / L)
    double* spec1 = nullptr;   // dev [max_tops][L], scrambled order of the single-level transform
    size_t lds1 = 0;
    int thr1 = 256;
    int v1p_min = 64;          // vectors from which the single-tile product is used
};

// ---------------------------------------------------------------------------
// SKI operator
// ---------------------------------------------------------------------------
// one W K_UU W^T term of the operator (kernels that share an
// active-dimension set share a grid, hence a term)
struct SkiTerm {
    rl_gridop* g = nullptr;
    int ngrid = 0;
    int *W_indptr = nullptr, *W_indices = nullptr;
    double* W_data = nullptr;
    int *WT_indptr = nullptr, *WT_indices = nullptr;
    double* WT_data = nullptr;
    int nnzWT = 0;
    // Interpolation structure, when W has it (every row at most 4 entries in
    // consecutive columns, rows sorted by first column -- cubic interpolation of
    // sorted inputs): W as base column + 4 weights per row, and W^T rebuilt so
    // that the data rows of every grid row are one CONSECUTIVE range starting
    // at WT_lo[r] (zero weights kept).  Both remove one level of dependent
    // loads from the fused products of the small-batch solver.
    int* W4_base = nullptr;
    double* W4_w = nullptr;
    int* WT_lo = nullptr;
    // largest data range / entry count of any RL_THREADS consecutive grid rows
    // (k_spmv_wt_staged stages them in LDS); 0 = no structure
    int wt_xmax = 0, wt_emax = 0;
    int w_xmax = 0;            // largest grid range of RL_THREADS consecutive data rows
    std::vector<int> h_base;   // host copy of W4_base (row blocks per output, polynomial rounds)
};

// buffers of one rl_solve_batch call
struct SolverWork {
    double* vec[10] = {nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr};
    double* S[2] = {nullptr, nullptr};
    int* I = nullptr;
    double* part[4] = {nullptr, nullptr, nullptr, nullptr};
    int* count = nullptr;   // [0] active systems, [1] global iteration counter, [2] done blocks
    double* resid = nullptr; // [nrhs] explicit residual norms
    double* lanczos = nullptr;
};

struct rl_ski {
    RlKnobs kn;         // environment switches as they were when the handle was created
    // Lanczos coefficients of a solve: kept between calls (a hipMalloc / hipFree
    // pair per solve costs ~200 us, as much as four C2 solver rounds)
    double* lanczos_buf = nullptr;
    size_t lanczos_bytes = 0;
    int device = 0;     // copy of g->device: the grid operator may be gone at destroy time
    std::vector<SkiTerm> extra;   // terms beyond the first (rl_ski_add_term)
    int max_ngrid = 0;
    rl_gridop* g = nullptr;
    int n = 0, ngrid = 0, nnz = 0, nnzWT = 0;
    int *W4_base = nullptr, *WT_lo = nullptr;     // interpolation structure (SkiTerm)
    int wt_xmax = 0, wt_emax = 0, w_xmax = 0;
    double* W4_w = nullptr;
    int *W_indptr = nullptr, *W_indices = nullptr;
    double* W_data = nullptr;
    int *WT_indptr = nullptr, *WT_indices = nullptr;
    double* WT_data = nullptr;
    double* noise_diag = nullptr;   // dev [n]
    bool has_noise = false;
    double *G1 = nullptr, *G2 = nullptr;   // dev [cap][D*m] grid-side temporaries
    int cap = 0;
    // Internal row order: data points sorted by grid position (W, WT and
    // noise_diag above are stored in THAT order); perm[i] = caller's row of
    // internal row i.  P1/P2: dev [pcap][n] staging for caller-order entry points.
    // polynomial rounds of small solves (rl_solver.h, Minres2Bufs::poly_part): row
    // blocks that lie inside one output, first block of each output, partial sums
    std::vector<int> h_base;
    int *poly_tab = nullptr, *poly_ob = nullptr;
    int poly_nblk = 0;              // 0: not built yet, -1: not applicable
    double* poly_part = nullptr;
    size_t poly_part_cap = 0;
    // row-polynomial form of large solver rounds (rl_rowpoly.h): F = W Phi, built per rank
    double* rp_F = nullptr;
    int rp_R = 0;                   // rank F was built for (0: none)
    double* rp_Fc = nullptr;        // the same in the CALLER's row order (rl_ski_mvm: no row permutations)
    int rp_Fc_R = 0;
    int* rp_base_c = nullptr;       // interpolation entries in the caller's row order (FLY kernels)
    double* rp_w4_c = nullptr;
    int *rp_runs = nullptr, *rp_run_ptr = nullptr, *rp_out_end = nullptr;
    int rp_nruns = 0;
    double* rp_part = nullptr;
    size_t rp_part_cap = 0;
    double* rp_nrm = nullptr;       // fused B: [nrhs] coefficients + [nrhs][rp_nruns] partial norms
    size_t rp_nrm_cap = 0;
    RpFuse rp_fuse{nullptr, nullptr, nullptr};   // set by the solver around ONE operator product
    // MINRES's P inside the expansion (rl_rowpoly.h RpPFuse): set by the solver around ONE
    // operator product together with rp_mid, which launches P's scalar head between the
    // coefficient map and the expansion; rp_pp: the head's coefficients [nrhs][RL_RP_PCW] and
    // three arrays of [nrhs][ceil(n / 256)] partial sums
    RpPFuse rp_pfuse{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::function<void(hipStream_t)> rp_mid;
    double* rp_pp = nullptr;
    size_t rp_pp_cap = 0;
    std::vector<int> eps_end;       // noise in runs: rows [eps_end[k-1], eps_end[k]) carry eps_val[k]
    std::vector<double> eps_val;    // (empty: more than RL_MAX_D runs)
    bool permuted = false;
    int* perm = nullptr;
    std::vector<int> h_perm;
    // rl_ski_mvm's row-polynomial form in the CALLER's row order reuses the runs, the output
    // borders and the noise array of the SORTED order: valid only when every output's rows
    // occupy the same range in both orders (checked once on the host, caller_order_same) and
    // the noise array reads the same in both (rl_ski_set_noise).  -1: not checked yet
    int caller_ranges_same = -1;
    bool caller_noise_same = true;
    double *P1 = nullptr, *P2 = nullptr;
    int pcap = 0;
    hipStream_t solver_stream = nullptr;   // capturable stream of rl_solve_batch
    int* pin_count = nullptr;              // pinned host [2]: active-system counts in flight
    hipEvent_t count_ev[2] = {nullptr, nullptr};
    // the solver's buffers are kept between calls (a solve frees eighteen of them
    // and a hipFree is ~100 us: 1.7 of the 2.9 ms a C2 solve spent outside its
    // rounds); capacities in elements.  RUNLMC_WS_CACHE_MB bounds what is kept.
    SolverWork ws;
    bool ws_valid = false;
    size_t ws_vec_cap = 0, ws_rhs_cap = 0, ws_part_cap = 0;
    // direct solves through the polynomial form (rl_direct.h): K~ = F M F^T + E
    unsigned long long noise_ver = 0;   // bumped by rl_ski_set_noise
    std::vector<double> h_noise;        // host copy of noise_diag (internal row order)
    std::vector<int> h_run_ptr, h_out_end;   // host copies of rp_run_ptr / rp_out_end
    std::vector<double> dz_U;           // host [D][r][r]: F_d^T F_d on the unnormalised q_j, per (handle, rank)
    int dz_U_R = 0;
    double* dz_Zt = nullptr;            // dev [D r][D r]: the solve map, scalings folded in
    size_t dz_Zt_cap = 0;
    double* dz_inv = nullptr;           // dev [n]: 1 / eps per row
    std::vector<double> dz_eps;         // host [D]: the noise level of each output ...
    const char* dz_eps_why = nullptr;   // ... or why there is none (not constant per output, not positive)
    unsigned long long dz_eps_ver = ~0ull;   // noise_ver those (and dz_inv) were derived from
    bool dz_valid = false;              // ... for the parameters / noise of the versions below
    bool dz_exact = false;              // every top row in the polynomial form: the factorisation IS K~^-1;
                                        // else it inverts the operator's projection on the subspace: a preconditioner
    const char* dz_fail_why = nullptr;  // "not available" decided for the versions below (not recomputed per solve)
    unsigned long long dz_fail_param_ver = 0, dz_fail_noise_ver = 0;
    int dz_fail_streak = 0, dz_fail_skip = 0;   // back-off of the attempts after failures in a row
    double *dz_p = nullptr, *dz_q = nullptr;       // dev [cap][n]: PCG's direction and operator product
    size_t dz_pq_cap = 0;
    double* dz_scal = nullptr;          // dev [cap][2]: PCG's (rho, rho_prev)
    unsigned long long dz_param_ver = 0, dz_noise_ver = 0;
    int dz_R = 0;
    // the 96-function preconditioner of an operator without a polynomial row (rl_solve.hip: hz_*)
    bool dz_want_sample = false;        // rl_ski_precond_sample was called: factorisations keep the square-root map
    double* dz_Zh = nullptr;            // dev [D r][D r]: that map, transposed (rl_solve.hip dz_host_map)
    size_t dz_Zh_cap = 0;
    bool dz_Zh_valid = false;
    double* dz_isq = nullptr;           // dev [n]: 1 / sqrt(eps) per row
    unsigned long long dz_isq_ver = ~0ull;
    double* dz_smp = nullptr;           // dev [cap][n]: scaled probes
    size_t dz_smp_cap = 0;
    double* dz_rec = nullptr;           // dev [cap][nrhs][2]: conjugate gradients' scalars of a recorded run
    size_t dz_rec_cap = 0;
    bool dz_hz = false;                 // the valid factorisation is THAT one (dz_Zt: [D 96][D 96])
    int hz_R = 0;                       // its basis size (blocks of 48; "96" below stands for it)
    bool hz_traced = false;
    int hz_Ruse = 0;                    // ... of which the current factorisation uses the first hz_Ruse
    bool hz_basis_tried = false;        // basis generated (or found unusable) once per handle
    const char* hz_why = nullptr;       // why the handle has none (decided once)
    std::vector<double> hz_hnu;         // host [96]: normalisation of the basis
    double* hz_beta = nullptr;          // dev [96]: recurrence coefficients
    double* hz_phi = nullptr;           // dev [96 rounded up to D][m]: the functions on the grid
    double* hz_tphi = nullptr;          // dev, same size: a top row applied to them
    double* hz_C = nullptr;             // dev [96][96]
    double* hz_F = nullptr;             // dev [96][n]: F = W Phi, unnormalised, degree-major
    double* hz_ones = nullptr;          // dev [n]
    std::vector<double> hz_U;           // host [D][96][96]: F_d^T F_d, once per handle
    double* hz_part = nullptr;          // dev [2][runs][cap][48]
    double* hz_zhat = nullptr;          // dev [2][cap][D][48]
    double* hz_tmp = nullptr;           // dev [cap][n]: the first half's expansion
    double* hz_S = nullptr;             // dev [cap][D R]: projections summed over the runs
    double* hz_P = nullptr;             // dev [RL_HZ_FS][cap][D R]: the map's partial products
    size_t hz_vec_cap = 0;
    double dz_logdet = 0.0;             // log det K~ of that factorisation
    double dz_cond = 0.0;               // ratio of the largest to the smallest pivot of chol(S), squared
    double *dz_res = nullptr, *dz_cor = nullptr;   // dev [cap][n]: residuals, corrections
    size_t dz_vec_cap = 0;
    double* dz_part = nullptr;          // dev [cap][RL_DZ_NBLK] partial sums, then [cap] norms
    int* dz_go = nullptr;               // dev [cap]: systems still being refined
    size_t dz_rhs_cap = 0;
};

// destroys a half-built handle on every early return of a create function
template <class H, int (*Destroy)(H*)>
struct HandleGuard {
    H* h;
    explicit HandleGuard(H* h_) : h(h_) {}
    ~HandleGuard() { if (h) (void)Destroy(h); }
    H* release() { H* r = h; h = nullptr; return r; }
};

// ---------------------------------------------------------------------------
// functions one translation unit defines and another calls
// ---------------------------------------------------------------------------
// rl_gridop.hip
bool stream_capturing(hipStream_t stream);
int lr_ensure(rl_gridop* g);
int lr_prepare(rl_gridop* g, int nvec);
int lr_reserve(rl_gridop* g, int nvec);
int gridop_prepare(rl_gridop* g, int nvec);
int lr_all_coeffs(rl_gridop* g, int R, std::vector<double>* hC, std::vector<char>* exact,
                  std::vector<double>* captured);
// (one chunk of a batched product on the transform kernels, optionally gathering W^T x itself:
// the solver's fused rounds call them directly)
int mvm_chunk_v2(rl_gridop* g, const MixParams& mp, const double* Xc, double* Yc, int nv, size_t pairs,
                 hipStream_t st, const Gather* gather = nullptr, int* bump = nullptr,
                 cplx* Tbuf = nullptr);
int mvm_chunk_v1(rl_gridop* g, const MixParams& mp, const double* Xc, double* Yc, int nv, size_t pairs,
                 hipStream_t stream, const Gather* gather = nullptr, int* bump = nullptr);
// rl_ski.hip
int upload_raw(void** dev, const void* host, size_t bytes);
int ski_reserve(rl_ski* s, int nvec);
int ski_wt_int(rl_ski* s, const double* Xp, double* G, int nvec, hipStream_t st, int* bump = nullptr);
int ski_reserve_perm(rl_ski* s, int nvec);
void permute_rows(rl_ski* s, const double* X, double* Y, int nvec, int scatter, hipStream_t st);
bool w_staged_p_ok(const rl_ski* s, int nvec);
bool rp_ok(const rl_ski* s, int nvec);
int rp_prepare(rl_ski* s, int nvec);
bool rp_ready(const rl_ski* s, int nvec);
int ski_mvm_int(rl_ski* s, const double* Xp, double* Yp, int nvec, hipStream_t st,
                int* bump = nullptr, bool noise = true);
// rl_solve.hip
void free_work(SolverWork& w);

