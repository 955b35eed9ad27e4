// Grid operator K_UU = sum_q B_q (x) T_q: handles, parameter updates, the forms of the product
// (transform kernels, polynomial-subspace form, recursive filters), products (see
// include/runlmc_hip.h; one of the three translation units of librunlmc_hip.so, rl_host.h).
#include "rl_host.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

extern "C" const char* rl_last_error(void) { return g_err.c_str(); }
extern "C" const char* rl_backend(void) { return RL_BACKEND_NAME; }
extern "C" int rl_abi_version(void) { return RL_ABI_VERSION; }
extern "C" int rl_device_count(int* count) {
    if (!count) return fail(RL_EINVAL, "count is NULL");
    RL_HIP(hipGetDeviceCount(count));
    return RL_OK;
}

// ---------------------------------------------------------------------------
// small host helpers
// ---------------------------------------------------------------------------
// RUNLMC_TRACE=1: one line on stderr the first time each kernel variant is chosen
void trace_once(const char* what) {
    static const bool on = getenv("RUNLMC_TRACE") != nullptr;
    if (!on) return;
    static std::vector<std::string> seen;
    for (const std::string& s : seen)
        if (s == what) return;
    seen.push_back(what);
    fprintf(stderr, "[runlmc] %s\n", what);
}

// One-time per DEVICE work (the > 64 KiB dynamic-LDS opt-ins are function
// attributes of the code object loaded on each device): true the first time a
// call site sees the current device.  `seen` is the call site's own bit mask.
bool first_on_device(unsigned long long* seen) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (*seen & bit) return false;
    *seen |= bit;
    return true;
}

RlKnobs read_knobs() {
    RlKnobs k;
    const char* dbg = getenv("RUNLMC_DEBUG");
    const bool debug = dbg != nullptr && dbg[0] != '\0' && dbg[0] != '0';
    auto ignored = [](const char* n) {
        // (every ignored variable is named, each once per process)
        static std::vector<std::string> told;
        if (std::find(told.begin(), told.end(), std::string(n)) == told.end()) {
            told.push_back(n);
            fprintf(stderr, "runlmc_hip: %s is a debug switch and is ignored without RUNLMC_DEBUG=1\n", n);
        }
    };
    auto user_flag = [](const char* n) { return getenv(n) != nullptr; };
    auto user_num = [](const char* n, long long dflt) {
        const char* e = getenv(n);
        return e ? atoll(e) : dflt;
    };
    auto flag = [&](const char* n) {
        if (getenv(n) == nullptr) return false;
        if (!debug) ignored(n);
        return debug;
    };
    auto num = [&](const char* n, long long dflt) {
        const char* e = getenv(n);
        if (e == nullptr) return dflt;
        if (!debug) ignored(n);
        return debug ? atoll(e) : dflt;
    };
    k.pow2_only = user_flag("RUNLMC_POW2_ONLY");
    k.chunk_mb = (int)num("RUNLMC_CHUNK_MB", 0);
    k.two_streams = (int)num("RUNLMC_TWO_STREAMS", -1);
    k.affine = (int)num("RUNLMC_AFFINE", -1);
    k.affine_kb = (int)num("RUNLMC_AFFINE_KB", 0);
    k.affine_max_kb = (int)num("RUNLMC_AFFINE_MAX_KB", 2048);
    k.no_v1p = flag("RUNLMC_NO_V1P");
    k.v1p_min = (int)num("RUNLMC_V1P_MIN", 0);
    k.no_lowrank = flag("RUNLMC_NO_LOWRANK");
    k.no_filter = flag("RUNLMC_NO_FILTER");
    k.no_lr_bound = flag("RUNLMC_NO_LR_BOUND");
    k.sf_carries1 = flag("RUNLMC_SF_CARRIES1");
    k.sf_scan2 = flag("RUNLMC_SF_SCAN2");
    k.poly_round = flag("RUNLMC_POLY_ROUND");
    k.no_poly_round = flag("RUNLMC_NO_POLY_ROUND");
    k.lr_min = num("RUNLMC_LR_MIN", -1);
    k.staged_wt = flag("RUNLMC_STAGED_WT");
    k.no_staged_wt = flag("RUNLMC_NO_STAGED_WT");
    k.no_w_poly = flag("RUNLMC_NO_W_POLY");
    k.no_rp = flag("RUNLMC_NO_RP");
    k.rp_stagger = (int)num("RUNLMC_RP_STAGGER", 7);
    k.no_lr_small = flag("RUNLMC_NO_LR_SMALL");
    k.no_precond_approx = flag("RUNLMC_NO_PRECOND_APPROX");
    k.no_precond_hi = flag("RUNLMC_NO_PRECOND_HI");
    k.precond_hi_passes = flag("RUNLMC_PRECOND_HI_PASSES");
    k.no_precond_hi_mixed = flag("RUNLMC_NO_PRECOND_HI_MIXED");
    k.precond_hi_min = num("RUNLMC_PRECOND_HI_MIN", 100000);
    k.precond_hi_rank = (int)num("RUNLMC_PRECOND_HI_RANK", 192);
    k.precond_hi_use = (int)num("RUNLMC_PRECOND_HI_USE", 0);
    k.rp_fly = (int)num("RUNLMC_RP_FLY", 1);
    k.no_rp_small = flag("RUNLMC_NO_RP_SMALL");
    k.no_rp_fuse = flag("RUNLMC_NO_RP_FUSE");
    k.rp_pfuse = !flag("RUNLMC_NO_RP_PFUSE");
    k.w_pfuse = !flag("RUNLMC_NO_W_PFUSE");
    k.rp_runlen = (int)num("RUNLMC_RP_RUNLEN", 0);
    // (k_spmv_w_poly exists for ranks 24, 32 and 36: a larger value would hand it coefficients
    // of a rank it has no instantiation for)
    k.w_poly_rmax = std::min(36, (int)num("RUNLMC_W_POLY_RMAX", 32));
    k.no_sort = flag("RUNLMC_NO_SORT");
    k.ws_cache_mb = user_num("RUNLMC_WS_CACHE_MB", -1);
    k.solver_maxblk = (int)num("RUNLMC_SOLVER_MAXBLK", 0);
    k.no_fuse_w = flag("RUNLMC_NO_FUSE_W");
    k.no_fuse_wt = flag("RUNLMC_NO_FUSE_WT");
    k.no_graph = flag("RUNLMC_NO_GRAPH");
#if defined(RL_EMU)
    k.minres_v1 = flag("RUNLMC_MINRES_V1");
#endif
    return k;
}

static int ilog2(int x) {
    int l = 0;
    while ((1 << l) < x) ++l;
    return l;
}

// Radix schedules.  Lengths are odd * 2^a with odd in {1, 3, 5, 9, 15, 25}; odd
// factors always come FIRST (the register first pass has no power-of-two
// assumption; an LDS pass needs a power-of-two butterfly distance, which holds
// for a second odd pass because what remains after it is a power of two).  Power-of-two parts 8 / 16 / 64 / 128 / 256 / 512 / 1024 use
// the schedules the fused kernels are instantiated for.
static FftPlan make_plan(int n) {
    FftPlan p;
    p.n = n;
    p.npass = 0;
    for (int i = 0; i < RL_MAX_PASSES; ++i) p.radix[i] = 1;
    int rem = n;
    // at most two odd passes (9 = 3*3, 15 = 3*5, 25 = 5*5); after the last odd
    // pass the remaining length is a power of two, so every later butterfly
    // distance is one too
    for (int odd : {3, 5})
        while (rem % odd == 0 && p.npass < 2) {
            p.radix[p.npass++] = odd;
            rem /= odd;
        }
    const int fixed[7][4] = {{8, 8, 0, 0},     {16, 16, 0, 0},   {64, 8, 8, 0},
                             {128, 8, 16, 0},  {256, 16, 16, 0}, {512, 8, 8, 8},
                             {1024, 8, 8, 16}};
    for (const auto& f : fixed)
        if (f[0] == rem) {
            for (int i = 1; i < 4 && f[i]; ++i) p.radix[p.npass++] = f[i];
            return p;
        }
    while (rem > 1) {
        int r = 8;
        while (rem % r) r /= 2;
        p.radix[p.npass++] = r;
        rem /= r;
    }
    return p;
}

// Row plans of the third-generation row kernel (rl_kernels3.h): N2 = RA * RB * 2
// with the last radix-2 pass done by the mix threads.  Returns false when the
// length has no such instantiation.
static bool make_plan_rows3(int n, FftPlan* p) {
    int ra, rb;
    switch (n) {
        case 128: ra = 8; rb = 8; break;
        case 256: ra = 16; rb = 8; break;
        case 512: ra = 16; rb = 16; break;
        default: return false;
    }
    p->n = n;
    p->npass = 3;
    for (int i = 0; i < RL_MAX_PASSES; ++i) p->radix[i] = 1;
    p->radix[0] = ra;
    p->radix[1] = rb;
    p->radix[2] = 2;
    return true;
}

// code of the fused (register first/last pass) instantiation that runs a
// plan: first radix * 100 + last radix; 0 if there is none
static int fused_code(const FftPlan& p) {
    if (p.npass < 2 || p.npass > 4) return 0;
    const int ra = p.radix[0], rb = p.radix[p.npass - 1];
    const int code = ra * 100 + rb;
    switch (code) {
        case 808: case 816: case 1616: return p.n >= 64 ? code : 0;
        case 308: case 316: case 508: case 516: return code;
        default: return 0;
    }
}

// Embedding length: the smallest odd * 2^k, odd in {1,3,5,9,15,25}, >= 2m (any
// length >= 2m - 1 embeds the Toeplitz matrix exactly; the reference uses the
// next power of two, bttb.py:16-19).  Split L = N1 * N2 with N2 a power of two
// (row transforms) and the odd factor in N1.  RUNLMC_POW2_ONLY=1 forces the
// reference's length.
static void choose_length(int m, bool pow2_only, int* L_out, int* N1_out, int* N2_out) {
    long best = 0;
    int best_odd = 1;
    for (int odd : {1, 3, 5, 9, 15, 25}) {
        if (odd != 1 && pow2_only) continue;
        long L = odd;
        while (L < 2L * m || L < 16) L *= 2;
        if (odd != 1 && L / odd < 64 * 8) continue;   // too short to be worth it
        if (best == 0 || L < best) { best = L; best_odd = odd; }
    }
    const int L = (int)best;
    const int P = L / best_odd;              // power-of-two part
    const int l = ilog2(P);
    int N2 = 1 << ((l + 1) / 2);
    if (best_odd != 1) {
        // prefer a split the fused kernels cover: N2 in 64..1024 and the
        // power-of-two part of N1 in {8, 16, 64, 128, 256}
        int bestN2 = 0;
        double bestScore = 1e300;
        for (int n2 = 64; n2 <= 1024 && n2 <= P / 8; n2 *= 2) {
            const int p1 = P / n2;
            if (p1 != 8 && p1 != 16 && p1 != 64 && p1 != 128 && p1 != 256) continue;
            const double score = std::fabs(std::log2((double)best_odd * p1 / n2));
            if (score < bestScore) { bestScore = score; bestN2 = n2; }
        }
        if (bestN2) N2 = bestN2;
    }
    *L_out = L;
    *N2_out = N2;
    *N1_out = L / N2;
}

// position -> frequency of the in-place DIF graph (tests/flow_model.py)
static std::vector<int> position_to_freq(const FftPlan& p) {
    std::vector<int> f(p.n);
    for (int pos = 0; pos < p.n; ++pos) {
        int rem = pos, ns = p.n, mult = 1, k = 0;
        for (int s = 0; s < p.npass; ++s) {
            int sub = ns / p.radix[s];
            int d = rem / sub;
            rem -= d * sub;
            k += d * mult;
            mult *= p.radix[s];
            ns = sub;
        }
        f[pos] = k;
    }
    return f;
}

// exp(-2 pi i k / n), accurate to the last bit or so (long double + octant
// symmetry through cosl/sinl of a reduced argument)
static void root_of_unity(long k, long n, double* re, double* im) {
    k %= n;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    // reduce to first octant for accuracy
    long double ang = two_pi * (long double)k / (long double)n;
    *re = (double)cosl(ang);
    *im = (double)(-sinl(ang));
}

static std::vector<cplx> unity_table(long count, long stride, long n) {
    std::vector<cplx> t(count);
    for (long i = 0; i < count; ++i) root_of_unity(i * stride, n, &t[i].x, &t[i].y);
    return t;
}


static size_t lds_cols(const rl_gridop* g) {
    return ((size_t)g->N1 * g->colsA + g->N1) * sizeof(cplx);
}
static size_t lds_rows(int N2, int cols) {
    return ((size_t)N2 * (cols | 1) + N2) * sizeof(cplx);
}

static int ensure_workspace(rl_gridop* g, size_t pairs) {
    if (pairs <= g->T_pairs) return RL_OK;
    if (g->T) RL_HIP(hipFree(g->T));
    g->T = nullptr;
    g->T_pairs = 0;
    RL_HIP(hipMalloc((void**)&g->T, pairs * g->D * (size_t)g->L * sizeof(cplx)));
    g->T_pairs = pairs;
    return RL_OK;
}

template <int D>
static void set_lds_attr_rows() {
#if !defined(RL_EMU)
    (void)hipFuncSetAttribute((const void*)k_rows_mix<D>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k2_rows_mix<D, 8, 8>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k2_rows_mix<D, 8, 16>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k2_rows_mix<D, 16, 16>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
}
template <int D>
static void set_lds_attr_rows3() {
#if !defined(RL_EMU)
    (void)hipFuncSetAttribute((const void*)k3_rows_mix<D, 8, 8>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k3_rows_mix<D, 16, 8>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k3_rows_mix<D, 16, 16>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
}
template <int RA, int RB>
static void set_lds_attr_cols2() {
#if !defined(RL_EMU)
    (void)hipFuncSetAttribute((const void*)k2_cols_fwd<RA, RB, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k2_cols_fwd<RA, RB, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k2_cols_inv<RA, RB>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
}

static void set_lds_attrs() {
#if !defined(RL_EMU)
    static unsigned long long seen = 0;
    if (!first_on_device(&seen)) return;
    (void)hipFuncSetAttribute((const void*)k_cols_fwd,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_cols_inv,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_rows_spec,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    set_lds_attr_cols2<8, 8>(); set_lds_attr_cols2<8, 16>(); set_lds_attr_cols2<16, 16>();
    set_lds_attr_cols2<3, 8>(); set_lds_attr_cols2<3, 16>();
    set_lds_attr_cols2<5, 8>(); set_lds_attr_cols2<5, 16>();
    set_lds_attr_rows<1>();  set_lds_attr_rows<2>();  set_lds_attr_rows<3>();
    set_lds_attr_rows<4>();  set_lds_attr_rows<5>();  set_lds_attr_rows<6>();
    set_lds_attr_rows<7>();  set_lds_attr_rows<8>();  set_lds_attr_rows<9>();
    set_lds_attr_rows<10>(); set_lds_attr_rows<11>(); set_lds_attr_rows<12>();
    set_lds_attr_rows<13>(); set_lds_attr_rows<14>(); set_lds_attr_rows<15>();
    set_lds_attr_rows<16>();
    set_lds_attr_rows3<1>();  set_lds_attr_rows3<2>();  set_lds_attr_rows3<3>();
    set_lds_attr_rows3<4>();  set_lds_attr_rows3<5>();  set_lds_attr_rows3<6>();
    set_lds_attr_rows3<7>();  set_lds_attr_rows3<8>();  set_lds_attr_rows3<9>();
    set_lds_attr_rows3<10>(); set_lds_attr_rows3<11>(); set_lds_attr_rows3<12>();
    set_lds_attr_rows3<13>(); set_lds_attr_rows3<14>(); set_lds_attr_rows3<15>();
    set_lds_attr_rows3<16>();
#define RL_SF_ATTR(D_)                                                                \
    (void)hipFuncSetAttribute((const void*)k_sf_apply<2, D_>,                         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)hipFuncSetAttribute((const void*)k_sf_apply<3, D_>,                         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    RL_SF_ATTR(1); RL_SF_ATTR(2); RL_SF_ATTR(3); RL_SF_ATTR(4); RL_SF_ATTR(5); RL_SF_ATTR(6);
    RL_SF_ATTR(7); RL_SF_ATTR(8); RL_SF_ATTR(9); RL_SF_ATTR(10); RL_SF_ATTR(11); RL_SF_ATTR(12);
    RL_SF_ATTR(13); RL_SF_ATTR(14); RL_SF_ATTR(15); RL_SF_ATTR(16);
#undef RL_SF_ATTR
#define RL_LRS_ATTR(R_)                                                                      \
    (void)hipFuncSetAttribute((const void*)k_lr_small_project<R_>,                            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);         \
    (void)hipFuncSetAttribute((const void*)k_lr_small_expand<R_>,                             \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    RL_LRS_ATTR(24); RL_LRS_ATTR(32); RL_LRS_ATTR(36); RL_LRS_ATTR(40); RL_LRS_ATTR(48);
#undef RL_LRS_ATTR
    (void)hipFuncSetAttribute((const void*)k_sf_carries<2>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_sf_carries<3>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
}


static unsigned div_magic(unsigned d);

// single-tile product (k1_product<D>)
template <int D>
static void launch1p_d(rl_gridop* g, unsigned pairs, hipStream_t st, const double* X, double* Y,
                       int nvec, int mode, const MixParams& mp, double* spec_out) {
#if !defined(RL_EMU)
    static unsigned long long seen = 0;
    if (first_on_device(&seen)) {
        (void)hipFuncSetAttribute((const void*)k1_product<D>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    const size_t lds = ((size_t)g->L * (D | 1) + g->L) * sizeof(cplx);
    RL_LAUNCH((k1_product<D>), dim3(pairs), dim3(g->thr1), lds, st, X, Y, nvec, g->geo, mode,
              g->planL, g->twL, mp, spec_out);
}
static int launch1p(rl_gridop* g, int D, unsigned pairs, hipStream_t st, const double* X,
                    double* Y, int nvec, int mode, const MixParams& mp, double* spec_out) {
    switch (D) {
#define RL_CASE(d) \
    case d: launch1p_d<d>(g, pairs, st, X, Y, nvec, mode, mp, spec_out); return RL_OK;
        RL_CASE(1) RL_CASE(2) RL_CASE(3) RL_CASE(4) RL_CASE(5) RL_CASE(6) RL_CASE(7)
        RL_CASE(8) RL_CASE(9) RL_CASE(10) RL_CASE(11) RL_CASE(12) RL_CASE(13)
        RL_CASE(14) RL_CASE(15) RL_CASE(16)
#undef RL_CASE
        default: return fail(RL_ELIMIT, "unsupported D");
    }
}


static size_t lr_min_elements(const rl_gridop* g);
static int gridop_create_impl(int device, int D, int m, int m1, int m2, int max_tops,
                              rl_gridop** out);


extern "C" int rl_gridop_create(int device, int D, int m, int max_tops, rl_gridop** out) {
    return gridop_create_impl(device, D, m, 0, 0, max_tops, out);
}

extern "C" int rl_gridop_create_2d(int device, int D, int m1, int m2, int max_tops,
                                   rl_gridop** out) {
    if (m1 < 1 || m2 < 1) {
        if (out) *out = nullptr;
        return fail(RL_EINVAL, "rl_gridop_create_2d: m1, m2 must be >= 1");
    }
    if ((long)m1 * m2 > (1L << 27)) return fail(RL_ELIMIT, "rl_gridop_create_2d: grid too large");
    return gridop_create_impl(device, D, m1 * m2, m1, m2, max_tops, out);
}

static int gridop_create_impl(int device, int D, int m, int m1, int m2, int max_tops,
                              rl_gridop** out) {
    if (!out) return fail(RL_EINVAL, "out is NULL");
    *out = nullptr;
    if (D < 1 || m < 1 || max_tops < 1)
        return fail(RL_EINVAL, "rl_gridop_create: D, m, max_tops must be >= 1");
    if ((long)m > (1L << 27)) return fail(RL_ELIMIT, "rl_gridop_create: m too large");
    RL_HIP(hipSetDevice(device));
    set_lds_attrs();
    if (D > RL_MAX_D) {
        // wide operator: see rl_gridop::wide
        if (D > 4096) return fail(RL_ELIMIT, "rl_gridop_create: D > 4096 outputs not supported");
        rl_gridop* w = new rl_gridop;
        HandleGuard<rl_gridop, rl_gridop_destroy> wguard(w);
        w->kn = read_knobs();
        w->device = device;
        w->D = D;
        w->m = m;
        w->max_tops = max_tops;
        w->geo = Geom{m, m1, m2};
        w->wide = true;
        RL_TRY(gridop_create_impl(device, 1, m, m1, m2, max_tops, &w->child));
        w->L = w->child->L;
        w->N1 = w->child->N1;
        w->N2 = w->child->N2;
        w->colsA = w->child->colsA;
        w->rowsB = w->child->rowsB;
        w->chunk_pairs = w->child->chunk_pairs;
        w->max_fac = max_tops * D;
        RL_HIP(hipMalloc((void**)&w->wide_B, (size_t)max_tops * D * D * sizeof(double)));
        *out = wguard.release();
        return RL_OK;
    }

    rl_gridop* g = new rl_gridop;
    HandleGuard<rl_gridop, rl_gridop_destroy> guard(g);
    g->kn = read_knobs();
    g->device = device;
    g->D = D;
    g->m = m;
    g->max_tops = max_tops;
    int L;
    g->geo = Geom{m, m1, m2};
    if (m1 == 0) {
        choose_length(m, g->kn.pow2_only, &L, &g->N1, &g->N2);
    } else {
        // 2-D: one circulant embedding per axis (reference bttb.py:112: next
        // power of two of twice each size), floored at 4
        g->N1 = 4;
        while (g->N1 < 2 * m1) g->N1 *= 2;
        g->N2 = 4;
        while (g->N2 < 2 * m2) g->N2 *= 2;
        L = g->N1 * g->N2;
    }
    g->L = L;
    const int l = ilog2(L);
    // rows per (first-generation) row workgroup: the largest divisor of N1 that
    // fits the soft LDS budget and still leaves >= 16 workgroups per pair; if
    // even one row does not fit, shrink N2 (longer column transforms)
    auto pick_rows = [&](int cols_per_row) {
        int best = 0;
        for (int r = 1; r <= g->N1; ++r) {
            if (g->N1 % r) continue;
            if (lds_rows(g->N2, r * cols_per_row) > kLdsSoft) break;
            if (r > 1 && g->N1 / r < 16) break;
            best = r;
        }
        return best;
    };
    for (;;) {
        int R = pick_rows(D);
        if (R == 0 && lds_rows(g->N2, D) <= kLdsHard) R = 1;
        if (R > 0) { g->rowsB = R; break; }
        if (g->N2 <= 4 || m1 != 0)
            return fail(RL_ELIMIT, "rl_gridop_create: D * row length exceeds LDS");
        g->N2 /= 2;
        g->N1 *= 2;
    }
    g->rowsS = std::max(1, pick_rows(1));
    // columns per k_cols_* workgroup
    int C = std::min(32, g->N2);
    while (C > 1 && ((size_t)g->N1 * C + g->N1) * sizeof(cplx) > kLdsSoft) C /= 2;
    if (C < 8 && g->N2 >= 8) {
        C = 8;
        while (C > 1 && ((size_t)g->N1 * C + g->N1) * sizeof(cplx) > kLdsHard) C /= 2;
    }
    if (((size_t)g->N1 * C + g->N1) * sizeof(cplx) > kLdsHard)
        return fail(RL_ELIMIT, "rl_gridop_create: grid too long for one LDS column tile");
    g->colsA = C;

    g->plan1 = make_plan(g->N1);
    g->plan2 = make_plan(g->N2);
    g->code1 = fused_code(g->plan1);
    g->code2 = fused_code(g->plan2);
    bool rows_ok = g->code2 == 808 || g->code2 == 816 || g->code2 == 1616;
    // third-generation row kernel wherever its lengths apply and its unpadded
    // tile of one row fits (the spectra are built with the same plan below)
    if (g->code1 != 0 &&
        (size_t)g->N2 * D * sizeof(cplx) <= kLdsHard && make_plan_rows3(g->N2, &g->plan2)) {
        g->rows3 = true;
        g->code2 = g->plan2.radix[0] * 100 + g->plan2.radix[1];
        rows_ok = true;
    }
    g->v2 = g->code1 != 0 && rows_ok;
    // polynomial-subspace form (rl_lowrank.h): any 1-D grid of at least twice the largest
    // rank, decided per parameter set by verification -- and only when a batch above the
    // gate (2^20 elements) asks for it, so handles that see small batches only (the
    // real-data fits) never pay the verification.  (Until round 4 grids under 2048 points
    // were excluded: RBF at m = 1000 then saturated at 6-11 % on the single-tile kernel.)
    // (m < 2^28: k_lr_project addresses a row's elements by 32-bit byte offsets)
    g->lr_try = m1 == 0 && m >= 2 * RL_LR_RMAX && m < (1 << 28) && !g->kn.no_lowrank;
    // the solver's two-kernel polynomial rounds: grids of >= 2048 points by default,
    // shorter ones on request (RUNLMC_POLY_ROUND=1)
    g->lr_round_try = g->lr_try && (m >= 2048 || g->kn.poly_round);
    // recursive-filter form (rl_filter.h): any 1-D grid; decided per top row from the row
    g->sf_try = m1 == 0 && m >= 64 && !g->kn.no_filter;
    g->lr_min = lr_min_elements(g);
    g->h_freq1 = position_to_freq(g->plan1);
    g->h_freq2 = position_to_freq(g->plan2);

    int rc;
    if ((rc = upload(&g->tw1, unity_table(g->N1, 1, g->N1))) != RL_OK) return rc;
    if ((rc = upload(&g->tw2, unity_table(g->N2, 1, g->N2))) != RL_OK) return rc;
    const int shift = l / 2;
    if ((rc = upload(&g->twlo, unity_table(1L << shift, 1, L))) != RL_OK) return rc;
    if ((rc = upload(&g->twhi, unity_table(((long)L + (1L << shift) - 1) >> shift, 1L << shift,
                                           L))) != RL_OK) return rc;
    if ((rc = upload(&g->freq1, g->h_freq1)) != RL_OK) return rc;
    g->twl.lo = m1 == 0 ? g->twlo : nullptr;   // 2-D: no inter-step twiddle
    g->twl.hi = g->twhi;
    g->twl.shift = shift;
    g->twl.mask = (1 << shift) - 1;

    g->max_fac = max_tops * D;
    RL_HIP(hipMalloc((void**)&g->tops, (size_t)max_tops * m * sizeof(double)));
    RL_HIP(hipMalloc((void**)&g->spec, (size_t)max_tops * L * sizeof(double)));
    RL_HIP(hipMalloc((void**)&g->facA, (size_t)g->max_fac * D * sizeof(double)));
    RL_HIP(hipMalloc((void**)&g->facW, (size_t)g->max_fac * sizeof(double)));
    RL_HIP(hipMalloc((void**)&g->facQ, (size_t)g->max_fac * sizeof(int)));
    RL_HIP(hipMalloc((void**)&g->kappa, (size_t)max_tops * D * sizeof(double)));
    std::vector<double> ones(D, 1.0);
    if ((rc = upload(&g->ones, ones)) != RL_OK) return rc;

    // intermediates of one chunk: large enough that a launch's tail does not
    // matter, small enough to sit in the 256 MiB Infinity Cache (measured: C2,
    // 1024 vectors 2.17 / 2.31 / 2.50 / 2.24 M MVM/s at 64 / 96 / 192 / 384 MB)
    // (re-measured with two streams at C5, 33 MB per pair: 2.93 / 2.97 / 3.10 / 3.08 ms
    // per 129-vector product at 64 / 96 / 128 / 192 MB -- the working set of both
    // streams then stays inside the Infinity Cache, which serves re-reads at
    // ~7 TB/s against ~5.5 from HBM, tools/mall_probe.py)
    size_t chunk_mb = (size_t)D * L * sizeof(cplx) >= ((size_t)8 << 20) ? 64 : 192;
    if (g->kn.chunk_mb > 0) chunk_mb = (size_t)g->kn.chunk_mb;
    g->chunk_pairs = std::max<size_t>(1, (chunk_mb << 20) / ((size_t)D * L * sizeof(cplx)));
    // Pair-affine order (rl_kernels2.h: affine_tile): every pair on ONE XCD through all
    // three kernels.  Measured (tools/affine_ab.py, profiles/r04/affine_*.txt): the order
    // itself pays -- C2: 19.6 -> 17.9 us at 17 vectors, 347 -> 320 us at 1024 --; chunks
    // small enough that a chunk's intermediates would stay in the XCDs' L2s (RUNLMC_AFFINE_KB)
    // do NOT: the fabric-side reads of T are the same with and without them (PMC: T is not
    // found in L2 by the next kernel) and the many small launches cost 25 %.
    {
        const size_t pairT = (size_t)D * L * sizeof(cplx);
        const bool can = g->v2 && g->rows3 && m1 == 0;
        // (pairs of 0.5 ... 2 MB gain 2-16 %; smaller ones are level, a 6.4 MB pair loses
        // 3-50 %, C5's 32 MB pair a factor of two: profiles/r04/affine_ab_shapes.txt)
        g->affine = can && (g->kn.affine >= 0 ? g->kn.affine != 0
                                              : (pairT >= ((size_t)512 << 10) &&
                                                 pairT <= ((size_t)g->kn.affine_max_kb << 10)));
        if (g->affine && g->kn.affine_kb > 0 && g->kn.chunk_mb <= 0)
            g->chunk_pairs = 8 * std::max<size_t>(1, ((size_t)g->kn.affine_kb << 10) / pairT);
    }
    // single-tile product for short grids
    if (m1 == 0 && L <= 2048 && !g->kn.no_v1p) {
        const size_t lds = ((size_t)L * (D | 1) + L) * sizeof(cplx);
        if (lds <= kLdsHard) {
            g->planL = make_plan(L);
            if ((rc = upload(&g->twL, unity_table(L, 1, L))) != RL_OK) return rc;
            RL_HIP(hipMalloc((void**)&g->spec1, (size_t)max_tops * L * sizeof(double)));
            g->lds1 = lds;
            g->thr1 = (size_t)D * L >= 4096 ? 512 : 256;
            g->v1p = true;
            if (g->kn.v1p_min > 0) g->v1p_min = g->kn.v1p_min;
        }
    }
    *out = guard.release();
    return RL_OK;
}

extern "C" int rl_gridop_destroy(rl_gridop* g) {
    if (!g) return RL_OK;
    (void)hipSetDevice(g->device);
    if (g->child) (void)rl_gridop_destroy(g->child);
    if (g->wide_B) (void)hipFree(g->wide_B);
    if (g->wide_Z) (void)hipFree(g->wide_Z);
    void* ptrs[] = {g->tw1, g->tw2, g->twlo, g->twhi, g->freq1, g->tops, g->spec,
                    g->facA, g->facW, g->facQ, g->kappa, g->ones, g->T,
                    g->T2[0], g->T2[1], g->T2[2], g->twL, g->spec1, g->mixtab, g->lr_beta, g->lr_nu, g->lr_phiJ, g->lr_stat, g->lr_M,
                    g->lr_B, g->lr_eye, g->lr_part, g->lr_zhat, g->lr_scr, g->lr_pw, g->lr_sel, g->lr_Cc, g->lr_Bc, g->lr_Mf, g->lr_spart, g->lr_Cx,
                    g->sf_tops, g->sf_blob, g->sf_blob_top, g->sf_pwp, g->sf_pw, g->sf_kappa, g->sf_facA, g->sf_facAW, g->sf_facJ,
                    g->sf_E, g->sf_Cin, g->sf_next};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    for (int i = 0; i < 3; ++i) {
        if (g->ev_join[i]) (void)hipEventDestroy(g->ev_join[i]);
        if (g->aux[i]) (void)hipStreamDestroy(g->aux[i]);
    }
    delete g;
    return RL_OK;
}

extern "C" int rl_gridop_info(const rl_gridop* g, int* L, int* N1, int* N2, int* colsA,
                              int* rowsB) {
    if (!g) return fail(RL_EINVAL, "gridop is NULL");
    if (L) *L = g->L;
    if (N1) *N1 = g->N1;
    if (N2) *N2 = g->N2;
    if (colsA) *colsA = g->colsA;
    if (rowsB) *rowsB = g->rowsB;
    return RL_OK;
}

extern "C" int rl_gridop_form(const rl_gridop* gc, int* rank, long long* min_elements) {
    if (!gc) return fail(RL_EINVAL, "gridop is NULL");
    if (gc->wide) {
        // (the child's gate counts ITS elements, nvec * D rows of m points: the same number)
        return rl_gridop_form(gc->child, rank, min_elements);
    }
    rl_gridop* g = const_cast<rl_gridop*>(gc);      // (runs the pending verification)
    if (g->Q >= 1) RL_TRY(lr_ensure(g));
    if (rank) *rank = g->lr_ok ? g->lr_r : 0;
    if (min_elements) *min_elements = (long long)g->lr_min;
    return RL_OK;
}

extern "C" int rl_gridop_top_forms(const rl_gridop* gc, int* forms, int* structured) {
    if (!gc) return fail(RL_EINVAL, "gridop is NULL");
    if (gc->wide) {
        // every top runs alone (single-top products of the child): each in its own form
        RL_TRY(rl_gridop_top_forms(gc->child, forms, nullptr));
        if (structured) {
            *structured = 1;
            for (int q = 0; q < gc->child->Q; ++q)
                if (gc->child->top_form[q] == 0) *structured = 0;
        }
        return RL_OK;
    }
    rl_gridop* g = const_cast<rl_gridop*>(gc);      // (runs the pending verification)
    if (g->Q < 1) return fail(RL_EINVAL, "grid operator has no parameters yet");
    RL_TRY(lr_ensure(g));
    if (forms)
        for (int q = 0; q < g->Q; ++q)
            forms[q] = q < (int)g->top_form.size() ? g->top_form[q] : 0;
    if (structured) *structured = (g->lr_ok || g->st_ok) ? 1 : 0;
    return RL_OK;
}

extern "C" int rl_gridop_form_stats(const rl_gridop* gc, int q, double* out4) {
    if (!gc || !out4) return fail(RL_EINVAL, "rl_gridop_form_stats: NULL argument");
    if (gc->wide) return rl_gridop_form_stats(gc->child, q, out4);
    rl_gridop* g = const_cast<rl_gridop*>(gc);      // (runs the pending verification)
    if (g->Q < 1) return fail(RL_EINVAL, "grid operator has no parameters yet");
    if (q < 0 || q >= g->Q) return fail(RL_EINVAL, "rl_gridop_form_stats: no such top row");
    RL_TRY(lr_ensure(g));
    for (int k = 0; k < 4; ++k)
        out4[k] = (size_t)(4 * q + k) < g->lr_vstat.size() ? g->lr_vstat[4 * q + k] : 0.0;
    return RL_OK;
}

extern "C" int rl_gridop_set_form_gate(rl_gridop* g, long long min_elements) {
    if (!g) return fail(RL_EINVAL, "gridop is NULL");
    if (g->wide) return rl_gridop_set_form_gate(g->child, min_elements);
    g->lr_min = min_elements < 0 ? lr_min_elements(g) : (size_t)min_elements;
    return RL_OK;
}

// spectra of tops [0, ntop) -> g->spec, on `stream`
static int build_spectra(rl_gridop* g, int ntop, hipStream_t stream) {
    const int npairs = (ntop + 1) / 2;
    // the spectrum pass uses the workspace as [npairs][1][L]
    const size_t need = ((size_t)npairs + g->D - 1) / g->D;
    RL_TRY(ensure_workspace(g, std::max<size_t>(need, 1)));
    dim3 gridA(g->N2 / g->colsA, 1, npairs);
    Gather no_gather;
    no_gather.indptr = nullptr;
    no_gather.lo = nullptr;
    RL_LAUNCH(k_cols_fwd, gridA, dim3(RL_THREADS), lds_cols(g), stream, g->tops, ntop, 1, g->geo,
              1, g->T, g->N1, g->N2, g->colsA, g->plan1, g->tw1, g->freq1, g->twl, no_gather,
              (int*)nullptr);
    dim3 gridS(g->N1 / g->rowsS, npairs);
    RL_LAUNCH(k_rows_spec, gridS, dim3(RL_THREADS), lds_rows(g->N2, g->rowsS), stream, g->T,
              g->spec, ntop, g->N1, g->N2, g->rowsS, g->plan2, g->tw2);
    if (g->v1p) {
        MixParams none{0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        RL_TRY(launch1p(g, 1, (unsigned)npairs, stream, g->tops, nullptr, ntop, 1, none,
                        g->spec1));
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// A setter validates EVERYTHING and builds its factors on the host before it
// touches the handle (set_check), then commits: tops + spectra (set_common) and
// factors + mix tables (set_factors).  A device failure half way leaves the
// handle without parameters (Q = 0), so that later products fail loudly
// instead of mixing new spectra with old factors.
static int set_check(rl_gridop* g, int Q, const double* tops) {
    if (!g) return fail(RL_EINVAL, "gridop is NULL");
    if (Q < 1 || Q > g->max_tops)
        return fail(RL_EINVAL, "rl_gridop_set: Q outside [1, max_tops]");
    if (!tops) return fail(RL_EINVAL, "rl_gridop_set: tops is NULL");
    return RL_OK;
}
static int set_commit(rl_gridop* g, int Q, const double* tops, const std::vector<double>& A,
                      const std::vector<double>& W, const std::vector<int>& Qi,
                      const std::vector<double>& kap);

static int set_common(rl_gridop* g, int Q, const double* tops) {
    RL_TRY(set_check(g, Q, tops));
    RL_HIP(hipSetDevice(g->device));
    RL_HIP(hipMemcpy(g->tops, tops, (size_t)Q * g->m * sizeof(double), hipMemcpyHostToDevice));
    RL_TRY(build_spectra(g, Q, nullptr));
    g->Q = Q;
    return RL_OK;
}

static int set_factors(rl_gridop* g, const std::vector<double>& A, const std::vector<double>& W,
                       const std::vector<int>& Qi, const std::vector<double>& kap) {
    const int nfac = (int)W.size();
    if (nfac > g->max_fac) return fail(RL_ELIMIT, "rl_gridop_set: total rank exceeds max_tops*D");
    if (nfac) {
        RL_HIP(hipMemcpy(g->facA, A.data(), A.size() * sizeof(double), hipMemcpyHostToDevice));
        RL_HIP(hipMemcpy(g->facW, W.data(), W.size() * sizeof(double), hipMemcpyHostToDevice));
        RL_HIP(hipMemcpy(g->facQ, Qi.data(), Qi.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    RL_HIP(hipMemcpy(g->kappa, kap.data(), kap.size() * sizeof(double), hipMemcpyHostToDevice));
    g->nfac = nfac;
    // mix tables of the third-generation row kernel (dc: D rows, gs: nfac rows).
    // (Measured against a mix that reads only the Q spectra and forms dc / gs in
    // registers -- a third of the table bytes, 100 more multiply-adds and 20 more
    // scalar loads per point: 3.79 vs 2.96 ms per C5 product.  Removed.)
    g->mixtab_ok = false;
    if (g->rows3 && nfac <= RL_MIXF) {
        const size_t rows = (size_t)g->D + nfac;
        if (rows > g->mixtab_rows) {
            if (g->mixtab) RL_HIP(hipFree(g->mixtab));
            g->mixtab = nullptr;
            g->mixtab_rows = 0;
            RL_HIP(hipMalloc((void**)&g->mixtab, rows * (size_t)g->L * sizeof(double)));
            g->mixtab_rows = rows;
        }
        MixParams mp{g->Q, g->nfac, g->spec, g->facA, g->facW, g->facQ, g->kappa, nullptr, nullptr};
        RL_LAUNCH(k_mix_tables, dim3((g->L + 255) / 256, (unsigned)rows), dim3(256), 0,
                  (hipStream_t) nullptr, mp, g->D, g->L, g->mixtab,
                  g->mixtab + (size_t)g->D * g->L);
        RL_HIP(hipGetLastError());
        g->mixtab_ok = true;
    }
    return RL_OK;
}

static int set_commit(rl_gridop* g, int Q, const double* tops, const std::vector<double>& A,
                      const std::vector<double>& W, const std::vector<int>& Qi,
                      const std::vector<double>& kap) {
    if ((int)W.size() > g->max_fac)
        return fail(RL_ELIMIT, "rl_gridop_set: total rank exceeds max_tops*D");
    g->lr_ok = false;
    g->lr_Mf_ok = false;
    g->lr_dirty = false;
    ++g->param_ver;
    int rc = set_common(g, Q, tops);
    if (rc == RL_OK) rc = set_factors(g, A, W, Qi, kap);
    g->st_ok = false;
    g->top_form.assign(Q, 0);
    if (rc == RL_OK && (g->lr_try || g->sf_try)) {
        g->h_tops.assign(tops, tops + (size_t)Q * g->m);
        g->lr_A = A;
        g->lr_W = W;
        g->lr_Qi = Qi;
        g->lr_kap = kap;
        g->lr_dirty = true;
    }
    if (rc != RL_OK) g->Q = 0;          // no half-updated operator
    return rc;
}

// wide operator (D > RL_MAX_D): the child takes the top rows alone, the parent keeps the
// dense couplings B_q.  A failure leaves the handle without parameters.
static int wide_set(rl_gridop* g, int Q, const double* tops, const std::vector<double>& B) {
    RL_HIP(hipSetDevice(g->device));
    g->Q = 0;
    std::vector<int> zero(Q, 0);
    std::vector<double> ones(Q, 1.0);
    RL_TRY(rl_gridop_set_lmc(g->child, Q, tops, zero.data(), nullptr, ones.data()));
    RL_HIP(hipMemcpy(g->wide_B, B.data(), B.size() * sizeof(double), hipMemcpyHostToDevice));
    g->Q = Q;
    return RL_OK;
}

extern "C" int rl_gridop_set_lmc(rl_gridop* g, int Q, const double* tops, const int* ranks,
                                 const double* coreg_vecs, const double* coreg_diags) {
    RL_TRY(set_check(g, Q, tops));
    if (!ranks || !coreg_diags) return fail(RL_EINVAL, "rl_gridop_set_lmc: NULL argument");
    const int D = g->D;
    if (g->wide) {
        std::vector<double> B((size_t)Q * D * D, 0.0);
        size_t wrow = 0;
        for (int q = 0; q < Q; ++q) {
            if (ranks[q] < 0) return fail(RL_EINVAL, "rl_gridop_set_lmc: negative rank");
            if (ranks[q] > 0 && !coreg_vecs)
                return fail(RL_EINVAL, "rl_gridop_set_lmc: coreg_vecs is NULL");
            for (int r = 0; r < ranks[q]; ++r, ++wrow)
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j)
                        B[((size_t)q * D + i) * D + j] +=
                            coreg_vecs[wrow * D + i] * coreg_vecs[wrow * D + j];
            for (int i = 0; i < D; ++i)
                B[((size_t)q * D + i) * D + i] += coreg_diags[(size_t)q * D + i];
        }
        return wide_set(g, Q, tops, B);
    }
    std::vector<double> A, W, kap(coreg_diags, coreg_diags + (size_t)Q * D);
    std::vector<int> Qi;
    size_t row = 0;
    long total_rank = 0;
    for (int q = 0; q < Q; ++q) {
        if (ranks[q] < 0) return fail(RL_EINVAL, "rl_gridop_set_lmc: negative rank");
        total_rank += ranks[q];
    }
    if (total_rank > g->max_fac) {
        // more factors than the handle was sized for (ranks above D are redundant
        // but legal in the reference): fold into dense B_q = A_q^T A_q + diag(kappa_q),
        // which re-factors to at most D factors per kernel
        if (!coreg_vecs) return fail(RL_EINVAL, "rl_gridop_set_lmc: coreg_vecs is NULL");
        std::vector<double> B((size_t)Q * D * D, 0.0);
        for (int q = 0; q < Q; ++q) {
            for (int r = 0; r < ranks[q]; ++r, ++row)
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j)
                        B[((size_t)q * D + i) * D + j] +=
                            coreg_vecs[row * D + i] * coreg_vecs[row * D + j];
            for (int i = 0; i < D; ++i)
                B[((size_t)q * D + i) * D + i] += coreg_diags[(size_t)q * D + i];
        }
        return rl_gridop_set_dense(g, Q, tops, B.data());
    }
    for (int q = 0; q < Q; ++q) {
        if (ranks[q] > 0 && !coreg_vecs)
            return fail(RL_EINVAL, "rl_gridop_set_lmc: coreg_vecs is NULL");
        for (int r = 0; r < ranks[q]; ++r, ++row) {
            A.insert(A.end(), coreg_vecs + row * D, coreg_vecs + (row + 1) * D);
            W.push_back(1.0);
            Qi.push_back(q);
        }
    }
    return set_commit(g, Q, tops, A, W, Qi, kap);
}

// cyclic Jacobi eigen-decomposition of a small symmetric matrix (row-major,
// overwritten); V's COLUMNS are eigenvectors
static void jacobi_eig(std::vector<double>& a, int n, std::vector<double>& w,
                       std::vector<double>& V) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                (i == j ? diag : off) += a[(size_t)i * n + j] * a[(size_t)i * n + j];
        if (off <= 1e-34 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double theta = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) /
                                 (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                    a[(size_t)k * n + p] = c * akp - s * akq;
                    a[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                    a[(size_t)p * n + k] = c * apk - s * aqk;
                    a[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = a[(size_t)i * n + i];
}

extern "C" int rl_gridop_set_dense(rl_gridop* g, int Q, const double* tops, const double* B) {
    RL_TRY(set_check(g, Q, tops));
    if (!B) return fail(RL_EINVAL, "rl_gridop_set_dense: B is NULL");
    const int D = g->D;
    if (g->wide) {
        for (int q = 0; q < Q; ++q)
            for (int i = 0; i < D; ++i)
                for (int j = 0; j < i; ++j) {
                    const double bij = B[((size_t)q * D + i) * D + j];
                    const double bji = B[((size_t)q * D + j) * D + i];
                    if (std::fabs(bij - bji) > 1e-12 * std::max(std::fabs(bij), std::fabs(bji)))
                        return fail(RL_EINVAL, "rl_gridop_set_dense: B_q is not symmetric");
                }
        return wide_set(g, Q, tops, std::vector<double>(B, B + (size_t)Q * D * D));
    }
    std::vector<double> A, W, kap((size_t)Q * D, 0.0);
    std::vector<int> Qi;
    for (int q = 0; q < Q; ++q) {
        std::vector<double> sym((size_t)D * D), w, V;
        double asym = 0.0, scale = 0.0;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                const double bij = B[((size_t)q * D + i) * D + j];
                const double bji = B[((size_t)q * D + j) * D + i];
                sym[(size_t)i * D + j] = 0.5 * (bij + bji);
                asym = std::max(asym, std::fabs(bij - bji));
                scale = std::max(scale, std::fabs(bij));
            }
        if (asym > 1e-12 * std::max(scale, 1e-300))
            return fail(RL_EINVAL, "rl_gridop_set_dense: B_q is not symmetric");
        jacobi_eig(sym, D, w, V);
        for (int f = 0; f < D; ++f) {
            if (w[f] == 0.0) continue;
            for (int b = 0; b < D; ++b) A.push_back(V[(size_t)b * D + f]);
            W.push_back(w[f]);
            Qi.push_back(q);
        }
    }
    return set_commit(g, Q, tops, A, W, Qi, kap);
}

template <int D>
static void launch_rows_mix(rl_gridop* g, size_t pairs, hipStream_t stream, const MixParams& mp) {
    dim3 gridB(g->N1 / g->rowsB, (unsigned)pairs);
    RL_LAUNCH(k_rows_mix<D>, gridB, dim3(RL_THREADS), lds_rows(g->N2, g->rowsB * D), stream,
              g->T, g->N1, g->N2, g->rowsB, g->plan2, g->tw2, g->freq1, g->twl, mp);
}

// ---- second-generation kernels: dispatch ------------------------------------
// 0 means "divide normally" (d == 1 has no 32-bit magic)
static unsigned div_magic(unsigned d) {
    return d <= 1 ? 0u : (unsigned)((1ull << 32) / d) + 1u;
}

// tile widths for a launch over `pairs` vector pairs: as large as LDS allows
// while the launch still has enough workgroups to cover the 256 CUs
static void choose_tiles(const rl_gridop* g, size_t pairs, Tile2* tp) {
    tp->N1 = g->N1;
    tp->N2 = g->N2;
    // column tiles: 16 columns = 256-byte rows of T; 32 when there is plenty of work
    int C = std::min(16, g->N2);
    if (g->N2 >= 32 && (size_t)g->N1 * 32 * sizeof(cplx) <= kLdsSoft &&
        (size_t)(g->N2 / 32) * g->D * pairs >= 1024)
        C = 32;
    while ((size_t)g->N1 * C * sizeof(cplx) > kLdsSoft && C > 8) C /= 2;
    while ((size_t)g->N1 * C * sizeof(cplx) > kLdsHard && C > 1) C /= 2;
    tp->C = C;
    tp->logC = ilog2(C);
    // row tiles: enough side-by-side transforms that the first pass gives every
    // one of the RL_THREADS threads a butterfly, more while work is plentiful
    const int sub = g->N2 / g->plan2.radix[0];
    auto lds = [&](int R) { return (size_t)g->N2 * ((R * g->D) | 1) * sizeof(cplx); };
    int R = 1;
    while (g->N1 % (R * 2) == 0 && R * g->D * sub < RL_THREADS && lds(2 * R) <= kLdsSoft)
        R *= 2;
    while (g->N1 % (R * 2) == 0 && lds(2 * R) <= 40 * 1024 &&
           (size_t)(g->N1 / (2 * R)) * pairs >= 2048)
        R *= 2;
    tp->R = R;
    tp->colsMagic = div_magic((unsigned)(R * g->D));
    // 512 threads when LDS leaves room for only one or two workgroups per CU
    // (big tiles: otherwise a CU would hold 4-8 waves) and the first pass has
    // that many butterflies to hand out; measured on C5: 4.31 -> 3.89 ms
    const int thr_max = RL_THREADS2;
    const size_t big = 48 * 1024;
    tp->thrR = (lds(R) > big && R * g->D * sub >= 512 && thr_max >= 512) ? 512 : RL_THREADS;
    tp->thrC = ((size_t)g->N1 * C * sizeof(cplx) > big &&
                (g->N1 / g->plan1.radix[0]) * C >= 512 && thr_max >= 512) ? 512 : RL_THREADS;
    // Launches that do not fill the chip (the probe batches of a solve: 9 pairs at
    // C2) are chains of instruction / LDS latencies with one wavefront per SIMD:
    // smaller tiles spread the same work over more, narrower workgroups
    // (measured at C2, 17 vectors: 26.1 -> 23.7 us per product; 64 vectors are
    // already past the point where it helps)
    {
        bool shrunk = false;
        while (R > 1 && (size_t)(g->N1 / R) * pairs < 512) {
            R /= 2;
            shrunk = true;
        }
        if (shrunk) {
            tp->R = R;
            tp->colsMagic = div_magic((unsigned)(R * g->D));
            tp->thrR = std::max(64, std::min(RL_THREADS, ((R * g->N2 + 63) / 64) * 64));
        }
        if (C > 8 && (size_t)(g->N2 / C) * g->D * pairs < 512) {
            C = 8;
            tp->C = C;
            tp->logC = ilog2(C);
            const int items = (g->N1 / g->plan1.radix[0]) * C;
            tp->thrC = std::max(64, std::min(RL_THREADS, ((items + 63) / 64) * 64));
        }
    }
    if (g->rows3) {
        // unpadded tile of R rows x D outputs; enough rows that pass A hands every
        // thread of a 256-thread workgroup a butterfly, at most 80 KiB (two
        // workgroups per CU); more rows only while the launch stays large
        const int sa = g->N2 / g->plan2.radix[0], nbf = g->N2 / g->plan2.radix[1];
        auto lds3 = [&](int r) { return (size_t)g->N2 * r * g->D * sizeof(cplx); };
        R = 1;
        while (g->N1 % (R * 2) == 0 && R * g->D * sa < RL_THREADS && lds3(2 * R) <= 80 * 1024)
            R *= 2;
        while (g->N1 % (R * 2) == 0 && lds3(2 * R) <= 40 * 1024 &&
               (size_t)(g->N1 / (2 * R)) * pairs >= 2048)
            R *= 2;
        while (R > 1 && (size_t)(g->N1 / R) * pairs < 512) R /= 2;
        tp->R = R;
        tp->colsMagic = div_magic((unsigned)(R * g->D));
        // workgroup size: the multiple of 64 that needs the fewest rounds over the
        // three phases, the smallest such; when LDS admits several workgroups per
        // CU, small enough that they also fit the waves the kernel's registers
        // allow (C5: 512 threads -- the 2 x 512 split re/im mix items in ONE round,
        // the 320 butterflies of the radix-16 passes on five of the eight waves;
        // measured 2.81 vs 2.95 ms per 129-vector product against 320 threads)
        const int items[3] = {R * g->D * sa, R * g->D * nbf, R * g->N2};
        const int weight[3] = {4, 4, 1};      // passes A and B run twice; a mix item is light
        // (the kernel is built for 128 VGPRs while D <= 12: 16 waves per CU)
        const int wgs = (int)std::min<size_t>(8, (160 * 1024) / lds3(R));
        const int waves = g->D <= 12 ? 16 : 8;
        const int tmax = wgs >= 2 ? std::max(256, (waves / wgs) * 64) : RL_THREADS3;
        long bestCost = -1;
        int bestT = 64;
        for (int t = 64; t <= tmax; t += 64) {
            long cost = 0;
            for (int k = 0; k < 3; ++k) cost += (long)weight[k] * ((items[k] + t - 1) / t);
            if (bestCost < 0 || cost < bestCost) { bestCost = cost; bestT = t; }
        }
        tp->thrR = bestT;
    }
}

template <int RA, int RB>
static void launch2_cols_fwd(rl_gridop* g, const Tile2& tp, size_t pairs, hipStream_t st,
                             const double* X, int nv, int D, int mode, const Gather& gs) {
    dim3 grid(g->N2 / tp.C, D, (unsigned)pairs);
    if (tp.aff > 0) grid = dim3((unsigned)(8 * ((pairs + 7) / 8) * (g->N2 / tp.C) * D));
    if (gs.indptr != nullptr)
        RL_LAUNCH((k2_cols_fwd<RA, RB, true>), grid, dim3(tp.thrC),
                  (size_t)g->N1 * tp.C * sizeof(cplx), st, X, nv, D, g->geo, mode, g->Tcur, tp,
                  g->plan1, g->tw1, g->freq1, g->twl, gs);
    else
        RL_LAUNCH((k2_cols_fwd<RA, RB, false>), grid, dim3(tp.thrC),
                  (size_t)g->N1 * tp.C * sizeof(cplx), st, X, nv, D, g->geo, mode, g->Tcur, tp,
                  g->plan1, g->tw1, g->freq1, g->twl, gs);
}
template <int RA, int RB>
static void launch2_cols_inv(rl_gridop* g, const Tile2& tp, size_t pairs, hipStream_t st,
                             double* Y, int nv) {
    const int colsNeeded = g->geo.m1 ? g->geo.m2 : std::min(g->m, g->N2);
    dim3 grid((colsNeeded + tp.C - 1) / tp.C, g->D, (unsigned)pairs);
    if (tp.aff > 0)
        grid = dim3((unsigned)(8 * ((pairs + 7) / 8) * ((colsNeeded + tp.C - 1) / tp.C) * g->D));
    RL_LAUNCH((k2_cols_inv<RA, RB>), grid, dim3(tp.thrC), (size_t)g->N1 * tp.C * sizeof(cplx),
              st, g->Tcur, Y, nv, g->D, g->geo, tp, g->plan1, g->tw1);
}
template <int D, int RA, int RB>
static void launch2_rows(rl_gridop* g, const Tile2& tp, size_t pairs, hipStream_t st,
                         const MixParams& mp, int* bump) {
    dim3 grid(g->N1 / tp.R, (unsigned)pairs);
    const size_t lds = (size_t)g->N2 * ((tp.R * D) | 1) * sizeof(cplx);
    RL_LAUNCH((k2_rows_mix<D, RA, RB>), grid, dim3(tp.thrR), lds, st, g->Tcur, tp, g->plan2,
              g->tw2, g->freq1, g->twl, mp, bump);
}
// third-generation row kernel: tile = R rows x D outputs, unpadded
template <int D, int RA, int RB>
static void launch3_rows(rl_gridop* g, const Tile2& tp, size_t pairs, hipStream_t st,
                         const MixParams& mp, int* bump) {
    dim3 grid(g->N1 / tp.R, (unsigned)pairs);
    if (tp.aff > 0) grid = dim3((unsigned)(8 * ((pairs + 7) / 8) * (g->N1 / tp.R)));
    const size_t lds = (size_t)g->N2 * tp.R * D * sizeof(cplx);
#if !defined(RL_EMU)
    {
        static bool told = false;
        if (!told && getenv("RUNLMC_TRACE") != nullptr) {
            told = true;
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &nb, (const void*)k3_rows_mix<D, RA, RB>, tp.thrR, lds);
            fprintf(stderr, "[runlmc] k3_rows_mix<%d,%d,%d>: R=%d threads=%d lds=%zu grid=%u x %u, "
                    "occupancy query: %d workgroups per CU\n", D, RA, RB, tp.R, tp.thrR, lds,
                    grid.x, grid.y, nb);
        }
    }
#endif
    RL_LAUNCH((k3_rows_mix<D, RA, RB>), grid, dim3(tp.thrR), lds, st, g->Tcur, tp, g->tw2,
              g->freq1, g->twl, mp, bump);
}
template <int D>
static void launch2_rows_code(rl_gridop* g, const Tile2& tp, size_t pairs, hipStream_t st,
                              const MixParams& mp, int* bump) {
    if (g->rows3) {
        trace_once("row kernel: k3_rows_mix");
        switch (g->code2) {
            case 808: launch3_rows<D, 8, 8>(g, tp, pairs, st, mp, bump); break;
            case 1608: launch3_rows<D, 16, 8>(g, tp, pairs, st, mp, bump); break;
            default: launch3_rows<D, 16, 16>(g, tp, pairs, st, mp, bump); break;
        }
        return;
    }
    switch (g->code2) {
        case 808: launch2_rows<D, 8, 8>(g, tp, pairs, st, mp, bump); break;
        case 816: launch2_rows<D, 8, 16>(g, tp, pairs, st, mp, bump); break;
        default: launch2_rows<D, 16, 16>(g, tp, pairs, st, mp, bump); break;
    }
}

int mvm_chunk_v2(rl_gridop* g, const MixParams& mp, const double* Xc, double* Yc, int nv, size_t pairs,
                 hipStream_t st, const Gather* gather, int* bump, cplx* Tbuf) {
    g->Tcur = Tbuf ? Tbuf : g->T;      // the launches below read it
    Tile2 tp;
    choose_tiles(g, pairs, &tp);
    tp.aff = g->affine ? (int)pairs : 0;
    if (tp.aff > 0) trace_once("transform kernels: pair-affine order (one XCD per pair)");
    Gather gs;
    if (gather != nullptr) gs = *gather; else gs.indptr = nullptr;
    switch (g->code1) {
        case 808: launch2_cols_fwd<8, 8>(g, tp, pairs, st, Xc, nv, g->D, 0, gs); break;
        case 816: launch2_cols_fwd<8, 16>(g, tp, pairs, st, Xc, nv, g->D, 0, gs); break;
        case 308: launch2_cols_fwd<3, 8>(g, tp, pairs, st, Xc, nv, g->D, 0, gs); break;
        case 316: launch2_cols_fwd<3, 16>(g, tp, pairs, st, Xc, nv, g->D, 0, gs); break;
        case 508: launch2_cols_fwd<5, 8>(g, tp, pairs, st, Xc, nv, g->D, 0, gs); break;
        case 516: launch2_cols_fwd<5, 16>(g, tp, pairs, st, Xc, nv, g->D, 0, gs); break;
        default: launch2_cols_fwd<16, 16>(g, tp, pairs, st, Xc, nv, g->D, 0, gs); break;
    }
    switch (g->D) {
#define RL_CASE(d) case d: launch2_rows_code<d>(g, tp, pairs, st, mp, bump); break;
        RL_CASE(1) RL_CASE(2) RL_CASE(3) RL_CASE(4) RL_CASE(5) RL_CASE(6) RL_CASE(7)
        RL_CASE(8) RL_CASE(9) RL_CASE(10) RL_CASE(11) RL_CASE(12) RL_CASE(13)
        RL_CASE(14) RL_CASE(15) RL_CASE(16)
#undef RL_CASE
        default: return fail(RL_ELIMIT, "unsupported D");
    }
    switch (g->code1) {
        case 808: launch2_cols_inv<8, 8>(g, tp, pairs, st, Yc, nv); break;
        case 816: launch2_cols_inv<8, 16>(g, tp, pairs, st, Yc, nv); break;
        case 308: launch2_cols_inv<3, 8>(g, tp, pairs, st, Yc, nv); break;
        case 316: launch2_cols_inv<3, 16>(g, tp, pairs, st, Yc, nv); break;
        case 508: launch2_cols_inv<5, 8>(g, tp, pairs, st, Yc, nv); break;
        case 516: launch2_cols_inv<5, 16>(g, tp, pairs, st, Yc, nv); break;
        default: launch2_cols_inv<16, 16>(g, tp, pairs, st, Yc, nv); break;
    }
    return RL_OK;
}

// one chunk through the first-generation kernels (every pass in LDS)
int mvm_chunk_v1(rl_gridop* g, const MixParams& mp, const double* Xc, double* Yc, int nv, size_t pairs,
                 hipStream_t stream, const Gather* gather, int* bump) {
    Gather gs;
    if (gather != nullptr) {
        gs = *gather;
    } else {
        gs.indptr = nullptr;
        gs.lo = nullptr;
    }
    const int colsNeeded = g->geo.m1 ? g->geo.m2 : std::min(g->m, g->N2);
    const unsigned tilesInv = (colsNeeded + g->colsA - 1) / g->colsA;
    dim3 gridA(g->N2 / g->colsA, g->D, (unsigned)pairs);
    RL_LAUNCH(k_cols_fwd, gridA, dim3(RL_THREADS), lds_cols(g), stream, Xc, nv, g->D, g->geo, 0,
              g->T, g->N1, g->N2, g->colsA, g->plan1, g->tw1, g->freq1, g->twl, gs, bump);
    switch (g->D) {
#define RL_CASE(d) case d: launch_rows_mix<d>(g, pairs, stream, mp); break;
        RL_CASE(1) RL_CASE(2) RL_CASE(3) RL_CASE(4) RL_CASE(5) RL_CASE(6) RL_CASE(7)
        RL_CASE(8) RL_CASE(9) RL_CASE(10) RL_CASE(11) RL_CASE(12) RL_CASE(13)
        RL_CASE(14) RL_CASE(15) RL_CASE(16)
#undef RL_CASE
        default: return fail(RL_ELIMIT, "unsupported D");
    }
    dim3 gridI(tilesInv, g->D, (unsigned)pairs);
    RL_LAUNCH(k_cols_inv, gridI, dim3(RL_THREADS), lds_cols(g), stream, g->T, Yc, nv, g->D,
              g->geo, g->N1, g->N2, g->colsA, g->plan1, g->tw1);
    return RL_OK;
}

// second chunk of intermediates + side stream of the two-stream batched product
// streams a chunked product runs on (measured at C5: three / four equal or worse)
static int product_streams() { return 2; }
static int prepare_two_streams(rl_gridop* g, size_t chunk) {
    const int want = product_streams() - 1;
    if (g->T2_pairs < chunk || g->nside < want) {
        for (int i = 0; i < 3; ++i) {
            if (g->T2[i]) RL_HIP(hipFree(g->T2[i]));
            g->T2[i] = nullptr;
        }
        g->T2_pairs = 0;
        for (int i = 0; i < want; ++i)
            RL_HIP(hipMalloc((void**)&g->T2[i], chunk * g->D * (size_t)g->L * sizeof(cplx)));
        g->T2_pairs = chunk;
    }
    if (!g->ev_fork) RL_HIP(hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < want; ++i)
        if (!g->aux[i]) {
            RL_HIP(hipStreamCreateWithFlags(&g->aux[i], hipStreamNonBlocking));
            RL_HIP(hipEventCreateWithFlags(&g->ev_join[i], hipEventDisableTiming));
        }
    g->nside = want;
    return RL_OK;
}
static bool wants_two_streams(const rl_gridop* g) {
    return g->kn.two_streams >= 0 ? g->kn.two_streams != 0
                                  : ((g->affine && g->kn.affine_kb > 0) ||
                                     (size_t)g->D * g->L * sizeof(cplx) >= ((size_t)8 << 20));
}


// ---------------------------------------------------------------------------
// polynomial-subspace form of the grid product (rl_lowrank.h)
// ---------------------------------------------------------------------------
// Orthonormal polynomials of degree < RL_LR_RMAX on the m equispaced points of
// [-1, 1] by the three-term (Stieltjes / Lanczos) recurrence
//     b_{j+1} p_{j+1} = s p_j - b_j p_{j-1},   p_0 = 1 / sqrt(m)
// in long double (the points are symmetric, so the diagonal coefficients vanish).
// Measured: |Phi^T Phi - I| <= 1e-14 for 48 functions at m = 2048 ... 100 004.
// Rows beyond RL_LR_RMAX + RL_LR_EXTRA of the function-major copy are zero
// padding (the set-time products read whole vectors of D blocks).
#define RL_LR_EXTRA 4          // omitted polynomials the verification checks beyond the rank
static int lr_make_basis(rl_gridop* g) {
    const int m = g->m, R = RL_LR_RMAX;
    std::vector<double> phiJ((size_t)(R + 16) * m, 0.0), beta(R), nu(R);
    std::vector<long double> prev(m, 0.0L), cur(m), nxt(m);
    const long double p0 = 1.0L / sqrtl((long double)m);
    for (int n = 0; n < m; ++n) cur[n] = p0;
    long double bj = 0.0L, nuj = p0;
    // (RL_LR_EXTRA functions beyond the largest rank: the verification looks at what
    // the operator does to the first polynomials a rank omits)
    for (int j = 0; j < R + RL_LR_EXTRA; ++j) {
        for (int n = 0; n < m; ++n) phiJ[(size_t)j * m + n] = (double)cur[n];
        // the kernels run the monic recurrence q_{j+1} = s q_j - b_j^2 q_{j-1},
        // Phi_j = nu_j q_j with nu_{j+1} = nu_j / b_{j+1}
        if (j < R) {
            beta[j] = (double)(bj * bj);
            nu[j] = (double)nuj;
        }
        long double nrm = 0.0L;
        for (int n = 0; n < m; ++n) {
            const long double sn = m > 1 ? -1.0L + 2.0L * n / (m - 1) : 0.0L;
            nxt[n] = sn * cur[n] - bj * prev[n];
            nrm += nxt[n] * nxt[n];
        }
        bj = sqrtl(nrm);
        if (!(bj > 0.0L)) return fail(RL_EINVAL, "polynomial basis degenerate");
        nuj /= bj;
        for (int n = 0; n < m; ++n) {
            prev[n] = cur[n];
            cur[n] = nxt[n] / bj;
        }
    }
    RL_TRY(upload(&g->lr_phiJ, phiJ));
    RL_TRY(upload(&g->lr_beta, beta));
    RL_TRY(upload(&g->lr_nu, nu));
    g->lr_hnu = nu;
    return RL_OK;
}

// Projection chunks of 64 * steps slots (a grid point of the first half and its
// mirror, rl_lowrank.h), steps a multiple of RL_LR_T: the
// longest chunk (the 64-lane reduction at its end costs about 8 lane-steps) that
// still leaves whole rounds of resident workgroups (2 per CU) well filled.
static int lr_steps(const rl_gridop* g, int nrows, int R) {
    // (a short grid is ONE chunk of as few lane-steps as hold its slots; a multiple of the
    // request ring's length)
    const int slots = (g->m + 1) / 2;
    if (slots <= 64 * RL_LR_T) return std::max(4, ((slots + 63) / 64 + 3) / 4 * 4);
    const int rowblocks = (nrows + RL_LR_ROWS(R) - 1) / RL_LR_ROWS(R);
    const double resident = 2.0 * RL_LR_CUS;
    int best = RL_LR_T;
    double best_cost = 1e300;
    for (int steps = RL_LR_T; steps <= 8 * RL_LR_T; steps *= 2) {
        const int chunks = ((g->m + 1) / 2 + 64 * steps - 1) / (64 * steps);
        const double rounds = std::ceil((double)chunks * rowblocks / resident);
        const double cost = rounds * (steps + 8);
        if (cost < best_cost * 0.97) {          // (longer only for a real gain)
            best_cost = cost;
            best = steps;
        }
        if (chunks == 1) break;
    }
    return best;
}
// partial sums: sized for the shortest chunks
static int lr_nchunks_max(const rl_gridop* g) {
    return ((g->m + 1) / 2 + 64 * RL_LR_T - 1) / (64 * RL_LR_T);
}
static size_t lr_part_need(const rl_gridop* g, int nvec) {
    return (size_t)lr_nchunks_max(g) * nvec * g->D * RL_LR_RMAX;
}
int lr_reserve(rl_gridop* g, int nvec) {
    const size_t rows = (size_t)nvec * g->D;
    const size_t need_part = lr_part_need(g, nvec), need_z = rows * RL_LR_RMAX;
    if (g->lr_part_cap < need_part) {
        if (g->lr_part) RL_HIP(hipFree(g->lr_part));
        g->lr_part = nullptr;
        g->lr_part_cap = 0;
        RL_HIP(hipMalloc((void**)&g->lr_part, need_part * sizeof(double)));
        g->lr_part_cap = need_part;
    }
    if (g->lr_zhat_cap < need_z) {
        if (g->lr_zhat) RL_HIP(hipFree(g->lr_zhat));
        g->lr_zhat = nullptr;
        g->lr_zhat_cap = 0;
        RL_HIP(hipMalloc((void**)&g->lr_zhat, need_z * sizeof(double)));
        g->lr_zhat_cap = need_z;
    }
    return RL_OK;
}

// default gate: batches below this many elements (k * D * m) stay on the transform
// path.  Measured (tools/form_crossover.py): at the C2 grid the two forms meet at 32
// vectors (0.64 M elements: 28 us each; 17 vectors 28 vs 19 us, 64 vectors 30 vs
// 33 us, 256 vectors 39 vs 83 us), at the C5 grid a two-vector batch (2 M elements)
// already takes 37 against 60 us.  RUNLMC_LR_MIN / rl_gridop_set_form_gate override.
static size_t lr_min_elements(const rl_gridop* g) {
    return g->kn.lr_min >= 0 ? (size_t)g->kn.lr_min : (size_t)1 << 20;
}

// launches the projection, returns the number of chunks (partial sums per row)
template <int R>
static int lr_project(rl_gridop* g, const double* X, int nrows, hipStream_t st,
                      double* part = nullptr) {
    const int steps = lr_steps(g, nrows, R);
    const int chunks = ((g->m + 1) / 2 + 64 * steps - 1) / (64 * steps);
    RL_LAUNCH((k_lr_project<R>), dim3(chunks, (nrows + RL_LR_ROWS(R) - 1) / RL_LR_ROWS(R)),
              dim3(64 * RL_LR_WAVES), (size_t)RL_LR_WAVES * R * 65 * sizeof(double), st, X, nrows,
              g->m, (const double*)g->lr_beta, steps, part != nullptr ? part : g->lr_part);
    return chunks;
}
template <int R>
static void lr_mix(rl_gridop* g, const double* part, int chunks, int nvec, int Q, const double* Cq,
                   const double* Bq, double* zhat, hipStream_t st) {
    int split3 = 0;
    const size_t mix_lds = lr_mix_lds(g->D, R, Q, &split3);
    RL_LAUNCH(k_lr_mix, dim3(nvec), dim3(RL_LR_MIXT), mix_lds, st, part, chunks, nvec, g->D, R, Q,
              Cq, Bq, (const double*)g->lr_nu, zhat, (const int*)nullptr, split3);
}
template <int R>
static void lr_expand(rl_gridop* g, const double* zhat, int nrows, double* Y, int accumulate,
                      hipStream_t st) {
    // rows per expansion workgroup: the basis values of a slot are generated once
    // per workgroup (48 instructions against 14 per row and slot).  16 rows when
    // that makes at least two resident rounds of workgroups (measured at C5, 1290
    // rows, row blocks numbered fastest: 4 / 8 / 16 / 32 rows: 200-211 us, flat),
    // otherwise as many row blocks as make one round.  Resident workgroups per
    // CU: 8 (58 VGPRs).
    const int per_cu = 8;
    const int nbx = ((g->m + 1) / 2 + 255) / 256;        // slots: a point and its mirror
    int rpb = 16;
    if ((size_t)nbx * ((nrows + 15) / 16) < (size_t)2 * per_cu * RL_LR_CUS) {
        const int nby = std::max(1, std::min(nrows, per_cu * RL_LR_CUS / nbx));
        rpb = (nrows + nby - 1) / nby;
    }
    if (accumulate)
        RL_LAUNCH((k_lr_expand<R, true>), dim3(nbx, (nrows + rpb - 1) / rpb), dim3(256), 0,
                  st, zhat, nrows, g->m, (const double*)g->lr_beta, rpb, Y);
    else
        RL_LAUNCH((k_lr_expand<R, false>), dim3(nbx, (nrows + rpb - 1) / rpb), dim3(256), 0,
                  st, zhat, nrows, g->m, (const double*)g->lr_beta, rpb, Y);
}
template <int R>
static void lr_launch(rl_gridop* g, const double* X, double* Y, int nvec, int Q, const double* Cq,
                      const double* Bq, hipStream_t st, int accumulate = 0) {
    const int nrows = nvec * g->D;
    const bool deferred = g->defer_expand && !accumulate && R <= g->kn.w_poly_rmax;
    // (Measured and dropped, round 5: a large batch as TWO half batches, the second one a
    // projection behind on a side stream, so that the first half's expansion writes while the
    // second half's projection reads -- the box copies the step's vectors at 5.2-5.8 TB/s where
    // reads-then-writes reach 4.3-4.8.  C5, same box, two alternations: 0.482 / 0.482 ms split
    // against 0.458 / 0.428 as one batch (periodic 0.546 / 0.547 against 0.518 / 0.508): the two
    // queues' workgroups do not share the chip the way one kernel's loads and stores do.)
    const int chunks = lr_project<R>(g, X, nrows, st);
    lr_mix<R>(g, g->lr_part, chunks, nvec, Q, Cq, Bq, g->lr_zhat, st);
    // (ski_mvm_int: the W kernel expands, k_spmv_w_poly -- ranks 24 and 32 only: at rank 48
    // the evaluation costs more than the two vector passes it saves -- measured, C5
    // periodic: 5.03 against 4.17 ms per solver round)
    if (deferred) {
        g->expand_deferred = true;
        return;
    }
    lr_expand<R>(g, g->lr_zhat, nrows, Y, accumulate, st);
}

// Y = Phi [sum_q B_q (x) C_q] Phi^T X for tops [q0, q0 + Q) with coupling Bq
static int lr_apply(rl_gridop* g, const double* X, double* Y, int nvec, int q0, int Q,
                    const double* Bq, hipStream_t st) {
    const double* Cq = g->lr_C + (size_t)q0 * g->lr_r * g->lr_r;
    switch (g->lr_r) {
        case 24: lr_launch<24>(g, X, Y, nvec, Q, Cq, Bq, st); break;
        case 32: lr_launch<32>(g, X, Y, nvec, Q, Cq, Bq, st); break;
        case 36: lr_launch<36>(g, X, Y, nvec, Q, Cq, Bq, st); break;
        case 40: lr_launch<40>(g, X, Y, nvec, Q, Cq, Bq, st); break;
        case 48: lr_launch<48>(g, X, Y, nvec, Q, Cq, Bq, st); break;
        default: return fail(RL_EINVAL, "low-rank path: bad basis size");
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// the same for the polynomial tops of an operator that also has filter tops: ADDS to Y,
// which the filter part has written (the streaming expansion absorbs the read-modify-write
// better than the filter kernel's tiles would)
static int lr_apply_compact(rl_gridop* g, const double* X, double* Y, int nvec, hipStream_t st) {
    switch (g->lr_r) {
        case 24: lr_launch<24>(g, X, Y, nvec, g->lr_np, g->lr_Cc, g->lr_Bc, st, 1); break;
        case 32: lr_launch<32>(g, X, Y, nvec, g->lr_np, g->lr_Cc, g->lr_Bc, st, 1); break;
        case 36: lr_launch<36>(g, X, Y, nvec, g->lr_np, g->lr_Cc, g->lr_Bc, st, 1); break;
        case 40: lr_launch<40>(g, X, Y, nvec, g->lr_np, g->lr_Cc, g->lr_Bc, st, 1); break;
        case 48: lr_launch<48>(g, X, Y, nvec, g->lr_np, g->lr_Cc, g->lr_Bc, st, 1); break;
        default: return fail(RL_EINVAL, "low-rank path: bad basis size");
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

static int mvm_with_mix(rl_gridop* g, const MixParams& mp, const double* X, double* Y, int nvec,
                        hipStream_t stream);

// ---------------------------------------------------------------------------
// recursive-filter form (rl_filter.h): detection of exponential-polynomial top
// rows on the host, tables, launches
// ---------------------------------------------------------------------------
struct SfFit {
    int deg = 0;
    long double ah = 0.0L;          // decay per grid step: rho = exp(-ah)
    long double c[3] = {0.0L, 0.0L, 0.0L};
};

// sum_i |t_i - (c0 + c1 i + c2 i^2) exp(-ah i)| <= RL_SF_TOL sum_i |t_i| ?  (long double;
// the power is re-anchored every 256 points; stops at the first excess, so a top
// row of another kind costs a handful of points)
static bool sf_check(const double* t, int m, const SfFit& f, long double tot) {
    const long double budget = (long double)RL_SF_TOL * tot, rho = expl(-f.ah);
    long double err = 0.0L, pw = 1.0L;
    for (int i = 0; i < m; ++i) {
        if ((i & 255) == 0) pw = expl(-f.ah * i);
        const long double model = (f.c[0] + (f.c[1] + f.c[2] * i) * i) * pw;
        err += fabsl((long double)t[i] - model);
        if (!(err <= budget)) return false;
        pw *= rho;
    }
    return true;
}

// Is the top row t_i = (c0 + c1 i + c2 i^2) rho^i?  The decimated sequence
// u_k = t_{k j} of such a row satisfies  sum_l binom(p+1, l) (-R)^l u_{p+1-l} = 0
// with R = rho^j (p: the degree) -- one polynomial equation for R from p + 2
// samples; j is taken where the row has fallen to about 0.6 of its largest
// entry, so that the roots are well separated.  Every root in (0, 1] is a
// candidate; the coefficients follow from the first p + 1 samples; the
// candidate is accepted by sf_check over the whole row.  Degrees 0, 1, 2 in turn.
static bool sf_detect(const double* t, int m, SfFit* fit) {
    if (m < 8) return false;
    long double tot = 0.0L;
    double amax = 0.0;
    int imax = 0;
    for (int i = 0; i < m; ++i) {
        if (!std::isfinite(t[i])) return false;
        const double a = std::fabs(t[i]);
        tot += a;
        if (a > amax) {
            amax = a;
            imax = i;
        }
    }
    if (amax == 0.0) return false;
    int below = -1;
    for (int i = imax; i < m; ++i)
        if (std::fabs(t[i]) <= 0.6 * amax) {
            below = i - imax;
            break;
        }
    for (int deg = 0; deg <= 2; ++deg) {
        const int span = deg + 1, jmax = (m - 1) / span;
        if (jmax < 1) continue;
        const int j = below < 0 ? jmax : std::max(1, std::min(below, jmax));
        long double u[4] = {0.0L, 0.0L, 0.0L, 0.0L}, coef[4];
        for (int k = 0; k <= span; ++k) u[k] = t[(size_t)k * j];
        static const int binom[4][4] = {{1, 0, 0, 0}, {1, 1, 0, 0}, {1, 2, 1, 0}, {1, 3, 3, 1}};
        for (int l = 0; l <= span; ++l) coef[l] = ((l & 1) ? -1.0L : 1.0L) * binom[span][l] * u[span - l];
        auto f = [&](long double R) {
            long double v = 0.0L;
            for (int l = span; l >= 0; --l) v = v * R + coef[l];
            return v;
        };
        std::vector<long double> roots;
        const int NS_ = 512;
        long double Ra = 0.0L, fa = f(0.0L);
        for (int k = 1; k <= NS_; ++k) {
            const long double Rb = (long double)k / NS_, fb = f(Rb);
            if (fb == 0.0L) roots.push_back(Rb);
            else if (fa != 0.0L && ((fa < 0.0L) != (fb < 0.0L))) {
                long double lo = Ra, hi = Rb, flo = fa;
                for (int it = 0; it < 80; ++it) {
                    const long double mid = 0.5L * (lo + hi), fm = f(mid);
                    if (fm == 0.0L) { lo = hi = mid; break; }
                    if ((fm < 0.0L) == (flo < 0.0L)) { lo = mid; flo = fm; } else hi = mid;
                }
                roots.push_back(0.5L * (lo + hi));
            }
            Ra = Rb;
            fa = fb;
        }
        for (long double R : roots) {
            if (!(R > 0.0L && R <= 1.0L)) continue;
            SfFit c;
            c.deg = deg;
            c.ah = -logl(R) / j;
            const long double p0 = u[0], p1 = u[1] / R, p2 = u[2] / (R * R), jj = (long double)j;
            c.c[0] = p0;
            if (deg == 1) c.c[1] = (p1 - p0) / jj;
            if (deg == 2) {
                c.c[2] = (p2 - 2.0L * p1 + p0) / (2.0L * jj * jj);
                c.c[1] = (p1 - p0) / jj - c.c[2] * jj;
            }
            if (sf_check(t, m, c, tot)) {
                *fit = c;
                return true;
            }
        }
    }
    return false;
}

// the chunk states' parity weights of one filter (rl_filter.h: k_sf_carries2): [t][a, b, c, e], t < G / 2,
// from long-double powers, rounded once
static void sf_parity_weights_chunk(const SfFit& f, double* out) {
    for (int t = 0; t < RL_SF_G / 2; ++t) {
        const int mi = RL_SF_G - 1 - t;
        const long double pt = expl(-f.ah * t), pm = expl(-f.ah * mi);
        out[4 * t + 0] = (double)((pt + pm) / 2);
        out[4 * t + 1] = (double)((pt - pm) / 2);
        out[4 * t + 2] = (double)((t * pt + mi * pm) / 2);
        out[4 * t + 3] = (double)((t * pt - mi * pm) / 2);
    }
}
static void sf_device_top(const SfFit& f, int m, SfTop* tp, SfBlk* bk, double* pw) {
    tp->rho = (double)expl(-f.ah);
    for (int k = 0; k < 3; ++k) tp->c[k] = (double)f.c[k];
    tp->rG = (double)expl(-f.ah * RL_SF_G);
    // k_sf_scan chains the chunks in 32 segments of seglen chunks each
    const int nchunks = (m + RL_SF_G - 1) / RL_SF_G, seglen = (nchunks + RL_SF_NSEG - 1) / RL_SF_NSEG;
    tp->rL = (double)expl(-f.ah * RL_SF_G * (long double)seglen);
    for (int j = 0; j <= RL_SF_G; ++j) pw[j] = (double)expl(-f.ah * j);
    bk->rho = tp->rho;
    for (int k = 0; k < 3; ++k) bk->c[k] = tp->c[k];
    for (int n = 0; n <= 16; ++n) bk->p32[n] = (double)expl(-f.ah * RL_SF_S * n);
}

static int sf_nchunks(const rl_gridop* g) { return (g->m + RL_SF_G - 1) / RL_SF_G; }
static size_t sf_need_E(const rl_gridop* g, int nvec, int NF, int NS) {
    return (size_t)sf_nchunks(g) * nvec * g->D * NF * 2 * NS;
}
static size_t sf_need_Cin(const rl_gridop* g, int nvec, int nchan, int NS) {
    return (size_t)sf_nchunks(g) * nvec * nchan * 2 * NS;
}
static bool sf_ready(const rl_gridop* g, int nvec, int NF, int nfac, int NS) {
    return g->sf_E_cap >= sf_need_E(g, nvec, NF, NS) &&
           g->sf_Cin_cap >= sf_need_Cin(g, nvec, g->D * NF + nfac, NS);
}
static int sf_reserve(rl_gridop* g, int nvec, int NF, int nfac, int NS) {
    const size_t needE = sf_need_E(g, nvec, NF, NS), needC = sf_need_Cin(g, nvec, g->D * NF + nfac, NS);
    if (g->sf_E_cap < needE) {
        if (g->sf_E) RL_HIP(hipFree(g->sf_E));
        g->sf_E = nullptr;
        g->sf_E_cap = 0;
        RL_HIP(hipMalloc((void**)&g->sf_E, needE * sizeof(double)));
        g->sf_E_cap = needE;
    }
    if (g->sf_Cin_cap < needC) {
        if (g->sf_Cin) RL_HIP(hipFree(g->sf_Cin));
        g->sf_Cin = nullptr;
        g->sf_Cin_cap = 0;
        RL_HIP(hipMalloc((void**)&g->sf_Cin, needC * sizeof(double)));
        g->sf_Cin_cap = needC;
    }
    return RL_OK;
}
static size_t sf_apply_lds(int D, int nfac, int NF, int nthr) {
    size_t b = ((size_t)(D + nfac) * RL_SF_PAD + sf_blob_doubles(NF, nfac, D) +
                (size_t)(D * NF + nfac) * 2 * 3 + 1) * sizeof(double);
#if defined(RL_EMU)
    b += (size_t)(nthr / 64) * 128 * sizeof(double);
#else
    (void)nthr;
#endif
    return b;
}

// Y = [filter part] X: carries -> scan -> apply
template <int NS>
static int sf_launch(rl_gridop* g, const SfParams& sp, const double* blob, const double* X,
                     double* Y, int nvec, hipStream_t st) {
    const int D = g->D, nrows = nvec * D, nch = sf_nchunks(g);
    if (D < 1 || D > 16) return fail(RL_ELIMIT, "filter form: D outside 1..16");
    // rows per carries workgroup: the filter powers are staged once per workgroup
    const int rpw = nrows >= 16 * 64 ? 64 : 16;
    if (NS == 2 && !g->kn.sf_carries1)
        RL_LAUNCH(k_sf_carries2, dim3(nch, (nrows + rpw - 1) / rpw), dim3(256),
                  ((size_t)sp.NF * (RL_SF_G / 2) * 4 + 256) * sizeof(double), st, X, nrows, g->m,
                  sp.NF, sp.pwp, rpw, g->sf_E);
    else
        RL_LAUNCH((k_sf_carries<NS>), dim3(nch, (nrows + rpw - 1) / rpw), dim3(256),
                  ((size_t)sp.NF * (RL_SF_G + 1) + 256) * sizeof(double), st, X, nrows, g->m, sp.NF,
                  sp.pw, rpw, g->sf_E);
    const int ncd = 2 * (D * sp.NF + sp.nfac);
    if ((nch + RL_SF_NSEG - 1) / RL_SF_NSEG <= RL_SF_SEGMAX && !g->kn.sf_scan2)
        RL_LAUNCH((k_sf_scan1<NS>), dim3((ncd + 7) / 8, nvec), dim3(256), 256 * NS * sizeof(double),
                  st, (const double*)g->sf_E, nch, nvec, D, sp, g->sf_Cin, g->sf_next);
    else
        RL_LAUNCH((k_sf_scan<NS>), dim3((ncd + 7) / 8, nvec), dim3(256), 256 * NS * sizeof(double),
                  st, (const double*)g->sf_E, nch, nvec, D, sp, g->sf_Cin, g->sf_next);
    // persistent workgroups of four waves, two per CU (256 registers a lane: a segment's
    // 32 points and the states of five filters; the tile's LDS allows two at C5), each
    // walking every (2 x CUs)-th tile.  (A workgroup of five waves puts two on the first
    // SIMD, and a second workgroup then finds no room there.)
    const int waves = 4;
#if defined(RL_EMU)
    const int ntiles = nch * nvec, resident = 7;            // (so that tests walk several tiles)
#else
    const size_t tile_lds = sf_apply_lds(D, sp.nfac, sp.NF, 256);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / tile_lds));
    const int ntiles = nch * nvec, resident = per_cu * RL_LR_CUS;
#endif
#define RL_SF_APPLY(D_)                                                                      \
    case D_:                                                                                    \
        RL_LAUNCH((k_sf_apply<NS, D_>), dim3(std::min(ntiles, resident)), dim3(64 * waves),     \
                  sf_apply_lds(D, sp.nfac, sp.NF, 64 * waves), st, X, Y, nvec, g->m, sp.NF,     \
                  sp.nfac, blob, (const double*)g->sf_Cin, g->sf_next);                         \
        break
    switch (D) {
        RL_SF_APPLY(1); RL_SF_APPLY(2); RL_SF_APPLY(3); RL_SF_APPLY(4);
        RL_SF_APPLY(5); RL_SF_APPLY(6); RL_SF_APPLY(7); RL_SF_APPLY(8);
        RL_SF_APPLY(9); RL_SF_APPLY(10); RL_SF_APPLY(11); RL_SF_APPLY(12);
        RL_SF_APPLY(13); RL_SF_APPLY(14); RL_SF_APPLY(15); RL_SF_APPLY(16);
        default: return fail(RL_ELIMIT, "filter form: no k_sf_apply for this D");
    }
#undef RL_SF_APPLY
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// the operator's filter part (every filter top with its couplings)
static int sf_apply_all(rl_gridop* g, const double* X, double* Y, int nvec, hipStream_t st) {
    SfParams sp{g->sf_n, g->sf_nfac, g->sf_tops, g->sf_pw, g->sf_kappa, g->sf_facA, g->sf_facAW,
                g->sf_facJ, g->sf_pwp};
    return g->sf_ns == 3 ? sf_launch<3>(g, sp, g->sf_blob, X, Y, nvec, st)
                         : sf_launch<2>(g, sp, g->sf_blob, X, Y, nvec, st);
}
// (I_D (x) T_q) X for one filter top
static int sf_apply_top(rl_gridop* g, int q, const double* X, double* Y, int nvec, hipStream_t st) {
    const int j = g->sf_slot[q];
    SfParams sp{1, 0, g->sf_tops + j, g->sf_pw + (size_t)j * (RL_SF_G + 1), g->ones, nullptr,
                nullptr, nullptr, g->sf_pwp + (size_t)j * 2 * RL_SF_G};
    const double* blob = g->sf_blob_top + (size_t)j * sf_blob_doubles(1, 0, g->D);
    return g->sf_top_ns[q] == 3 ? sf_launch<3>(g, sp, blob, X, Y, nvec, st)
                                : sf_launch<2>(g, sp, blob, X, Y, nvec, st);
}

// ---------------------------------------------------------------------------
// set-time work: which form does each top row take?
// ---------------------------------------------------------------------------
// small kernels of the polynomial verification (everything stays on the device
// until ONE copy per rank tried brings the verdicts of all tops back)
//   C[i][j] = (c_ij + c_ji) / 2,  c_ij = nu_i sum_chunks part[chunk][row j][i]
//   grid (r): workgroup i, thread (j = tid % 64, chunk class tid / 64 of four)
__global__ void __launch_bounds__(256)
k_lr_finish_C(const double* __restrict__ part, int nparts, int nrows, int r,
              const double* __restrict__ nu, double* __restrict__ C) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);       // [2][4][64]
    const int i = blockIdx.x, j = threadIdx.x & 63, cls = threadIdx.x >> 6;
    double a = 0.0, b = 0.0;
    if (j < r)
        for (int c = cls; c < nparts; c += 4) {
            a += part[((size_t)c * nrows + j) * r + i];
            b += part[((size_t)c * nrows + i) * r + j];
        }
    red[cls * 64 + j] = a;
    red[256 + cls * 64 + j] = b;
    __syncthreads();
    if (cls == 0 && j < r) {
        a = (red[j] + red[64 + j]) + (red[128 + j] + red[192 + j]);
        b = (red[256 + j] + red[320 + j]) + (red[384 + j] + red[448 + j]);
        C[i * r + j] = 0.5 * (nu[i] * a + nu[j] * b);
    }
}
#define RL_LR_NB 64            // workgroups of a comparison
//   out[b][0] = max |y1 - y2|, out[b][1] = max |y1|, out[b][2] = entries that are not finite
__global__ void __launch_bounds__(256)
k_lr_compare(const double* __restrict__ y1, const double* __restrict__ y2, size_t n,
             double* __restrict__ out) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);       // [3][256]
    double dmax = 0.0, ymax = 0.0, bad = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double a = y1[i], b = y2[i], d = fabs(a - b);
        if (!(d <= 1e300)) bad += 1.0;              // (NaN and infinities fail the comparison)
        else dmax = d > dmax ? d : dmax;
        ymax = fabs(a) > ymax ? fabs(a) : ymax;
    }
    const int tid = threadIdx.x;
    red[tid] = dmax;
    red[256 + tid] = ymax;
    red[512 + tid] = bad;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] = red[tid] > red[tid + s] ? red[tid] : red[tid + s];
            red[256 + tid] = red[256 + tid] > red[256 + tid + s] ? red[256 + tid] : red[256 + tid + s];
            red[512 + tid] += red[512 + tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        out[blockIdx.x * 3 + 0] = red[0];
        out[blockIdx.x * 3 + 1] = red[256];
        out[blockIdx.x * 3 + 2] = red[512];
    }
}
//   out[row] = max_i |V[row][i]|   (grid: rows)
__global__ void __launch_bounds__(256)
k_lr_rowmax(const double* __restrict__ V, int m, double* __restrict__ out) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);
    const double* v = V + (size_t)blockIdx.x * m;
    double mx = 0.0;
    for (int i = threadIdx.x; i < m; i += 256) {
        const double a = fabs(v[i]);
        mx = (a > mx || !(a <= 1e300)) ? a : mx;    // (a NaN sticks)
    }
    const int tid = threadIdx.x;
    red[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const double o = red[tid + s];
            if (o > red[tid] || !(o <= 1e300)) red[tid] = o;
        }
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = red[0];
}
// Power iteration of the verification's operator-norm bound (below): two tiny kernels per
// vector block and step, no host round trip.
//   k_lr_pw_diff   out = a - b (b may be NULL: out = a), part[block] = sum of out^2
//   k_lr_pw_scale  v = d / ||d||_2 (d itself when the norm is 0 or not finite), *rec = ||d||_2
//                  -- every workgroup sums the RL_LR_NB partials in the same fixed order
#define RL_LR_NPOW 8           // steps
#define RL_LR_TOL_OP 2e-13     // accepted estimate of ||T - Phi C Phi^T||_2 / ||T||_2
#if defined(RL_EMU)
#define RL_LR_PWB 8            // (the emulator pays per thread it starts, not per element)
#else
#define RL_LR_PWB 512          // workgroups of its vector kernels (64: 20 us per kernel at C5, 10^6 entries)
#endif
//   (grid (RL_LR_PWB, D): row a of the D rows of length m is its own vector -- every row
//   iterates on the top row the selectors of lr_verify gave it)
__global__ void __launch_bounds__(256)
k_lr_pw_diff(const double* a, const double* b, int m, double* out, double* __restrict__ part) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);       // [256]
    const size_t off = (size_t)blockIdx.y * m;
    double s = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
        const double d = b ? a[off + i] - b[off + i] : a[off + i];
        out[off + i] = d;
        s = fma(d, d, s);
    }
    const int tid = threadIdx.x;
    red[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}
//   rec: stat + top_of_row[a] * stride + off  (rows without a top: nothing recorded)
__global__ void __launch_bounds__(256)
k_lr_pw_scale(const double* __restrict__ d, int m, const double* __restrict__ part,
              double* __restrict__ v, double* __restrict__ stat, const int* __restrict__ top_of_row,
              int stride, int off) {
    RL_SMEM(smem);
    double* red = reinterpret_cast<double*>(smem);       // [256]
    const int row = blockIdx.y, tid = threadIdx.x;
    // (every workgroup sums the row's partials itself, in the same fixed order: strided per
    // thread, then a tree -- one thread adding 512 of them in turn was 90 us of this kernel)
    double sp = 0.0;
    for (int b = tid; b < (int)gridDim.x; b += 256) sp += part[(size_t)row * gridDim.x + b];
    red[tid] = sp;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    const double nrm = sqrt(red[0]);
    const double inv = (nrm > 0.0 && nrm <= 1e300) ? 1.0 / nrm : 1.0;
    const size_t o = (size_t)row * m;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) v[o + i] = d[o + i] * inv;
    if (blockIdx.x == 0 && threadIdx.x == 0 && top_of_row[row] >= 0)
        stat[(size_t)top_of_row[row] * stride + off] = nrm;
}
// per-top verdict record on the device: RL_LR_NB x 3 comparison partials, the row maxima
// of T Phi_j, j < RL_LR_RMAX + 16, then the power iteration's norms ||E v_k||, ||T w_k||
#define RL_LR_STATW (RL_LR_NB * 3 + RL_LR_RMAX + 16 + 2 * RL_LR_NPOW)

// Polynomial verification of the tops with want[q] != 0: builds C_q = Phi^T T_q Phi
// with the transform kernels of this handle and accepts, per top and rank,
//   (i)  the product of a fixed random vector through both forms agrees to
//        RL_LR_TOL of its largest entry (the D blocks are D independent trials),
//   (ii) the first RL_LR_EXTRA orthonormal polynomials the rank OMITS -- for which
//        the form returns zero by construction -- have  max|T Phi_{r+k}| <=
//        RL_LR_TOL max_j max|T Phi_j|: the operator's action outside the subspace
//        is at roundoff (this is the adversarial input of the form; a random
//        vector only carries 1/sqrt(m) of its norm in any one direction),
//   (iii) nothing in either product is NaN or infinite,
//   (iv) a BOUND instead of a draw (round 5): RL_LR_NPOW steps of the power iteration on
//        E = T_q - Phi C_q Phi^T through the two products of this handle, started from the
//        fixed random vector, against the same iteration on T_q itself:
//        max_k ||E v_k||_2 <= RL_LR_TOL_OP ||T w||_2.  E is symmetric, so ||E v_k|| climbs
//        monotonically to ||E||_2: whatever direction the form is worst in -- a component
//        the fixed trial vector happens to miss, an omitted polynomial beyond the four of
//        (ii) -- grows by the ratio of E's leading eigenvalues at every step, and the
//        accepted quantity is the operator's error for EVERY input relative to ||T||_2.
//        (tests: an RBF row plus a small cosine whose frequency the trial vector is blind
//        to passes (i)-(iii) and is rejected here; RUNLMC_NO_LR_BOUND, debug, skips it.)
// Tries r = 24, 32, 48: the first rank that every wanted top passes is used; if
// none, rank 48 with whatever passes.  pass[q] is set for the tops in the form.
static int lr_verify(rl_gridop* g, const std::vector<char>& want, std::vector<char>* pass) {
    const int D = g->D, m = g->m, Q = g->Q;
    const size_t vec = (size_t)D * m;
    pass->assign(Q, 0);
    if (!g->lr_phiJ) {
        RL_TRY(lr_make_basis(g));
        // verdict records in front of C, so that one copy brings both back
        double* cs = nullptr;
        RL_HIP(hipMalloc((void**)&cs, (size_t)g->max_tops * (RL_LR_STATW + RL_LR_RMAX * RL_LR_RMAX) * sizeof(double)));
        // (zeros: the grouped power iteration multiplies the C of tops that are no candidates by a
        // zero coupling -- they have to be finite)
        RL_HIP(hipMemset(cs, 0, (size_t)g->max_tops * (RL_LR_STATW + RL_LR_RMAX * RL_LR_RMAX) * sizeof(double)));
        g->lr_stat = cs;
        g->lr_C = cs + (size_t)g->max_tops * RL_LR_STATW;
        RL_HIP(hipMalloc((void**)&g->lr_Cc, (size_t)g->max_tops * RL_LR_RMAX * RL_LR_RMAX * sizeof(double)));
        RL_HIP(hipMalloc((void**)&g->lr_B, (size_t)g->max_tops * D * D * sizeof(double)));
        RL_HIP(hipMalloc((void**)&g->lr_Bc, (size_t)g->max_tops * D * D * sizeof(double)));
        std::vector<double> eye((size_t)D * D, 0.0);
        for (int a = 0; a < D; ++a) eye[(size_t)a * D + a] = 1.0;
        RL_TRY(upload(&g->lr_eye, eye));
        // scratch: a fixed random vector, two results, the packed T Phi block
        const size_t nvr = (RL_LR_RMAX + RL_LR_EXTRA + D - 1) / D;
        RL_HIP(hipMalloc((void**)&g->lr_scr, (3 + nvr) * vec * sizeof(double)));
        // power iteration: [v | w], [T v | T w], partial sums
        RL_HIP(hipMalloc((void**)&g->lr_pw, (4 * vec + 2 * (size_t)RL_LR_PWB * D) * sizeof(double)));
        // selector couplings of the grouped power iteration, per group of D candidate tops:
        // kappa [Q][D] | B [Q][D][D] | top of each output row [D] (ints, padded to doubles)
        RL_HIP(hipMalloc((void**)&g->lr_sel, (size_t)g->max_tops *
                         ((size_t)g->max_tops * D + (size_t)g->max_tops * D * D + D) * sizeof(double)));
        std::vector<double> xr(vec);
        unsigned long long st = 0x9E3779B97F4A7C15ull;          // fixed seed: same trials every time
        for (size_t i = 0; i < vec; ++i) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            xr[i] = ((double)(st >> 11) / 9007199254740992.0) * 2.0 - 1.0;
        }
        // every output row of the trial vector has unit 2-norm: the power iteration of (iv)
        // starts from it, and its step-0 entries ||E v_0||, ||T w_0|| are then Rayleigh-type
        // lower estimates like the later ones (unnormalised they carried a factor ~sqrt(m / 3)
        // into the recorded ||T||_2 and loosened the test by it; the sampled tests (i), (ii)
        // are relative to the result's largest entry and do not see the scale)
        for (int a = 0; a < D; ++a) {
            long double ss = 0.0L;
            for (int i = 0; i < m; ++i) ss += (long double)xr[(size_t)a * m + i] * xr[(size_t)a * m + i];
            const double inv = ss > 0.0L ? (double)(1.0L / sqrtl(ss)) : 1.0;
            for (int i = 0; i < m; ++i) xr[(size_t)a * m + i] *= inv;
        }
        RL_HIP(hipMemcpy(g->lr_scr, xr.data(), vec * sizeof(double), hipMemcpyHostToDevice));
    }
    double* xr = g->lr_scr;
    double* y1 = xr + vec;
    double* y2 = y1 + vec;
    double* tphi = y2 + vec;
    hipStream_t st = nullptr;
    std::vector<double> back;
    std::vector<char> best;
    // candidate tops in groups of D, each group's selectors (see (iv) below)
    std::vector<int> cand;
    for (int q = 0; q < Q; ++q)
        if (want[q]) cand.push_back(q);
    const size_t sel_stride = (size_t)Q * D + (size_t)Q * D * D + D;
    {
        const size_t ngroups = (cand.size() + D - 1) / D;
        std::vector<double> sel(std::max<size_t>(1, ngroups) * sel_stride, 0.0);
        for (size_t grp = 0; grp < ngroups; ++grp) {
            double* ks = sel.data() + grp * sel_stride;
            double* bs = ks + (size_t)Q * D;
            int* rows = reinterpret_cast<int*>(bs + (size_t)Q * D * D);
            for (int a = 0; a < D; ++a) rows[a] = -1;
            for (int a = 0; a < D && grp * D + a < cand.size(); ++a) {
                const int q = cand[grp * D + a];
                ks[(size_t)q * D + a] = 1.0;
                bs[((size_t)q * D + a) * D + a] = 1.0;
                rows[a] = q;
            }
        }
        RL_HIP(hipMemcpy(g->lr_sel, sel.data(), sel.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    for (int r : {24, 32, 36, 40, 48}) {
        // (a caller who knows where a sibling handle's rows were accepted starts the ladder
        // there: rl_gridop_set_rank_hint)
        if (r < g->lr_rank_hint && r < 48) continue;
        g->lr_r = r;
        const int nvr = (r + RL_LR_EXTRA + D - 1) / D, nrows = nvr * D;
        RL_TRY(lr_reserve(g, std::max(nvr, 1)));
        // (the grouped power iteration multiplies the C of tops that are no candidates by a zero
        // coupling: they have to be finite -- a slot may hold what an earlier parameter set, or
        // another rank's layout, left there)
        RL_HIP(hipMemsetAsync(g->lr_C, 0, (size_t)g->max_tops * RL_LR_RMAX * RL_LR_RMAX * sizeof(double), st));
        for (int q = 0; q < Q; ++q) {
            if (!want[q]) continue;
            MixParams mp{1, 0, g->spec + (size_t)q * g->L, nullptr, nullptr, nullptr, g->ones,
                         nullptr, nullptr};
            double* stat = g->lr_stat + (size_t)q * RL_LR_STATW;
            // rows j of tphi = T_q Phi_j (the function-major basis IS a batch of vectors)
            g->lr_bypass = true;
            int rc = mvm_with_mix(g, mp, g->lr_phiJ, tphi, nvr, st);
            g->lr_bypass = false;
            if (rc != RL_OK) return rc;
            RL_LAUNCH(k_lr_rowmax, dim3(nrows), dim3(256), 256 * sizeof(double), st,
                      (const double*)tphi, m, stat + RL_LR_NB * 3);
            // C_q[i][j] = Phi_i . (T_q Phi_j): the projection of those rows
            int nparts = 0;
            switch (r) {
                case 24: nparts = lr_project<24>(g, tphi, nrows, st); break;
                case 32: nparts = lr_project<32>(g, tphi, nrows, st); break;
                case 36: nparts = lr_project<36>(g, tphi, nrows, st); break;
                case 40: nparts = lr_project<40>(g, tphi, nrows, st); break;
                default: nparts = lr_project<48>(g, tphi, nrows, st); break;
            }
            RL_LAUNCH(k_lr_finish_C, dim3(r), dim3(256), 512 * sizeof(double), st,
                      (const double*)g->lr_part, nparts, nrows, r, (const double*)g->lr_nu,
                      g->lr_C + (size_t)q * r * r);
            // trial: T_q xr through both forms
            g->lr_bypass = true;
            rc = mvm_with_mix(g, mp, xr, y1, 1, st);
            g->lr_bypass = false;
            if (rc != RL_OK) return rc;
            RL_TRY(lr_apply(g, xr, y2, 1, q, 1, g->lr_eye, st));
            RL_LAUNCH(k_lr_compare, dim3(RL_LR_NB), dim3(256), 3 * 256 * sizeof(double), st,
                      (const double*)y1, (const double*)y2, vec, stat);
        }
        // (iv) power iteration on E = T - Phi C Phi^T (vector v) and on T (vector w), both from
        // the trial vector.  ALL candidate tops at once: selector couplings give output row a
        // the top row of slot a (kappa_q = e_a for the transform kernels, B_q = e_a e_a^T for
        // the polynomial form), so one two-vector transform product [v | w] and one
        // polynomial product advance every top's iteration, D tops per group -- per step ten
        // launches whatever Q (one group per step and top, as first built: 2.9 ms per update
        // at C2 against 0.42 for the sampled tests alone).
        for (size_t grp = 0; grp * D < cand.size(); ++grp) {
            double* vw = g->lr_pw;
            double* Tvw = vw + 2 * vec;
            double* parts = Tvw + 2 * vec;
            const double* ksel = g->lr_sel + grp * sel_stride;
            const double* bsel = ksel + (size_t)Q * D;
            const int* rows = reinterpret_cast<const int*>(bsel + (size_t)Q * D * D);
            MixParams mps{Q, 0, g->spec, nullptr, nullptr, nullptr, ksel, nullptr, nullptr};
            RL_HIP(hipMemcpyAsync(vw, xr, vec * sizeof(double), hipMemcpyDeviceToDevice, st));
            RL_HIP(hipMemcpyAsync(vw + vec, xr, vec * sizeof(double), hipMemcpyDeviceToDevice, st));
            const int recE = RL_LR_NB * 3 + RL_LR_RMAX + 16;
            for (int k = 0; k < RL_LR_NPOW; ++k) {
                g->lr_bypass = true;
                int rc = mvm_with_mix(g, mps, vw, Tvw, 2, st);
                g->lr_bypass = false;
                if (rc != RL_OK) return rc;
                RL_TRY(lr_apply(g, vw, y2, 1, 0, Q, bsel, st));
                const dim3 pg(RL_LR_PWB, D);
                RL_LAUNCH(k_lr_pw_diff, pg, dim3(256), 256 * sizeof(double), st, (const double*)Tvw,
                          (const double*)y2, m, Tvw, parts);
                RL_LAUNCH(k_lr_pw_scale, pg, dim3(256), 256 * sizeof(double), st, (const double*)Tvw, m,
                          (const double*)parts, vw, g->lr_stat, rows, RL_LR_STATW, recE + k);
                RL_LAUNCH(k_lr_pw_diff, pg, dim3(256), 256 * sizeof(double), st,
                          (const double*)(Tvw + vec), (const double*)nullptr, m, Tvw + vec,
                          parts + (size_t)RL_LR_PWB * D);
                RL_LAUNCH(k_lr_pw_scale, pg, dim3(256), 256 * sizeof(double), st,
                          (const double*)(Tvw + vec), m, (const double*)(parts + (size_t)RL_LR_PWB * D),
                          vw + vec, g->lr_stat, rows, RL_LR_STATW, recE + RL_LR_NPOW + k);
            }
        }
        RL_HIP(hipGetLastError());
        // the one round trip of this rank: verdict records of all tops + their C
        back.resize((size_t)g->max_tops * RL_LR_STATW + (size_t)Q * r * r);
        RL_HIP(hipMemcpy(back.data(), g->lr_stat, back.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<char> ok(Q, 0);
        bool all = true;
        for (int q = 0; q < Q; ++q) {
            if (!want[q]) continue;
            const double* stat = back.data() + (size_t)q * RL_LR_STATW;
            double dmax = 0.0, ymax = 0.0, bad = 0.0;
            for (int b = 0; b < RL_LR_NB; ++b) {
                dmax = std::max(dmax, stat[b * 3]);
                ymax = std::max(ymax, stat[b * 3 + 1]);
                bad += stat[b * 3 + 2];
            }
            const double* rm = stat + RL_LR_NB * 3;
            double inside = 0.0, outside = 0.0;
            bool finite = bad == 0.0 && std::isfinite(ymax);
            for (int j = 0; j < r + RL_LR_EXTRA; ++j) {
                if (!std::isfinite(rm[j])) finite = false;
                (j < r ? inside : outside) = std::max(j < r ? inside : outside, rm[j]);
            }
            const bool trial = dmax <= RL_LR_TOL * ymax || (ymax == 0.0 && dmax == 0.0);
            const bool tail = outside <= RL_LR_TOL * inside || (inside == 0.0 && outside == 0.0);
            // (iv): the largest ||E v_k|| against ||T w|| after the last step
            const double* rec = rm + RL_LR_RMAX + 16;
            double sigE = 0.0, sigT = 0.0;
            for (int k = 0; k < RL_LR_NPOW; ++k) {
                if (!std::isfinite(rec[k]) || !std::isfinite(rec[RL_LR_NPOW + k])) finite = false;
                sigE = std::max(sigE, rec[k]);
                sigT = std::max(sigT, rec[RL_LR_NPOW + k]);
            }
            const bool bound = g->kn.no_lr_bound || sigE <= RL_LR_TOL_OP * sigT || (sigE == 0.0 && sigT == 0.0);
            if ((int)g->lr_vstat.size() < 4 * g->max_tops) g->lr_vstat.assign((size_t)4 * g->max_tops, 0.0);
            g->lr_vstat[4 * q + 0] = ymax > 0.0 ? dmax / ymax : dmax;
            g->lr_vstat[4 * q + 1] = inside > 0.0 ? outside / inside : outside;
            g->lr_vstat[4 * q + 2] = sigE;
            g->lr_vstat[4 * q + 3] = sigT;
            ok[q] = finite && trial && tail && bound;
            all = all && ok[q];
        }
        best = ok;
        if (all) break;
    }
    *pass = best;
    const int r = g->lr_r;
    g->lr_hC.assign(back.begin() + (size_t)g->max_tops * RL_LR_STATW,
                    back.begin() + (size_t)g->max_tops * RL_LR_STATW + (size_t)Q * r * r);
    return RL_OK;
}

// Decides the form of every top row for the current parameters and builds the
// tables of the forms in use:
//   lr_ok  every top is in the polynomial form (rl_lowrank.h) -- as before;
//   st_ok  every top is in the polynomial or in the filter form and at least one in
//          the latter: the operator is  [polynomial part] + [filter part];
//   otherwise operator products run on the transform kernels; single-top products
//   (rl_gridop_mvm_top: the gradient's dK products) still take each top's own form.
static int forms_setup(rl_gridop* g, const std::vector<double>& A, const std::vector<double>& W,
                       const std::vector<int>& Qi, const std::vector<double>& kap) {
    const int D = g->D, m = g->m, Q = g->Q;
    g->lr_ok = false;
    g->st_ok = false;
    g->top_form.assign(Q, 0);
    g->sf_slot.assign(Q, -1);
    g->sf_top_ns.assign(Q, 2);
    g->sf_n = 0;
    g->sf_nfac = 0;
    g->lr_np = 0;
    if (!g->lr_try && !g->sf_try) return RL_OK;
    // dense couplings B_q = sum_{f of q} w_f a_f a_f^T + diag(kappa_q): the polynomial form's
    // products and the factorisations of rl_solve.hip (exact or as a preconditioner) use them
    std::vector<double> B((size_t)Q * D * D, 0.0);
    for (size_t f = 0; f < W.size(); ++f)
        for (int a = 0; a < D; ++a)
            for (int b = 0; b < D; ++b)
                B[((size_t)Qi[f] * D + a) * D + b] += W[f] * A[f * D + a] * A[f * D + b];
    for (int q = 0; q < Q; ++q)
        for (int a = 0; a < D; ++a) B[((size_t)q * D + a) * D + a] += kap[(size_t)q * D + a];
    g->lr_hB = B;
    // 1. exponential-polynomial tops (host, from the rows themselves)
    std::vector<SfFit> fits(Q);
    int nfilt = 0;
    if (g->sf_try && g->h_tops.size() >= (size_t)Q * m)
        for (int q = 0; q < Q && nfilt < RL_SF_MAXTOPS; ++q)
            if (sf_detect(g->h_tops.data() + (size_t)q * m, m, &fits[q])) {
                g->top_form[q] = 2;
                g->sf_slot[q] = nfilt++;
                g->sf_top_ns[q] = fits[q].c[2] != 0.0L ? 3 : 2;
            }
    g->sf_n = nfilt;
    // 2. polynomial verification of the others (device; skipped while backing off
    //    after rejections in a row: 0, 1, 3, ... 31 parameter updates)
    std::vector<char> want(Q, 0), pass(Q, 0);
    int nwant = 0;
    for (int q = 0; q < Q; ++q)
        if (g->top_form[q] == 0) {
            want[q] = 1;
            ++nwant;
        }
    if (nwant && g->lr_try) {
        if (g->lr_skip > 0) {
            --g->lr_skip;
        } else {
            RL_TRY(lr_verify(g, want, &pass));
            bool all = true;
            for (int q = 0; q < Q; ++q)
                if (want[q]) {
                    if (pass[q]) g->top_form[q] = 1;
                    else all = false;
                }
            if (all) {
                g->lr_rejects = 0;
            } else {
                g->lr_rejects = std::min(g->lr_rejects + 1, 6);
                g->lr_skip = (1 << (g->lr_rejects - 1)) - 1;
            }
        }
    }
    int npoly = 0, nfft = 0;
    for (int q = 0; q < Q; ++q) {
        npoly += g->top_form[q] == 1;
        nfft += g->top_form[q] == 0;
    }
    g->lr_np = npoly;
    // 3. tables.  Filter tops: parameters + powers (single-top products need them
    //    whatever the operator as a whole does)
    if (nfilt) {
        // (each buffer under its own guard: an allocation that fails part-way is retried
        // at the next parameter update instead of leaving later pointers null)
        auto need = [](void** p, size_t bytes) -> int {
            if (*p) return RL_OK;
            RL_HIP(hipMalloc(p, bytes));
            return RL_OK;
        };
        RL_TRY(need((void**)&g->sf_tops, (size_t)g->max_tops * sizeof(SfTop)));
        RL_TRY(need((void**)&g->sf_next, sizeof(int)));
        RL_TRY(need((void**)&g->sf_blob, (size_t)sf_blob_doubles(g->max_tops, g->max_fac, D) * sizeof(double)));
        RL_TRY(need((void**)&g->sf_blob_top, (size_t)g->max_tops * sf_blob_doubles(1, 0, D) * sizeof(double)));
        RL_TRY(need((void**)&g->sf_pw, (size_t)g->max_tops * (RL_SF_G + 1) * sizeof(double)));
        RL_TRY(need((void**)&g->sf_pwp, (size_t)g->max_tops * 2 * RL_SF_G * sizeof(double)));
        RL_TRY(need((void**)&g->sf_kappa, (size_t)g->max_tops * D * sizeof(double)));
        RL_TRY(need((void**)&g->sf_facA, (size_t)std::max(g->max_fac, 1) * D * sizeof(double)));
        RL_TRY(need((void**)&g->sf_facAW, (size_t)std::max(g->max_fac, 1) * D * sizeof(double)));
        RL_TRY(need((void**)&g->sf_facJ, (size_t)std::max(g->max_fac, 1) * sizeof(int)));
        std::vector<SfTop> tops(nfilt);
        std::vector<SfBlk> blks(nfilt);
        std::vector<double> pw((size_t)nfilt * (RL_SF_G + 1)), kp((size_t)nfilt * D), fa, faw;
        std::vector<double> pwpv((size_t)nfilt * 2 * RL_SF_G);
        std::vector<int> fj;
        int ns = 2;
        for (int q = 0; q < Q; ++q) {
            const int j = g->sf_slot[q];
            if (j < 0) continue;
            sf_device_top(fits[q], m, &tops[j], &blks[j], pw.data() + (size_t)j * (RL_SF_G + 1));
            sf_parity_weights_chunk(fits[q], pwpv.data() + (size_t)j * 2 * RL_SF_G);
            for (int a = 0; a < D; ++a) kp[(size_t)j * D + a] = kap[(size_t)q * D + a];
            ns = std::max(ns, g->sf_top_ns[q]);
        }
        for (size_t f = 0; f < W.size(); ++f) {
            const int j = g->sf_slot[Qi[f]];
            if (j < 0) continue;
            for (int a = 0; a < D; ++a) {
                fa.push_back(A[f * D + a]);
                faw.push_back(W[f] * A[f * D + a]);
            }
            fj.push_back(j);
        }
        g->sf_ns = ns;
        g->sf_nfac = (int)fj.size();
        RL_HIP(hipMemcpy(g->sf_tops, tops.data(), tops.size() * sizeof(SfTop), hipMemcpyHostToDevice));
        RL_HIP(hipMemcpy(g->sf_pw, pw.data(), pw.size() * sizeof(double), hipMemcpyHostToDevice));
        RL_HIP(hipMemcpy(g->sf_pwp, pwpv.data(), pwpv.size() * sizeof(double), hipMemcpyHostToDevice));
        RL_HIP(hipMemcpy(g->sf_kappa, kp.data(), kp.size() * sizeof(double), hipMemcpyHostToDevice));
        // the blocks k_sf_apply stages in LDS: the whole filter part, and every top alone
        {
            const int nf = (int)fj.size();
            auto build = [&](int NFb, int nfb, const double* kapb, const double* fab,
                             const double* fawb, const int* fjb, const SfBlk* bk, double* out) {
                double* o = out;
                for (int e = 0; e < NFb * D; ++e) *o++ = kapb[e];
                for (int e = 0; e < nfb * D; ++e) *o++ = fab[e];
                for (int e = 0; e < nfb * D; ++e) *o++ = fawb[e];
                for (int f = 0; f < nfb; ++f) *o++ = (double)fjb[f];
                std::memcpy(o, bk, (size_t)NFb * sizeof(SfBlk));
            };
            std::vector<double> blob(sf_blob_doubles(nfilt, nf, D));
            build(nfilt, nf, kp.data(), fa.data(), faw.data(), fj.data(), blks.data(), blob.data());
            RL_HIP(hipMemcpy(g->sf_blob, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice));
            const int one = sf_blob_doubles(1, 0, D);
            std::vector<double> tblob((size_t)nfilt * one), ones(D, 1.0);
            for (int j = 0; j < nfilt; ++j)
                build(1, 0, ones.data(), nullptr, nullptr, nullptr, &blks[j], tblob.data() + (size_t)j * one);
            RL_HIP(hipMemcpy(g->sf_blob_top, tblob.data(), tblob.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        if (!fj.empty()) {
            RL_HIP(hipMemcpy(g->sf_facA, fa.data(), fa.size() * sizeof(double), hipMemcpyHostToDevice));
            RL_HIP(hipMemcpy(g->sf_facAW, faw.data(), faw.size() * sizeof(double), hipMemcpyHostToDevice));
            RL_HIP(hipMemcpy(g->sf_facJ, fj.data(), fj.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    if (nfft > 0) return RL_OK;          // some top needs the transforms: so does the operator
    if (npoly) {
        const int r = g->lr_r;
        if (nfilt == 0) {
            RL_HIP(hipMemcpy(g->lr_B, B.data(), B.size() * sizeof(double), hipMemcpyHostToDevice));
            g->lr_ok = true;
            g->lr_Mf_ok = false;
            if ((size_t)D * m <= RL_LR_SMALL_MAX && !g->kn.no_lr_small && (int)g->lr_hnu.size() >= r) {
                // the whole coefficient map of the small-batch product (k_lr_small_expand)
                //   Mf[a][i][b][j] = nu_i nu_j sum_q B_q[a][b] C_q[i][j]
                const size_t Dr = (size_t)D * r;
                std::vector<double> Mf(Dr * Dr, 0.0);
                for (int q = 0; q < Q; ++q)
                    for (int a = 0; a < D; ++a)
                        for (int b = 0; b < D; ++b) {
                            const double bq = B[((size_t)q * D + a) * D + b];
                            if (bq == 0.0) continue;
                            for (int i = 0; i < r; ++i)
                                for (int j = 0; j < r; ++j)
                                    Mf[((size_t)a * r + i) * Dr + (size_t)b * r + j] +=
                                        bq * g->lr_hnu[i] * g->lr_hnu[j] * g->lr_hC[((size_t)q * r + i) * r + j];
                        }
                if (g->lr_Mf_cap < Mf.size()) {
                    if (g->lr_Mf) RL_HIP(hipFree(g->lr_Mf));
                    g->lr_Mf = nullptr;
                    g->lr_Mf_cap = 0;
                    RL_HIP(hipMalloc((void**)&g->lr_Mf, Mf.size() * sizeof(double)));
                    g->lr_Mf_cap = Mf.size();
                }
                RL_HIP(hipMemcpy(g->lr_Mf, Mf.data(), Mf.size() * sizeof(double), hipMemcpyHostToDevice));
                g->lr_Mf_ok = true;
            }
            if (r == RL_LR_RS && !g->kn.no_poly_round) {
                // the whole coefficient map of the solver's polynomial rounds
                //   M[a][i][b][j] = nu_i nu_j sum_q B_q[a][b] C_q[i][j]
                std::vector<double> hnu(RL_LR_RMAX);
                RL_HIP(hipMemcpy(hnu.data(), g->lr_nu, hnu.size() * sizeof(double), hipMemcpyDeviceToHost));
                std::vector<double> M((size_t)D * r * D * r, 0.0);
                for (int q = 0; q < Q; ++q)
                    for (int a = 0; a < D; ++a)
                        for (int b = 0; b < D; ++b) {
                            const double bq = B[((size_t)q * D + a) * D + b];
                            if (bq == 0.0) continue;
                            for (int i = 0; i < r; ++i)
                                for (int j = 0; j < r; ++j)
                                    M[(((size_t)a * r + i) * D + b) * r + j] +=
                                        bq * g->lr_hC[((size_t)q * r + i) * r + j];
                        }
                for (int a = 0; a < D; ++a)
                    for (int i = 0; i < r; ++i)
                        for (int b = 0; b < D; ++b)
                            for (int j = 0; j < r; ++j)
                                M[(((size_t)a * r + i) * D + b) * r + j] *= hnu[i] * hnu[j];
                if (!g->lr_M) RL_HIP(hipMalloc((void**)&g->lr_M, M.size() * sizeof(double)));
                RL_HIP(hipMemcpy(g->lr_M, M.data(), M.size() * sizeof(double), hipMemcpyHostToDevice));
            }
            return RL_OK;
        }
        // polynomial tops next to filter tops: their C and B contiguous
        std::vector<double> Bc, Cc;
        for (int q = 0; q < Q; ++q) {
            if (g->top_form[q] != 1) continue;
            Bc.insert(Bc.end(), B.begin() + (size_t)q * D * D, B.begin() + (size_t)(q + 1) * D * D);
            Cc.insert(Cc.end(), g->lr_hC.begin() + (size_t)q * r * r,
                      g->lr_hC.begin() + (size_t)(q + 1) * r * r);
        }
        RL_HIP(hipMemcpy(g->lr_Bc, Bc.data(), Bc.size() * sizeof(double), hipMemcpyHostToDevice));
        RL_HIP(hipMemcpy(g->lr_Cc, Cc.data(), Cc.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // the filter part's tile (D + nfac rows of RL_SF_PAD doubles) has to fit LDS
    // (and its incoming states four registers of each of 256 threads)
    g->st_ok = nfilt > 0 && sf_apply_lds(D, g->sf_nfac, nfilt, 512) <= kLdsHard &&
               (D * nfilt + g->sf_nfac) * 2 * 3 <= 4 * 256;
    // (few outputs, a rank-48 polynomial part AND a filter part: the transform kernels are
    // level or ahead -- measured at D = 4, Q = 3, m = 5000, 1024 vectors: 0.36 against 0.33 ms;
    // at D = 10 the two parts together take 1.65 against 2.7 ms)
    if (g->st_ok && npoly > 0 && g->lr_r >= 40 && D < 8) g->st_ok = false;
    return RL_OK;
}

// before a capture: pending verification and buffers for batches of nvec vectors
int lr_prepare(rl_gridop* g, int nvec) {
    if ((!g->lr_try && !g->sf_try) || (size_t)nvec * g->D * g->m < g->lr_min) return RL_OK;
    RL_TRY(lr_ensure(g));
    if (g->lr_ok || (g->st_ok && g->lr_np)) RL_TRY(lr_reserve(g, nvec));
    if (g->st_ok) RL_TRY(sf_reserve(g, nvec, g->sf_n, g->sf_nfac, g->sf_ns));
    return RL_OK;
}

// runs the verification the last parameter update left pending
int lr_ensure(rl_gridop* g) {
    if (!g->lr_dirty) return RL_OK;
    g->lr_dirty = false;
    RL_HIP(hipSetDevice(g->device));
    return forms_setup(g, g->lr_A, g->lr_W, g->lr_Qi, g->lr_kap);
}

// C_q = Phi_R^T T_q Phi_R of EVERY top row on the first R orthonormal polynomials, host out
// [Q][R][R] (symmetrised), for the factorisations of rl_solve.hip: exact[q] = 1 when the row is IN the
// polynomial form at that rank (C from the accepted form), else the row's projection computed here
// through the transform kernels -- what a preconditioner built on the polynomial subspace
// takes for a Matern row; captured[q] = trace(C_q) / trace(T_q): the share of the row's
// spectrum the subspace holds.
int lr_all_coeffs(rl_gridop* g, int R, std::vector<double>* hC, std::vector<char>* exact,
                  std::vector<double>* captured) {
    const int D = g->D, m = g->m, Q = g->Q;
    if (g->wide || !g->lr_try || Q < 1) return fail(RL_ELIMIT, "no polynomial basis on this grid");
    if (R != 24 && R != 32 && R != 36 && R != 40 && R != 48) return fail(RL_EINVAL, "bad basis size");
    RL_HIP(hipSetDevice(g->device));
    RL_TRY(lr_ensure(g));
    if (!g->lr_phiJ) {
        // (a handle none of whose rows was ever a candidate has no basis yet: the verification's
        // set-up, with nothing to verify; the basis size it leaves behind is nobody's)
        std::vector<char> none((size_t)Q, 0), pass;
        RL_TRY(lr_verify(g, none, &pass));
    }
    if (g->lr_np == 0) g->lr_r = R;
    if (!g->lr_Cx) RL_HIP(hipMalloc((void**)&g->lr_Cx, (size_t)RL_LR_RMAX * RL_LR_RMAX * sizeof(double)));
    hC->assign((size_t)Q * R * R, 0.0);
    exact->assign(Q, 0);
    captured->assign(Q, 0.0);
    const size_t vec = (size_t)D * m;
    double* tphi = g->lr_scr + 3 * vec;
    hipStream_t st = nullptr;
    std::vector<double> one((size_t)R * R);
    for (int q = 0; q < Q; ++q) {
        double* dst = hC->data() + (size_t)q * R * R;
        if ((int)g->top_form.size() > q && g->top_form[q] == 1 && g->lr_r == R &&
            g->lr_hC.size() >= (size_t)(q + 1) * R * R) {
            std::memcpy(one.data(), g->lr_hC.data() + (size_t)q * R * R, one.size() * sizeof(double));
            (*exact)[q] = 1;
        } else {
            const int nvr = (R + D - 1) / D, nrows = nvr * D;
            RL_TRY(lr_reserve(g, std::max(nvr, 1)));
            MixParams mp{1, 0, g->spec + (size_t)q * g->L, nullptr, nullptr, nullptr, g->ones, nullptr, nullptr};
            g->lr_bypass = true;
            const int rc = mvm_with_mix(g, mp, g->lr_phiJ, tphi, nvr, st);
            g->lr_bypass = false;
            if (rc != RL_OK) return rc;
            int nparts = 0;
            switch (R) {
                case 24: nparts = lr_project<24>(g, tphi, nrows, st); break;
                case 32: nparts = lr_project<32>(g, tphi, nrows, st); break;
                case 36: nparts = lr_project<36>(g, tphi, nrows, st); break;
                case 40: nparts = lr_project<40>(g, tphi, nrows, st); break;
                default: nparts = lr_project<48>(g, tphi, nrows, st); break;
            }
            RL_LAUNCH(k_lr_finish_C, dim3(R), dim3(256), 512 * sizeof(double), st,
                      (const double*)g->lr_part, nparts, nrows, R, (const double*)g->lr_nu, g->lr_Cx);
            RL_HIP(hipGetLastError());
            RL_HIP(hipMemcpy(one.data(), g->lr_Cx, one.size() * sizeof(double), hipMemcpyDeviceToHost));
        }
        double tr = 0.0;
        for (int i = 0; i < R; ++i) {
            tr += one[(size_t)i * R + i];
            for (int j = 0; j < R; ++j) dst[(size_t)i * R + j] = 0.5 * (one[(size_t)i * R + j] + one[(size_t)j * R + i]);
        }
        const double t0 = g->h_tops.size() > (size_t)q * m ? g->h_tops[(size_t)q * m] : 0.0;
        (*captured)[q] = t0 > 0.0 ? tr / (t0 * m) : 0.0;
        for (size_t e = 0; e < (size_t)R * R; ++e)
            if (!std::isfinite(dst[e])) return fail(RL_EINVAL, "a top row's projection is not finite");
    }
    return RL_OK;
}

// ---------------------------------------------------------------------------
// wide operator (D > RL_MAX_D outputs): Y[v][a][i] (+)= sum_b B[a][b] Z[v][b][i]
//   grid (ceil(m / 256), ceil(D / 8), nvec)   block 256
// A thread owns a grid point and 8 outputs a; the D values Z[v][.][i] stream through
// once per block of 8 outputs (B rows are scalar loads).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_wide_mix(const double* __restrict__ Z, const double* __restrict__ B, int D, int m,
           double* __restrict__ Y, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int a0 = blockIdx.y * 8, v = blockIdx.z;
    if (i >= m) return;
    const double* z = Z + (size_t)v * D * m + i;
    double acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0;
    for (int b = 0; b < D; ++b) {
        const double zb = z[(size_t)b * m];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int a = a0 + k < D ? a0 + k : D - 1;
            acc[k] = fma(B[(size_t)a * D + b], zb, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (a0 + k < D) {
            double* y = Y + ((size_t)v * D + a0 + k) * m + i;
            *y = accumulate ? *y + acc[k] : acc[k];
        }
}

static int wide_reserve(rl_gridop* g, int nvec) {
    const size_t need = (size_t)nvec * g->D * g->m;
    if (g->wide_Z_cap >= need) return RL_OK;
    if (g->wide_Z) RL_HIP(hipFree(g->wide_Z));
    g->wide_Z = nullptr;
    g->wide_Z_cap = 0;
    RL_HIP(hipMalloc((void**)&g->wide_Z, need * sizeof(double)));
    g->wide_Z_cap = need;
    return RL_OK;
}

// the operator: one single-top product of the child per top row, one mix pass each
static int wide_mvm(rl_gridop* g, const double* X, double* Y, int nvec, hipStream_t stream) {
    if (nvec < 0) return fail(RL_EINVAL, "nvec < 0");
    if (nvec == 0) return RL_OK;
    if (!X || !Y) return fail(RL_EINVAL, "X or Y is NULL");
    if (X == Y) return fail(RL_EINVAL, "X and Y may not alias");
    if (g->Q < 1) return fail(RL_EINVAL, "grid operator has no parameters yet");
    RL_HIP(hipSetDevice(g->device));
    if (g->wide_Z_cap < (size_t)nvec * g->D * g->m) {
        if (stream_capturing(stream))
            return fail(RL_EINVAL, "wide operator: workspace not reserved before a capture");
        RL_TRY(wide_reserve(g, nvec));
    }
    const int D = g->D, m = g->m;
    for (int q = 0; q < g->Q; ++q) {
        RL_TRY(rl_gridop_mvm_top(g->child, q, X, g->wide_Z, nvec * D, stream));
        RL_LAUNCH(k_wide_mix, dim3((m + 255) / 256, (D + 7) / 8, nvec), dim3(256), 0, stream,
                  (const double*)g->wide_Z, (const double*)(g->wide_B + (size_t)q * D * D), D, m, Y,
                  q > 0 ? 1 : 0);
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

bool stream_capturing(hipStream_t stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return stream != nullptr && hipStreamIsCapturing(stream, &cs) == hipSuccess &&
           cs != hipStreamCaptureStatusNone;
}

static int mvm_with_mix(rl_gridop* g, const MixParams& mp, const double* X, double* Y, int nvec,
                        hipStream_t stream) {
    if (!g) return fail(RL_EINVAL, "gridop is NULL");
    if (nvec < 0) return fail(RL_EINVAL, "nvec < 0");
    if (nvec == 0) return RL_OK;
    if (!X || !Y) return fail(RL_EINVAL, "X or Y is NULL");
    if (X == Y) return fail(RL_EINVAL, "X and Y may not alias");
    if (g->Q < 1) return fail(RL_EINVAL, "grid operator has no parameters yet");
    RL_HIP(hipSetDevice(g->device));
    const bool big = (g->lr_try || g->sf_try) && !g->lr_bypass &&
                     (size_t)nvec * g->D * g->m >= g->lr_min;
    // (nothing may be allocated or copied inside a capture: a pending verification
    // waits for the next product outside one; the solver runs it before it captures)
    const bool capturing = big && stream_capturing(stream);
    if (big && g->lr_dirty && !capturing) {
        // (the verification runs whole products of its own: never with a deferred expansion)
        const bool defer = g->defer_expand;
        g->defer_expand = false;
        const int rc = lr_ensure(g);
        g->defer_expand = defer;
        if (rc != RL_OK) return rc;
    }
    if (big && !g->lr_dirty) {
        // a batch large enough to fill the chip with projection / filter workgroups:
        // each top row in the form it was found in at set time (forms_setup)
        const bool single = mp.nfac == 0 && mp.Q == 1 && mp.kappa == g->ones;
        const int q1 = single ? (int)((mp.spec - g->spec) / g->L) : -1;
        const int form = single ? g->top_form[q1] : (g->lr_ok ? 1 : (g->st_ok ? 3 : 0));
        const bool poly_part = form == 1 || (form == 3 && g->lr_np > 0);
        bool ready = true;
        if (poly_part)
            ready = g->lr_part_cap >= lr_part_need(g, nvec) &&
                    g->lr_zhat_cap >= (size_t)nvec * g->D * RL_LR_RMAX;
        const int sNF = single ? 1 : g->sf_n, sfac = single ? 0 : g->sf_nfac;
        const int sNS = single ? (form == 2 ? g->sf_top_ns[q1] : 2) : g->sf_ns;
        if (form >= 2) ready = ready && sf_ready(g, nvec, sNF, sfac, sNS);
        if (form != 0 && !ready && !capturing) {
            if (poly_part) RL_TRY(lr_reserve(g, nvec));
            if (form >= 2) RL_TRY(sf_reserve(g, nvec, sNF, sfac, sNS));
            ready = true;
        }
        if (form != 0 && ready) {
            if (form == 1) {
                trace_once("grid product: polynomial-subspace form (k_lr_project / mix / expand)");
                if (single) return lr_apply(g, X, Y, nvec, q1, 1, g->lr_eye, stream);
                return lr_apply(g, X, Y, nvec, 0, g->Q, g->lr_B, stream);
            }
            if (form == 2) {
                trace_once("grid product: recursive-filter form (k_sf_carries / scan / apply)");
                return sf_apply_top(g, q1, X, Y, nvec, stream);
            }
            trace_once("grid product: recursive-filter part + polynomial part");
            RL_TRY(sf_apply_all(g, X, Y, nvec, stream));
            if (g->lr_np > 0) RL_TRY(lr_apply_compact(g, X, Y, nvec, stream));
            return RL_OK;
        }
    }
    // A small batch (below the gate) of an operator that IS wholly in the polynomial form: two
    // launches spread over the chip (k_lr_small_project / k_lr_small_expand, rl_lowrank.h).  The pending verification of
    // the form runs for such a batch too when the grid could take this path (outside captures;
    // back-off after rejections as for big batches); a handle whose gate was moved
    // (rl_gridop_set_form_gate) keeps to what the caller asked for.
    const bool small_try = !big && g->lr_try && !g->lr_bypass && !g->kn.no_lr_small &&
                           (size_t)g->D * g->m <= RL_LR_SMALL_MAX && g->lr_min == lr_min_elements(g) &&
                           !(mp.nfac == 0 && mp.Q == 1 && mp.kappa == g->ones);
    if (small_try && g->lr_dirty && !stream_capturing(stream)) RL_TRY(lr_ensure(g));
    if (small_try && !g->lr_dirty && g->lr_ok && g->lr_Mf_ok && nvec <= 65535) {
        const int nseg = lr_small_nseg(g->m);
        const size_t need = (size_t)nvec * g->D * nseg * RL_LR_RMAX;
        if (g->lr_spart_cap < need && !stream_capturing(stream)) {
            if (g->lr_spart) RL_HIP(hipFree(g->lr_spart));
            g->lr_spart = nullptr;
            g->lr_spart_cap = 0;
            RL_HIP(hipMalloc((void**)&g->lr_spart, need * sizeof(double)));
            g->lr_spart_cap = need;
        }
        if (g->lr_spart_cap >= need) {
            trace_once("grid product: polynomial-subspace form, small batch (k_lr_small_project / k_lr_small_expand)");
            const dim3 grid(g->D * nseg, nvec), blk(RL_LR_SMALL_WG);
#define RL_LR_SMALL(R_)                                                                           \
            RL_LAUNCH(k_lr_small_project<R_>, grid, blk, lr_small_project_lds(R_), stream, X, g->D, g->m, nseg, \
                      (const double*)g->lr_beta, g->lr_spart);                                     \
            RL_LAUNCH(k_lr_small_expand<R_>, grid, blk, lr_small_expand_lds(g->D, R_), stream,     \
                      (const double*)g->lr_spart, g->D, g->m, nseg, (const double*)g->lr_beta,     \
                      (const double*)g->lr_Mf, Y)
            switch (g->lr_r) {
                case 24: RL_LR_SMALL(24); break;
                case 32: RL_LR_SMALL(32); break;
                case 36: RL_LR_SMALL(36); break;
                case 40: RL_LR_SMALL(40); break;
                default: RL_LR_SMALL(48); break;
            }
#undef RL_LR_SMALL
            RL_HIP(hipGetLastError());
            return RL_OK;
        }
    }
    if (g->v1p && (nvec >= g->v1p_min || g->D <= 2)) {
        // short grid, enough pairs to fill the chip with one workgroup per pair
        // (measured: D=13, m=238: 21 vs 46 us at 256 vectors, 95 vs 228 us at 2048,
        // but 19.6 vs 16.5 us at 16 -- a few workgroups walking all D transforms
        // are slower than three launches spread over the chip): one kernel,
        // nothing through global memory
        trace_once("grid product: k1_product (single tile)");
        MixParams mp1 = mp;
        mp1.spec = g->spec1 + (mp.spec - g->spec);
        RL_TRY(launch1p(g, g->D, (unsigned)(((size_t)nvec + 1) / 2), stream, X, Y, nvec, 0, mp1,
                        nullptr));
        RL_HIP(hipGetLastError());
        return RL_OK;
    }
    const size_t total_pairs = ((size_t)nvec + 1) / 2;
    size_t chunk = std::min(total_pairs, g->chunk_pairs);
    RL_TRY(ensure_workspace(g, chunk));
    const size_t vec_len = (size_t)g->D * g->m;
    // two streams for a product of several chunks
    bool two = false;
    // (measured: pays when a pair's intermediates are large and a chunk holds
    // only a few pairs -- C5, 4.01 -> 3.77 ms per 129-vector product; hurts
    // when chunks hold hundreds of pairs -- C2, 2.51 -> 2.32 M MVM/s)
    const bool want_two = wants_two_streams(g);
    if (g->v2 && total_pairs > chunk && want_two) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = stream != nullptr &&
                               hipStreamIsCapturing(stream, &cs) == hipSuccess &&
                               cs != hipStreamCaptureStatusNone;
        // inside a capture nothing may be allocated: the solver prepares the
        // second workspace beforehand (prepare_two_streams) or stays on one
        if (!capturing) RL_TRY(prepare_two_streams(g, chunk));
        if (g->T2_pairs >= chunk && g->nside > 0) {
            RL_HIP(hipEventRecord(g->ev_fork, stream));
            for (int i = 0; i < g->nside; ++i)
                RL_HIP(hipStreamWaitEvent(g->aux[i], g->ev_fork, 0));
            two = true;
        }
    }
    int parity = 0;
    for (size_t p0 = 0; p0 < total_pairs; p0 += chunk) {
        const size_t pairs = std::min(chunk, total_pairs - p0);
        const int v0 = (int)(2 * p0);
        const int nv = std::min<int>(nvec - v0, (int)(2 * pairs));
        const double* Xc = X + (size_t)v0 * vec_len;
        double* Yc = Y + (size_t)v0 * vec_len;
        if (g->v2) {
            hipStream_t cst = stream;
            cplx* tb = g->T;
            if (two) {
                cst = parity ? g->aux[parity - 1] : stream;
                tb = parity ? g->T2[parity - 1] : g->T;
                parity = (parity + 1) % (g->nside + 1);
            }
            RL_TRY(mvm_chunk_v2(g, mp, Xc, Yc, nv, pairs, cst, nullptr, nullptr, tb));
            continue;
        }
        RL_TRY(mvm_chunk_v1(g, mp, Xc, Yc, nv, pairs, stream));
    }
    if (two) {
        for (int i = 0; i < g->nside; ++i) {
            RL_HIP(hipEventRecord(g->ev_join[i], g->aux[i]));
            RL_HIP(hipStreamWaitEvent(stream, g->ev_join[i], 0));
        }
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_gridop_mvm(rl_gridop* g, const double* X, double* Y, int nvec, void* stream) {
    if (!g) return fail(RL_EINVAL, "gridop is NULL");
    if (g->wide) return wide_mvm(g, X, Y, nvec, (hipStream_t)stream);
    MixParams mp{g->Q, g->nfac, g->spec, g->facA, g->facW, g->facQ, g->kappa,
                 g->mixtab_ok ? g->mixtab : nullptr,
                 g->mixtab_ok ? g->mixtab + (size_t)g->D * g->L : nullptr};
    return mvm_with_mix(g, mp, X, Y, nvec, (hipStream_t)stream);
}

extern "C" int rl_gridop_mvm_top(rl_gridop* g, int q, const double* X, double* Y, int nvec,
                                 void* stream) {
    if (!g) return fail(RL_EINVAL, "gridop is NULL");
    if (q < 0 || q >= g->Q) return fail(RL_EINVAL, "rl_gridop_mvm_top: q out of range");
    if (g->wide) {
        if (nvec < 0) return fail(RL_EINVAL, "nvec < 0");
        return rl_gridop_mvm_top(g->child, q, X, Y, nvec * g->D, stream);
    }
    MixParams mp{1, 0, g->spec + (size_t)q * g->L, nullptr, nullptr, nullptr, g->ones, nullptr,
                 nullptr};
    return mvm_with_mix(g, mp, X, Y, nvec, (hipStream_t)stream);
}

extern "C" int rl_gridop_spectrum_host(rl_gridop* g, int q, double* out) {
    if (!g || !out) return fail(RL_EINVAL, "NULL argument");
    if (g->wide) return rl_gridop_spectrum_host(g->child, q, out);
    if (q < 0 || q >= g->Q) return fail(RL_EINVAL, "rl_gridop_spectrum_host: q out of range");
    RL_HIP(hipSetDevice(g->device));
    std::vector<double> scr(g->L);
    RL_HIP(hipDeviceSynchronize());
    RL_HIP(hipMemcpy(scr.data(), g->spec + (size_t)q * g->L, (size_t)g->L * sizeof(double),
                     hipMemcpyDeviceToHost));
    for (int r = 0; r < g->N1; ++r)
        for (int c = 0; c < g->N2; ++c) {
            // 1-D: frequency k1 + N1 k2 of the length-L transform;
            // 2-D: entry (k1, k2) of the N1 x N2 transform, row-major
            const size_t k = g->geo.m1 ? (size_t)g->h_freq1[r] * g->N2 + g->h_freq2[c]
                                       : g->h_freq1[r] + (size_t)g->N1 * g->h_freq2[c];
            out[k] = scr[(size_t)r * g->N2 + c] * (double)g->L;
        }
    return RL_OK;
}


// everything a product of nvec vectors allocates lazily, before a graph capture: pending
// verification and buffers of the structured forms, intermediates of the transform kernels
// (second set for the two-stream chunks); a wide operator: its row buffer and its child
int gridop_prepare(rl_gridop* g, int nvec) {
    if (g->wide) {
        RL_TRY(wide_reserve(g, nvec));
        return gridop_prepare(g->child, nvec * g->D);
    }
    RL_TRY(lr_prepare(g, nvec));
    const size_t pairs = ((size_t)nvec + 1) / 2;
    RL_TRY(ensure_workspace(g, std::min(pairs, g->chunk_pairs)));
    if (g->v2 && pairs > g->chunk_pairs && wants_two_streams(g))
        RL_TRY(prepare_two_streams(g, g->chunk_pairs));
    return RL_OK;
}

extern "C" int rl_gridop_set_rank_hint(rl_gridop* g, int rank) {
    if (!g) return fail(RL_EINVAL, "rl_gridop_set_rank_hint: NULL handle");
    if (rank != 0 && rank != 24 && rank != 32 && rank != 36 && rank != 40 && rank != 48)
        return fail(RL_EINVAL, "rl_gridop_set_rank_hint: rank must be 0 or one of 24, 32, 36, 40, 48");
    if (g->wide) return rl_gridop_set_rank_hint(g->child, rank);
    g->lr_rank_hint = rank;
    return RL_OK;
}

extern "C" int rl_gridop_project(rl_gridop* g, const double* X, int nvec, int rank, double* out,
                                 void* stream) {
    if (!g || !X || !out) return fail(RL_EINVAL, "rl_gridop_project: NULL argument");
    if (nvec < 0) return fail(RL_EINVAL, "rl_gridop_project: nvec < 0");
    if (g->wide || !g->lr_try) return fail(RL_ELIMIT, "rl_gridop_project: this grid has no polynomial basis (1-D grids of >= 96 points, D <= 16)");
    if (rank != 24 && rank != 32 && rank != 36 && rank != 40 && rank != 48)
        return fail(RL_EINVAL, "rl_gridop_project: rank must be one of 24, 32, 36, 40, 48");
    if (nvec == 0) return RL_OK;
    RL_HIP(hipSetDevice(g->device));
    // (the basis comes with a handle's first verification of the polynomial form)
    if (g->Q >= 1) RL_TRY(lr_ensure(g));
    if (!g->lr_beta || !g->lr_nu)
        return fail(RL_ELIMIT, "rl_gridop_project: no verification of the polynomial form has run on this handle yet");
    hipStream_t st = (hipStream_t)stream;
    RL_TRY(lr_reserve(g, nvec));
    const int nrows = nvec * g->D;
    int chunks = 0;
    switch (rank) {
        case 24: chunks = lr_project<24>(g, X, nrows, st); break;
        case 32: chunks = lr_project<32>(g, X, nrows, st); break;
        case 36: chunks = lr_project<36>(g, X, nrows, st); break;
        case 40: chunks = lr_project<40>(g, X, nrows, st); break;
        default: chunks = lr_project<48>(g, X, nrows, st); break;
    }
    RL_LAUNCH(k_lr_coeffs, dim3(((size_t)nrows * rank + 255) / 256), dim3(256), 0, st,
              (const double*)g->lr_part, chunks, nrows, rank, (const double*)g->lr_nu, out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

extern "C" int rl_gridop_poly_coeffs(rl_gridop* g, int q, double* out, int cap, int* rank) {
    if (!g || !rank) return fail(RL_EINVAL, "rl_gridop_poly_coeffs: NULL argument");
    if (q < 0 || q >= g->Q) return fail(RL_EINVAL, "rl_gridop_poly_coeffs: q out of range");
    *rank = 0;
    if (g->wide || !g->lr_try) return RL_OK;
    RL_HIP(hipSetDevice(g->device));
    RL_TRY(lr_ensure(g));
    if (q >= (int)g->top_form.size() || g->top_form[q] != 1) return RL_OK;
    const int r = g->lr_r;
    if ((int)g->lr_hC.size() < (q + 1) * r * r) return RL_OK;
    if (out != nullptr) {
        if (cap < r * r) return fail(RL_EINVAL, "rl_gridop_poly_coeffs: out holds fewer than rank^2 values");
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < r; ++j)
                out[(size_t)i * r + j] = 0.5 * (g->lr_hC[((size_t)q * r + i) * r + j] +
                                                g->lr_hC[((size_t)q * r + j) * r + i]);
    }
    *rank = r;
    return RL_OK;
}

