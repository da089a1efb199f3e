// C ABI (include/plshts.h): plan management and the stream-ordered transform entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/plshts.h"
#include "device_plan.h"
#include "plshts_internal.h"
#include "ringfft.h"

namespace plshts {
int rings_per_group(int spin, const DevPlan &P);
void launch_prep0(const DevPlan &P, const double *alm, const double *fl, double *prep, hipStream_t st, int nb = 1);
int seed_family_rg(const DevPlan &P, int spin, int fam);
bool launch_seed_gen(const DevPlan &P, const DevSpinTab *S, int spin, int fam, int rg, int *il, double *st, int *sc, hipStream_t s);
bool launch_prep0_lowrank(const DevPlan &P, const double *alm, const double *fl, double *prep, hipStream_t st, int nb, int nmodes, const double *pm,
                          double *parts);
void launch_preps(const DevPlan &P, const DevSpinTab &S, int spin, const double *alm, const double *fl, double *prep, hipStream_t st, bool gonly);
void launch_synth0(const DevPlan &P, const double *prep, double *phase, hipStream_t st, int nb = 1);
bool synth0_pairs(const DevPlan &P, int nb);
void launch_synths(const DevPlan &P, const DevSpinTab &S, int spin, const double *prep, double *phase, hipStream_t st, bool gonly, int nb = 1);
void launch_synths_pair(const DevPlan &P, const DevSpinTab &S, int spin, const double *prep, const double *prep2, double *phase, hipStream_t st);
void launch_synths_gpair(const DevPlan &P, const DevSpinTab &S, int spin, const double *prep, const double *prep2, double *phase, hipStream_t st);
void launch_synths_batch2(const DevPlan &P, const DevSpinTab &S, int spin, const double *prep, const double *prep2, double *phase, hipStream_t st,
                          int npairs_b = 1);
void launch_anal0(const DevPlan &P, const double *phase, double *partial, const double *fl, double *alm, hipStream_t st,
                  const double *add = nullptr, const double *fl_add = nullptr, int nb = 1, int lr_nmodes = 0, int lr_nparts = 0, int lr_pstride = 0,
                  const double *lr_rm = nullptr, const double *lr_parts = nullptr, const PostDots *dots = nullptr, int64_t lr_bstride = 0);
int post_dots_count(const DevPlan &P);
void tproj_parts_layout(int64_t n, int *nparts, int *pstride);
int64_t tproj_parts_bstride();
void launch_anals(const DevPlan &P, const DevSpinTab &S, int spin, int64_t nent, const double *phase, double *partial,
                  const double *fl, double *alm, hipStream_t st);
void launch_anals_gc(const DevPlan &P, const DevSpinTab &S, int spin, int64_t nent, const double *phase, double *partial, const double *fl,
                     double *almG, double *almC, hipStream_t st, const double *addG, const double *addC, const double *flG, const double *flC,
                     int nb = 1, const PostDots *dots = nullptr);
void launch_preps_gc(const DevPlan &P, const DevSpinTab &S, int spin, const double *almG, const double *almC, const double *fl, double *prep,
                     hipStream_t st, int nb = 1);
void launch_almxfl(int lmax, const double *in, const double *fl, int nfl, double *out, hipStream_t st, int nb = 1);
void launch_alm_copy(int lmax_in, const double *in, int lmax_out, double *out, hipStream_t st, int nb = 1);
void launch_alm2cl(int lmax, const double *a, const double *b, double *cl, hipStream_t st);
void launch_axpy(int64_t n, double a, const double *x, const double *y, double *out, hipStream_t st);
void launch_alm_dot(int lmax, int lmin, const double *a, const double *b, int accumulate, double *parts, hipStream_t st, int nb = 1);
void launch_axpy_dev(int64_t n, const double *num, const double *den, double sign, const double *x, double *y, hipStream_t st, int nb = 1);
void launch_cg_fused(int nf, const int *lmax, int lmin, const double *const *a, const double *const *b1, const double *const *b2, double *parts1,
                     double *parts2, const double *den, double *const *y1, const double *const *x1, double sign1, double *const *y2,
                     const double *const *x2, double sign2, unsigned *bar, hipStream_t st, int nbatch = 1, const double *active = nullptr);
void launch_cg_axpy_pre(int nf, const int *lmax, int npre, const double *pre1, const double *pre2, const double *den, double *parts1, double *parts2,
                        double *const *y1, const double *const *x1, double sign1, double *const *y2, const double *const *x2, double sign2, hipStream_t st,
                        int nbatch, const double *active, int y1_assign);
void launch_alm_splice(int lmax_lo, const double *lo, int lmax_hi, const double *hi, int lsplit, double *out, hipStream_t st,
                       const double *fl_hi = nullptr, int nb = 1, const double *dot_q = nullptr, int dot_lmin = 0, double *dot_pre = nullptr);
int alm_splice_dot_count(int lmax_hi);
int gemv_split_dot_count(int nf, int lmax_lo, int lmax_hi);
void launch_template_project(int64_t n, int nmodes, double *t, const double *n_inv, const double *pm, const double *rm, double *parts, hipStream_t st,
                             int nb = 1, double *t_apply = nullptr, int phase = 0);
void launch_almxfl_add(int lmax, const double *a, const double *b, const double *fl, int nfl, double *out, hipStream_t st, int nb = 1);
void launch_gemv(int nrows, int ncols, int64_t lda, const double *A, const double *x, double *y, hipStream_t st);
void launch_gemv_split(int nf, int64_t lda, const double *A, const double *const *hi, const int *map, int lmax_lo, int lmax_hi, const double *const *fl_hi,
                       double *const *out, hipStream_t st, const double *const *dot_q = nullptr, int dot_lmin = 0, double *dot_pre = nullptr);
void launch_gemv_nb(int nrows, int ncols, int64_t lda, const double *A, int nb, const double *x, double *y, hipStream_t st);
void launch_copy_slim(const double *src, double *dst, int64_t ndoubles, int nblocks, hipStream_t st);
void launch_store_addresses(int n, const unsigned long long *vals, unsigned long long *dst, hipStream_t st);
void launch_map_mul(int64_t n, const double *a, const double *b, double *out, hipStream_t st);
void launch_map_qu_weight(int64_t n, double *q, double *u, const double *nqq, const double *nqu, const double *nuu, hipStream_t st, int nb = 1,
                          int64_t bstride = 0);
void launch_qe_lens_product(int64_t n, const double *tmap, const double *gt, const double *ct, const double *rep, const double *imp,
                            const double *g3, const double *c3, const double *g1, const double *c1, double *outr, double *outi, hipStream_t st);
void launch_map_cmul(int64_t n, const double *ar, const double *ai, double s1, const double *br, const double *bi, double s2,
                     double sign, double *outr, double *outi, int accumulate, hipStream_t st);
void launch_fma_peak(int mode, int iters, double *out, int nblk, hipStream_t st);
void launch_map_add_normal(int64_t n, const double *in, double *out, double sigma, uint64_t key, hipStream_t st);
void launch_alm_unit_phases(int lmax, double *out, uint64_t key, hipStream_t st);
void launch_alm_lincomb(int lmax, int nout, const int *nterm, const double *const *alm, const double *const *fl, double *const *out, hipStream_t st);
void launch_template_project_md(const DevPlan &P, int nb, double *t, const double *n_inv, int weighted, const double *pinv, double *scratch,
                                hipStream_t st);
void launch_phase_pack(const DevPlan &P, int ncomp, double *phase, double *buf, int pair0, int pstride, int mg0, int mgstride, bool unpack, hipStream_t st);
void launch_alm_keep_mgroups(int lmax, double *alm, int mg0, int mgstride, int nb, hipStream_t st);
int64_t map_pack_doubles(const DevPlan &P, int pair0, int pstride);
void launch_map_pack_rings(const DevPlan &P, int ncomp, double *map, double *buf, int pair0, int pstride, bool unpack, hipStream_t st);
}  // namespace plshts

using namespace plshts;

static thread_local std::string g_err;
static int fail(const std::string &msg)
{
    g_err = msg;
    return 1;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

struct pl_plan {
    int device = 0;
    DevPlan P{};
    DevFFT F{};
    FftStreams fs{};
    DevSpinTab S[kMaxSpin + 1]{};
    // published with a release store once S / nent / the seed tables of the spin are complete, read with an acquire load on the lock-free fast path
    std::atomic<bool> have_spin[kMaxSpin + 1] = {};
    int64_t nent[kMaxSpin + 1] = {0, 0, 0, 0};
    std::vector<void *> allocs;
    bool seed_tables = true;    // pl_plan_opts.seed_tables
    pl_plan *parent = nullptr;  // forked plan: geometry / recursion / FFT tables belong to (and are freed by) the parent
    int64_t bytes = 0;
    // workspaces (grown on demand)
    double *phase = nullptr; int64_t phase_cap = 0;
    double *prep = nullptr; int64_t prep_cap = 0;
    double *prep2 = nullptr; int64_t prep2_cap = 0;  // second input of pl_alm2map_pair
    double *wmap = nullptr; int64_t wmap_cap = 0;    // pl_cg_fwd_tt: the weighted map between the two transforms
    double *tparts = nullptr; int64_t tparts_cap = 0;  // pl_cg_fwd_tt: per ring pair partial sums of the template coefficients
    double *partial = nullptr; int64_t partial_cap = 0;
    double *h_alm = nullptr; int64_t h_alm_cap = 0;   // device staging for host-pointer calls
    double *h_map = nullptr; int64_t h_map_cap = 0;
    double *h_fl = nullptr;
    // pl_plan_arm_post_dots: scalar products wanted from the next pl_cg_fwd_* call on this plan (one shot)
    PostDots dots{}; int dots_nf = 0; bool dots_armed = false;
    // optional per-stage timing with HIP events on the caller's stream (pl_profile_*)
    bool profiling = false;
    struct Ev { int kind; hipEvent_t e0, e1; };
    std::vector<Ev> events;
};

enum { PK_LEG_SYNTH0 = 0, PK_LEG_SYNTHS, PK_LEG_ANAL0, PK_LEG_ANALS, PK_FFT_SYNTH, PK_FFT_ANAL, PK_LEG_SYNTHS_GRAD, PK_LEG_SYNTHS_PAIR, PK_LEG_SYNTHS_BATCH2, PK_LEG_SYNTH0_PAIR, PK_NKINDS };
static_assert(PK_NKINDS == PL_PROFILE_KINDS, "include/plshts.h PL_PROFILE_KINDS out of date");

struct ProfScope {
    pl_plan *p; hipStream_t st; hipEvent_t e0 = nullptr, e1 = nullptr; int kind;
    ProfScope(pl_plan *p_, int kind_, hipStream_t st_) : p(p_), st(st_), kind(kind_)
    {
        if (!p->profiling) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
        (void)hipEventRecord(e0, st);
    }
    ~ProfScope()
    {
        if (!e0) return;
        (void)hipEventRecord(e1, st);
        p->events.push_back({kind, e0, e1});
    }
};

template <typename T>
static int upload(pl_plan *p, const std::vector<T> &v, const T **out)
{
    void *d = nullptr;
    size_t nb = v.size() * sizeof(T);
    if (nb == 0) nb = sizeof(T);
    HIPCHK(hipMalloc(&d, nb));
    p->allocs.push_back(d);
    p->bytes += nb;
    if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T *>(d);
    return 0;
}

static int grow(pl_plan *p, double **buf, int64_t *cap, int64_t ndoubles)
{
    if (*cap >= ndoubles) return 0;
    // An outgrown workspace is retired, not freed: a captured HIP graph (qcinv.multigrid) may still replay kernels that
    // were recorded with its address.  Workspaces grow at most a few times (spin 0 -> spin s), so this stays small.
    if (*buf) p->allocs.push_back(*buf);
    *buf = nullptr; *cap = 0;
    // + 64: the Legendre kernels fetch coefficients 8 entries at a time and may read (never use) past the last m
    HIPCHK(hipMalloc(reinterpret_cast<void **>(buf), (ndoubles + 64) * sizeof(double)));
    *cap = ndoubles;
    p->bytes += ndoubles * 8;
    return 0;
}

// The spin tables of a plan are built on first use and shared with its forks, which other host threads may be launching on (two solvers
// of one process, filt_cinv.run_tp): building and publishing them is serialised by one process-wide lock (taken only until a plan has
// the tables of that spin, a few times per process).
static std::recursive_mutex g_spin_mutex;

// The seed tables of one spin (device_plan.h DevSeedTab; both kernel families), built on the device by the families' own phase A.
// A failed allocation leaves the plan without that table (the kernels then recurse from l = m): never an error.
static void build_seed_tables(pl_plan *p, int spin)
{
    if (!p->seed_tables || p->parent) return;
    const DevPlan &P = p->P;
    for (int fam = 0; fam < 2; ++fam) {
        DevSeedTab &T = spin == 0 ? (fam == 0 ? p->P.seed_syn0 : p->P.seed_ana0) : (fam == 0 ? p->S[spin].seed_syn : p->S[spin].seed_ana);
        T = DevSeedTab{};
        const int rg = seed_family_rg(P, spin, fam);
        const int ngroups = (P.npairs + rg - 1) / rg, npad = ngroups * rg, w = spin == 0 ? 1 : 2;
        const size_t n_il = (size_t)(P.mmax + 1) * ngroups, n_e = (size_t)(P.mmax + 1) * npad;
        int *il = nullptr, *sc = nullptr;
        double *st = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&il), n_il * sizeof(int)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&st), n_e * 2 * w * sizeof(double)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&sc), n_e * w * sizeof(int)) != hipSuccess) {
            (void)hipGetLastError();
            if (il) (void)hipFree(il);
            if (st) (void)hipFree(st);
            if (sc) (void)hipFree(sc);
            continue;
        }
        const DevPlan Pc = p->P;  // (by value: the table under construction is not part of what the generator reads)
        const bool ok = launch_seed_gen(Pc, spin == 0 ? nullptr : &p->S[spin], spin, fam, rg, il, st, sc, nullptr) && hipGetLastError() == hipSuccess &&
                        hipDeviceSynchronize() == hipSuccess;
        if (!ok) { (void)hipGetLastError(); (void)hipFree(il); (void)hipFree(st); (void)hipFree(sc); continue; }
        p->allocs.push_back(il); p->allocs.push_back(st); p->allocs.push_back(sc);
        p->bytes += (int64_t)(n_il * sizeof(int) + n_e * 2 * w * sizeof(double) + n_e * w * sizeof(int));
        T.il = il; T.st = st; T.sc = sc; T.rg = rg; T.npad = npad;
    }
}

static int ensure_spin_locked(pl_plan *p, int spin);

static int ensure_spin(pl_plan *p, int spin)
{
    if (spin < 1 || spin > kMaxSpin) return fail("spin must be 1, 2 or 3");
    if (p->have_spin[spin].load(std::memory_order_acquire)) return 0;
    std::lock_guard<std::recursive_mutex> lock(g_spin_mutex);
    try {
        return ensure_spin_locked(p, spin);
    } catch (const std::exception &e) {
        return fail(std::string("spin tables: ") + e.what());
    }
}

static int ensure_spin_locked(pl_plan *p, int spin)
{
    if (p->have_spin[spin].load(std::memory_order_acquire)) return 0;
    if (p->parent) {  // tables live in the parent
        if (ensure_spin_locked(p->parent, spin)) return 1;
        p->S[spin] = p->parent->S[spin];
        p->nent[spin] = p->parent->nent[spin];
        p->have_spin[spin].store(true, std::memory_order_release);
        return 0;
    }
    SpinTables t;
    build_spin_tables(spin, p->P.lmax, p->P.mmax, t);
    DevSpinTab &S = p->S[spin];
    t.ab.resize(t.ab.size() + 64, 0.0);  // 8-entry scalar loads may read past the last m (values unused)
    if (upload(p, t.off, &S.off) || upload(p, t.ab, &S.ab) || upload(p, t.beta, &S.beta) || upload(p, t.seedfac_n, &S.seedfac_n) ||
        upload(p, t.seedfac_p, &S.seedfac_p) || upload(p, t.psin, &S.psin) || upload(p, t.phalf, &S.phalf) ||
        upload(p, t.usecos_n, &S.usecos_n) || upload(p, t.usecos_p, &S.usecos_p))
        return 1;
    RingGeom g;
    build_geometry(p->P.nside, g);
    std::vector<int> mlim(g.npairs);
    for (int i = 0; i < g.npairs; ++i) mlim[i] = mlim_ring(p->P.lmax, spin, g.sth[i], g.cth[i]);
    for (int i = 1; i < g.npairs; ++i) mlim[i] = std::max(mlim[i], mlim[i - 1]);  // non-decreasing towards the equator (it is; k_posts relies on it: the ring groups an order keeps are the last ones)
    if (upload(p, mlim, &S.mlim)) return 1;
    S.gstart = nullptr;
    p->nent[spin] = t.off.back();
    build_seed_tables(p, spin);
    p->have_spin[spin].store(true, std::memory_order_release);
    return 0;
}

static inline int ncomp_of(int spin) { return spin == 0 ? 1 : 2; }
static inline const int *mlim_of(pl_plan *p, int spin) { return spin == 0 ? p->P.mlim0 : p->S[spin].mlim; }

extern "C" {

int pl_version(void) { return 1; }
const char *pl_last_error(void) { return g_err.c_str(); }

int pl_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return -1; }
    return n;
}

static int plan_create_impl(int nside, int lmax, int rank, int nranks, const pl_plan_opts *opts, pl_plan **out);

int pl_plan_create(int nside, int lmax, pl_plan **out) { return plan_create_impl(nside, lmax, 0, 1, nullptr, out); }

int pl_plan_create_opts(int nside, int lmax, int rank, int nranks, const pl_plan_opts *opts, pl_plan **out)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("pl_plan_create_opts: rank out of range");
    return plan_create_impl(nside, lmax, rank, nranks, opts, out);
}

// A plan for one of `nranks` shards of a single transform (north_star: "m-blocks shard across the GPUs"; SURVEY.md 8(e), secondary):
// its Legendre launches cover the m-groups rank, rank + nranks, ... (4 consecutive orders each), its ring-FFT launches the ring
// pairs rank, rank + nranks, ...  Between the two stages the ranks exchange phase slices (pl_phase_pack / pl_phase_unpack around
// one all-to-all, plancklens_amd/parallel.py); tables and workspaces are those of an ordinary plan.
int pl_plan_create_shard(int nside, int lmax, int rank, int nranks, pl_plan **out)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("pl_plan_create_shard: rank out of range");
    return plan_create_impl(nside, lmax, rank, nranks, nullptr, out);
}

static int plan_create_body(int nside, int lmax, int rank, int nranks, const pl_plan_opts &opts, pl_plan *&p, pl_plan **out);

// The tables of a plan are built on the host (std::vector, long double) before they are uploaded -- about 1 GB at nside 4096: a host
// allocation failure must come back as an error code like every other failure, not end the calling process (include/plshts.h).
static int plan_create_impl(int nside, int lmax, int rank, int nranks, const pl_plan_opts *opts_in, pl_plan **out)
{
    if (!out) return fail("null plan pointer");
    *out = nullptr;
    if (nside < 1 || nside > 8192) return fail("nside out of range [1, 8192]");
    if (lmax < 0 || lmax > 4 * nside) return fail("lmax out of range [0, 4 nside]");
    pl_plan_opts opts = {0, -1, -1, -1, -1, -1};
    if (opts_in) opts = *opts_in;
    pl_plan *p = nullptr;
    try {
        return plan_create_body(nside, lmax, rank, nranks, opts, p, out);
    } catch (const std::exception &e) {
        *out = nullptr;
        if (p) pl_plan_destroy(p);
        return fail(std::string("pl_plan_create: ") + e.what());
    }
}

static int plan_create_body(int nside, int lmax, int rank, int nranks, const pl_plan_opts &opts, pl_plan *&p, pl_plan **out)
{
    p = new pl_plan();
    if (hipGetDevice(&p->device) != hipSuccess) { delete p; p = nullptr; return fail("no HIP device (hipGetDevice failed)"); }
    DevPlan &P = p->P;
    P.nside = nside; P.lmax = lmax; P.mmax = lmax;
    P.npix = 12LL * nside * nside;
    P.nalm = (int64_t)(lmax + 1) * (lmax + 2) / 2;
    P.mstride = (lmax + 1 + 3) / 4 * 4;
    P.mg0 = rank; P.mgstride = nranks;
    RingGeom g;
    build_geometry(nside, g);
    P.npairs = g.npairs;
    int rc = upload(p, g.cth, &P.cth) || upload(p, g.sth, &P.sth) || upload(p, g.chalf, &P.chalf) || upload(p, g.shalf, &P.shalf) ||
             upload(p, g.phi0, &P.phi0) || upload(p, g.nphi, &P.nphi) || upload(p, g.ofs_n, &P.ofs_n) || upload(p, g.ofs_s, &P.ofs_s);
    Spin0Tables t0;
    build_spin0_tables(lmax, lmax, t0);
    t0.ab.resize(t0.ab.size() + 64, 0.0);  // 8-entry scalar loads may read past the last m (values unused)
    rc = rc || upload(p, t0.off, &P.off0) || upload(p, t0.ab, &P.ab0) || upload(p, t0.alpha, &P.alpha0) || upload(p, t0.eps, &P.eps0) ||
         upload(p, t0.seed, &P.seed0);
    P.nent0 = t0.off.back();
    p->nent[0] = P.nent0;
    std::vector<int> mlim(g.npairs);
    for (int i = 0; i < g.npairs; ++i) mlim[i] = mlim_ring(lmax, 0, g.sth[i], g.cth[i]);
    for (int i = 1; i < g.npairs; ++i) mlim[i] = std::max(mlim[i], mlim[i - 1]);  // non-decreasing towards the equator (it is; k_post0 relies on it: the ring groups an order keeps are the last ones)
    rc = rc || upload(p, mlim, &P.mlim0);
    if (rc) { pl_plan_destroy(p); p = nullptr; return 1; }

    // FFT tables: ring lengths 4 q, q = 1 .. nside.  Every ring pair is served either by a register-resident kernel of
    // transform size N = 256 << c (ringfft.hip; synthesis and analysis share the class lists and the band-limited
    // Bluestein tables) or, for the short polar rings, aliased rings and anything unusual, by the LDS-resident generic
    // kernel.  opts.fft_legacy sends every pair to the generic kernel.  (Plan options are arguments of pl_plan_create_opts, not
    // environment variables: two plans of one process differ only where the caller said so.)
    DevFFT &F = p->F;
    const bool all_legacy = opts.fft_legacy > 0;  // (negative = "the measured default" like every other field: off)
    std::vector<int> K2of(nside + 1, 0), MofA(nside + 1, 0), clsA(nside + 1, -1), splitA(nside + 1, 0);
    // smallest half-size for which a Bluestein ring is split into two half-size convolutions (opts.fft_split_min: 0 = never)
    const int split_min = opts.fft_split_min >= 0 ? opts.fft_split_min : 512;
    {
        std::vector<int> mlmax(nside + 1, 0);
        for (int i = 0; i < g.npairs; ++i) {
            const int q = g.nphi[i] / 4;
            for (int s = 0; s <= kMaxSpin; ++s) {
                int ml = mlim_ring(lmax, s, g.sth[i], g.cth[i]);
                if (ml > lmax) ml = lmax;
                if (ml > mlmax[q]) mlmax[q] = ml;
            }
        }
        auto cls_of = [](int N) { int c = -1; if (N >= 256 && N <= (256 << (kFftClasses - 1))) { c = 0; while ((256 << c) < N) ++c; } return c; };
        for (int q = 1; q <= nside; ++q) {
            const int K = (mlmax[q] + 3) / 4 + 1;  // sub-DFT bins c = k1 or k1 - q with 4 |c| <= mlim + 3
            K2of[q] = K;
            if (all_legacy) continue;
            // the ring's own sub-DFT length: no band limit needed, only one order per bin (mlim <= n / 2; the kernels know the bin n / 2 --
            // the belt of a grid with lmax = 2 nside, which every coarse level of the CG chains is)
            // (opts.fft_nyq_min = shortest sub-DFT routed this way, 0 = never: measured 17 % / 7 % faster stages (synthesis / analysis) at
            // q = 2048, even at q <= 512, a slower analysis at q = 1024 -- the generic kernel runs those belts as well)
            const int nyq_min = opts.fft_nyq_min >= 0 ? opts.fft_nyq_min : 2048;
            if ((q & (q - 1)) == 0) { if ((nyq_min > 0 && q >= nyq_min) ? mlmax[q] <= 2 * q : 2 * K + 1 < q) clsA[q] = cls_of(q); continue; }
            if (2 * K + 1 >= q) continue;                               // aliased ring (mlim >= n / 2 - 5): generic kernel
            int Na = 256;
            while (Na < q + 2 * K + 1) Na <<= 1;       // only the 2 K + 1 in-band bins are non-zero (synthesis) / needed (analysis)
            // Two half-size convolutions instead (ringfft.hip, SPLIT): output / input halves [0, Nh / 2) and [Nh / 2, q) with
            // Nh = Na / 2 -- 3 transforms of size Nh for 2 of size Na, and the only register route for Na = 2 * (largest class)
            const int Nh = Na / 2, hh = Nh / 2;
            if (split_min > 0 && Nh >= split_min && cls_of(Nh) >= 0 && q > hh && q <= Nh && hh + 2 * K + 1 <= Nh && q - hh + 2 * K + 1 <= Nh) {  // (q <= Nh: a thread owns the pixels tl + G j, j < 8)
                clsA[q] = cls_of(Nh); MofA[q] = Nh; splitA[q] = 1;
                continue;
            }
            clsA[q] = cls_of(Na);
            if (clsA[q] >= 0) MofA[q] = Na;
        }
    }
    {   // A plan whose register classes would hold only a few ring pairs (nside 256 at lmax = 2 nside: 25 of 512) runs every ring in the
        // generic kernel: the stage is then one launch without a fork / join of side streams, and the CG operators can take the whole
        // pixel-space part in one (k_ring_roundtrip).  opts.fft_min_fast = d: below 1 / d of the pairs (default 8; 0: never).
        const int min_fast = opts.fft_min_fast >= 0 ? opts.fft_min_fast : 8;
        // ... and so does every small grid (opts.fft_generic_nside, default 512): there a stage of ~10 class kernels is ~10 launch
        // latencies on three streams for microseconds of work each (round 5, tools/kernel_bench.py with plan options: nside 256 stages
        // 0.072 -> 0.017-0.031 ms, nside 512 0.099-0.108 -> 0.044-0.084 ms; at nside 1024 the classes win 1.5-2x)
        const int generic_nside = opts.fft_generic_nside >= 0 ? opts.fft_generic_nside : 512;
        int64_t nfast = 0;
        for (int i = 0; i < g.npairs; ++i) nfast += clsA[g.nphi[i] / 4] >= 0;
        if ((min_fast > 0 && nfast > 0 && nfast * min_fast < g.npairs) || (nfast > 0 && nside <= generic_nside))
            for (int q = 1; q <= nside; ++q) { clsA[q] = -1; MofA[q] = 0; splitA[q] = 0; }
    }
    // (A power-of-two ring length below nside is met by a single ring pair per hemisphere pair; it still gets the launch of its
    // direct class: left to the generic kernel those few long rings size its workgroups and LDS for every short polar ring.)
    std::vector<int> listA[kFftClasses], dirA[kFftClasses], splA[kFftClasses], legacyA;
    for (int i = g.npairs - 1; i >= 0; --i) {  // longest rings first
        if (i % nranks != rank) continue;  // shard plan: the ring pairs of this rank only (interleaved: every class stays balanced)
        const int q = g.nphi[i] / 4;
        const bool direct = (q & (q - 1)) == 0;
        if (clsA[q] >= 0) (direct ? dirA : (splitA[q] ? splA : listA))[clsA[q]].push_back(i); else legacyA.push_back(i);
    }
    std::vector<int> Mof(nside + 1, 0), qlist, qlistA;
    std::vector<int64_t> woff(nside + 1, 0), coff(nside + 1, 0), coffA(nside + 1, 0);
    int64_t nw = 0, nc = 0, ncA = 0;
    int Lmax = 1, M2max = 2;
    for (int q = 1; q <= nside; ++q) {
        const bool generic = clsA[q] < 0;  // this ring length runs in the generic kernel
        if ((q & (q - 1)) == 0) { if (generic && q > Lmax) Lmax = q; continue; }
        woff[q] = nw; nw += q;
        qlist.push_back(q);
        if (MofA[q]) { coffA[q] = ncA; ncA += (splitA[q] ? 2 : 1) * MofA[q]; if (MofA[q] > M2max) M2max = MofA[q]; qlistA.push_back(q); }
        if (!generic) continue;
        int M = 2;
        while (M < 2 * q - 1) M <<= 1;
        Mof[q] = M; coff[q] = nc; nc += M;
        if (M > Lmax) Lmax = M;
    }
    F.Lmax = Lmax;
    F.Mtw = Lmax < 2 ? 2 : Lmax;
    if (F.Mtw < M2max) F.Mtw = M2max;
    for (int c = 0; c < kFftClasses; ++c) if ((!listA[c].empty() || !dirA[c].empty() || !splA[c].empty()) && F.Mtw < (256 << c)) F.Mtw = 256 << c;
    if ((size_t)Lmax * 16 > 160 * 1024 - 256) { pl_plan_destroy(p); p = nullptr; return fail("ring FFT workspace exceeds the 160 KiB LDS of a CU"); }
    {   // LDS twiddle tables of the largest generic transform (radix-8 passes + one radix-4/2 tail), if they fit beside it
        int k = 0; while ((1 << k) < Lmax) ++k;
        const int rt = (k % 3 == 0) ? 8 : (k % 3 == 2 ? 4 : 2);
        int tot = (rt == 8 ? 3 : rt == 4 ? 2 : 1);
        for (int64_t L = (int64_t)rt * 8; L <= Lmax; L *= 8) tot += 3 * (int)(L / 8);
        F.twl_cap = ((size_t)(Lmax + tot) * 16 <= 160 * 1024 - 256) ? tot : 0;
    }
    double *tw = nullptr, *chirp = nullptr, *filt = nullptr, *filtA = nullptr;
    const int *qlist_dev = nullptr, *qlistA_dev = nullptr;
    auto dalloc = [&](double **ptr, int64_t nd) -> int {
        if (nd < 2) nd = 2;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(ptr), nd * sizeof(double)));
        p->allocs.push_back(*ptr);
        p->bytes += nd * 8;
        return 0;
    };
    std::vector<double> ringc((size_t)(nside + 1) * 8, 0.0);
    for (int q = 1; q <= nside; ++q) {
        if (clsA[q] < 0) continue;
        const long double n = 4.0L * q, N = 256 << clsA[q], G = N / 8, pi = 3.14159265358979323846264338327950288L;
        const long double ang[4] = {pi / n, 2 * pi * G / n, 4 * pi * G / n, 4 * pi * N / n};
        for (int k = 0; k < 4; ++k) { ringc[(size_t)q * 8 + 2 * k] = (double)cosl(ang[k]); ringc[(size_t)q * 8 + 2 * k + 1] = (double)sinl(ang[k]); }
    }
    const double *ringc_dev = nullptr;
    if (upload(p, ringc, &ringc_dev)) { pl_plan_destroy(p); p = nullptr; return 1; }
    F.ringc = reinterpret_cast<const double2 *>(ringc_dev);
    rc = dalloc(&tw, F.Mtw) || dalloc(&chirp, 2 * nw) || dalloc(&filt, 2 * nc) || dalloc(&filtA, 2 * ncA) ||
         upload(p, Mof, &F.Mof) || upload(p, woff, &F.woff) || upload(p, coff, &F.coff) || upload(p, K2of, &F.K2of) ||
         upload(p, qlist, &qlist_dev) || upload(p, qlistA, &qlistA_dev) ||
         upload(p, MofA, &F.A.Mof) || upload(p, coffA, &F.A.coff) || upload(p, splitA, &F.A.split) || upload(p, legacyA, &F.A.legacy_pairs);
    F.A.legacy_n = (int)legacyA.size();
    F.A.legacy_qmax = 1;
    for (int i : legacyA) if (g.nphi[i] / 4 > F.A.legacy_qmax) F.A.legacy_qmax = g.nphi[i] / 4;
    for (int c = 0; c < kFftClasses && !rc; ++c) {
        rc = upload(p, listA[c], &F.A.cls_pairs[c]) || upload(p, dirA[c], &F.A.dir_pairs[c]) || upload(p, splA[c], &F.A.split_pairs[c]);
        F.A.cls_n[c] = (int)listA[c].size();
        F.A.dir_n[c] = (int)dirA[c].size();
        F.A.split_n[c] = (int)splA[c].size();
    }
    if (rc) { pl_plan_destroy(p); p = nullptr; return 1; }
    F.tw = reinterpret_cast<const double2 *>(tw);
    F.chirp = reinterpret_cast<const double2 *>(chirp);
    F.filt = reinterpret_cast<const double2 *>(filt);
    F.A.filt = reinterpret_cast<const double2 *>(filtA);
    hipError_t e = launch_twiddles(tw, F.Mtw, nullptr);
    if (e == hipSuccess) e = launch_bluestein_setup(F, qlist_dev, (int)qlist.size(), chirp, filt, nullptr);
    if (e == hipSuccess) e = launch_bluestein_setup2(F, qlistA_dev, (int)qlistA.size(), M2max, filtA, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = fft_streams_create(p->fs);
    if (e != hipSuccess) { pl_plan_destroy(p); p = nullptr; return fail(std::string("FFT table setup: ") + hipGetErrorString(e)); }
    p->seed_tables = opts.seed_tables != 0;
    build_seed_tables(p, 0);
    *out = p;
    return 0;
}

int pl_plan_fork(pl_plan *parent, pl_plan **out)
{
    if (!parent || !out) return fail("null plan pointer");
    *out = nullptr;
    if (parent->parent) return fail("fork the original plan, not a fork");
    pl_plan *p = nullptr;
    try { p = new pl_plan(); } catch (const std::exception &e) { return fail(std::string("pl_plan_fork: ") + e.what()); }
    p->device = parent->device;
    p->P = parent->P;
    p->F = parent->F;
    p->parent = parent;
    p->nent[0] = parent->nent[0];
    hipError_t e = fft_streams_create(p->fs);
    if (e != hipSuccess) { delete p; return fail(std::string("fork: ") + hipGetErrorString(e)); }
    *out = p;
    return 0;
}

int pl_plan_destroy(pl_plan *p)
{
    if (!p) return 0;
    for (auto &ev : p->events) { (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1); }  // profiling records never read
    p->events.clear();
    fft_streams_destroy(p->fs);
    for (void *d : p->allocs) (void)hipFree(d);
    if (p->phase) (void)hipFree(p->phase);
    if (p->prep) (void)hipFree(p->prep);
    if (p->prep2) (void)hipFree(p->prep2);
    if (p->wmap) (void)hipFree(p->wmap);
    if (p->tparts) (void)hipFree(p->tparts);
    if (p->partial) (void)hipFree(p->partial);
    if (p->h_alm) (void)hipFree(p->h_alm);
    if (p->h_map) (void)hipFree(p->h_map);
    if (p->h_fl) (void)hipFree(p->h_fl);
    delete p;
    return 0;
}

// The sub-grid (ring pairs pair0, pair0 + pair_stride, ...; m-groups mg0, mg0 + mg_stride, ...) of a phase array of ncomp components,
// as one contiguous buffer [pair][component][m-group][4 orders][4 doubles] -- what one rank sends another between the Legendre and
// ring-FFT stages of a sharded transform -- and back.
int pl_phase_pack(pl_plan *p, int ncomp, const double *phase, double *buf, int pair0, int pair_stride, int mg0, int mg_stride, void *stream)
{
    if (!p || !phase || !buf || ncomp < 1 || pair_stride < 1 || mg_stride < 1 || pair0 < 0 || mg0 < 0) return fail("pl_phase_pack: bad arguments");
    launch_phase_pack(p->P, ncomp, const_cast<double *>(phase), buf, pair0, pair_stride, mg0, mg_stride, false, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_phase_unpack(pl_plan *p, int ncomp, double *phase, const double *buf, int pair0, int pair_stride, int mg0, int mg_stride, void *stream)
{
    if (!p || !phase || !buf || ncomp < 1 || pair_stride < 1 || mg_stride < 1 || pair0 < 0 || mg0 < 0) return fail("pl_phase_unpack: bad arguments");
    launch_phase_pack(p->P, ncomp, phase, const_cast<double *>(buf), pair0, pair_stride, mg0, mg_stride, true, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// The pixels of ring pairs pair0, pair0 + pair_stride, ... of an ncomp-component map as one buffer [component][pair: north ring, south
// ring] (pl_map_pack_doubles per component) and back: what a rank contributes to the all-gather that completes a sharded synthesis.
int64_t pl_map_pack_doubles(const pl_plan *p, int pair0, int pair_stride)
{
    if (!p || pair_stride < 1 || pair0 < 0) return 0;
    return map_pack_doubles(p->P, pair0, pair_stride);
}
int pl_map_pack_rings(pl_plan *p, int ncomp, const double *map, double *buf, int pair0, int pair_stride, void *stream)
{
    if (!p || !map || !buf || ncomp < 1 || pair_stride < 1 || pair0 < 0) return fail("pl_map_pack_rings: bad arguments");
    launch_map_pack_rings(p->P, ncomp, const_cast<double *>(map), buf, pair0, pair_stride, false, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}
int pl_map_unpack_rings(pl_plan *p, int ncomp, double *map, const double *buf, int pair0, int pair_stride, void *stream)
{
    if (!p || !map || !buf || ncomp < 1 || pair_stride < 1 || pair0 < 0) return fail("pl_map_unpack_rings: bad arguments");
    launch_map_pack_rings(p->P, ncomp, map, const_cast<double *>(buf), pair0, pair_stride, true, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// doubles of the pl_phase_pack buffer for that sub-grid
int64_t pl_phase_pack_doubles(const pl_plan *p, int ncomp, int pair0, int pair_stride, int mg0, int mg_stride)
{
    if (!p || pair_stride < 1 || mg_stride < 1) return 0;
    const int64_t np = p->P.npairs > pair0 ? (p->P.npairs - pair0 + pair_stride - 1) / pair_stride : 0;
    const int nmg_all = (p->P.mmax + 4) / 4;
    const int64_t nm = nmg_all > mg0 ? (nmg_all - mg0 + mg_stride - 1) / mg_stride : 0;
    return np * ncomp * nm * 16;
}

// alm (nb arrays back to back) with every entry outside the m-groups mg0, mg0 + mg_stride, ... set to zero: the share of one rank of
// a sharded analysis, ready for a sum over ranks (each entry is non-zero on exactly one rank: the sum is exact)
int pl_alm_keep_mgroups(int lmax, int nb, double *alm, int mg0, int mg_stride, void *stream)
{
    if (lmax < 0 || nb < 1 || !alm || mg0 < 0 || mg_stride < 1) return fail("pl_alm_keep_mgroups: bad arguments");
    launch_alm_keep_mgroups(lmax, alm, mg0, mg_stride, nb, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_profile_enable(pl_plan *p, int on)
{
    if (!p) return fail("null plan");
    p->profiling = on != 0;
    return 0;
}

int pl_profile_read(pl_plan *p, double *ms_sum, int64_t *counts)
{
    if (!p) return fail("null plan");
    for (int k = 0; k < PK_NKINDS; ++k) { ms_sum[k] = 0.0; counts[k] = 0; }
    hipError_t err = hipSuccess;
    for (auto &ev : p->events) {  // every event is destroyed, whatever happens to the earlier ones
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(ev.e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev.e0, ev.e1);
        if (e == hipSuccess) { ms_sum[ev.kind] += ms; counts[ev.kind] += 1; }
        else if (err == hipSuccess) err = e;
        (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1);
    }
    p->events.clear();
    if (err != hipSuccess) return fail(std::string("pl_profile_read: ") + hipGetErrorString(err));
    return 0;
}

int64_t pl_plan_npix(const pl_plan *p) { return p ? p->P.npix : 0; }
int64_t pl_plan_nalm(const pl_plan *p) { return p ? p->P.nalm : 0; }
int64_t pl_plan_bytes(const pl_plan *p) { return p ? p->bytes : 0; }
// Recursion steps the Legendre kernels of a family (fam 0: synthesis, 1: analysis) of this plan execute per launch, from the plan's seed
// table: sum over the live (m, ring group) waves of (steps of m - first step of the wave) x ring-pair slots of the wave (pruned and padded slots of
// a live wave run along).  Units: (l, ring pair) steps for spin >= 1, two-l steps for spin 0.  -1: no table (every launch recurses from l = m).
static int64_t executed_steps_impl(pl_plan *p, int spin, int fam, bool useful_only)
{
    if (!p || spin < 0 || spin > kMaxSpin || fam < 0 || fam > 1) return -1;
    if (spin > 0 && ensure_spin(p, spin)) return -1;
    const DevPlan &P = p->P;
    const DevSeedTab &T = spin == 0 ? (fam == 0 ? P.seed_syn0 : P.seed_ana0) : (fam == 0 ? p->S[spin].seed_syn : p->S[spin].seed_ana);
    if (T.rg <= 0) return -1;
    try {
        const int ngroups = T.npad / T.rg;
        std::vector<int> il((size_t)(P.mmax + 1) * ngroups), mlim(P.npairs);
        if (hipMemcpy(il.data(), T.il, il.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(mlim.data(), spin == 0 ? P.mlim0 : p->S[spin].mlim, mlim.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        int64_t steps = 0;
        for (int g = 0; g < ngroups; ++g) {
            const int ip0 = g * T.rg, ip1 = std::min(P.npairs, (g + 1) * T.rg);
            int gl = -1;  // largest order any ring of the group keeps
            for (int ip = ip0; ip < ip1; ++ip) gl = std::max(gl, mlim[ip]);
            for (int m = 0; m <= std::min(P.mmax, gl); ++m) {
                const int n = spin == 0 ? (P.lmax - m) / 2 + 1 : P.lmax - std::max(m, spin) + 1;
                if (n <= 0) continue;
                int64_t slots = T.rg;
                if (useful_only) {  // mlim is non-decreasing towards the equator: the rings of the group that keep order m are its last ones
                    slots = 0;
                    for (int ip = ip0; ip < ip1; ++ip) slots += mlim[ip] >= m;
                }
                steps += (int64_t)std::max(0, n - il[(size_t)m * ngroups + g]) * slots;
            }
        }
        return steps;
    } catch (const std::exception &) {
        return -1;
    }
}
int64_t pl_plan_executed_steps(pl_plan *p, int spin, int fam) { return executed_steps_impl(p, spin, fam, false); }
// The same count over the ring-pair slots that hold a ring which keeps the order (ip < npairs and m <= mlim[ip]): the steps whose results are
// used.  pl_plan_executed_steps - pl_plan_useful_steps = lanes of live waves that run along on pruned rings or padding.
int64_t pl_plan_useful_steps(pl_plan *p, int spin, int fam) { return executed_steps_impl(p, spin, fam, true); }
void *pl_plan_side_stream(const pl_plan *p, int i) { return (p && p->fs.ok && i >= 0 && i < FftStreams::kN) ? (void *)p->fs.s[i] : nullptr; }

int64_t pl_plan_phase_doubles(const pl_plan *p, int spin)
{
    if (!p) return 0;
    return (int64_t)p->P.npairs * p->P.mstride * 4 * ncomp_of(spin);
}

// lr_* (spin 0): the coefficient pass of a low-rank update on the same input (c = lr_pm alm, partial sums into lr_parts) asked to ride in the
// prologue launch; *lr_done tells whether it did (coarse grids) -- if not the caller launches it
static int legendre_synth_impl(pl_plan *p, int spin, const double *alm, const double *fl, double *phase, void *stream, bool gonly, int nb = 1,
                               int lr_nmodes = 0, const double *lr_pm = nullptr, double *lr_parts = nullptr, bool *lr_done = nullptr)
{
    if (!p) return fail("null plan");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (spin == 0) {  // nb > 1: nb consecutive alm arrays -> phase array of nb components (spin 0 only)
        if (grow(p, &p->prep, &p->prep_cap, p->P.nent0 * 4 * nb)) return 1;
        const bool fused = lr_done && lr_nmodes > 0 && launch_prep0_lowrank(p->P, alm, fl, p->prep, st, nb, lr_nmodes, lr_pm, lr_parts);
        if (lr_done) *lr_done = fused;
        if (!fused) launch_prep0(p->P, alm, fl, p->prep, st, nb);
        { ProfScope ps(p, synth0_pairs(p->P, nb) ? PK_LEG_SYNTH0_PAIR : PK_LEG_SYNTH0, st); launch_synth0(p->P, p->prep, phase, st, nb); }
    } else {
        if (ensure_spin(p, spin)) return 1;
        if (grow(p, &p->prep, &p->prep_cap, p->nent[spin] * 4)) return 1;
        launch_preps(p->P, p->S[spin], spin, alm, fl, p->prep, st, gonly);
        { ProfScope ps(p, gonly ? PK_LEG_SYNTHS_GRAD : PK_LEG_SYNTHS, st); launch_synths(p->P, p->S[spin], spin, p->prep, phase, st, gonly); }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_legendre_synth(pl_plan *p, int spin, const double *alm, const double *fl, double *phase, void *stream)
{
    return legendre_synth_impl(p, spin, alm, fl, phase, stream, false);
}

int pl_legendre_synth_grad(pl_plan *p, int spin, const double *almG, const double *fl, double *phase, void *stream)
{
    if (spin == 0) return fail("gradient-only synthesis is a spin >= 1 transform");
    return legendre_synth_impl(p, spin, almG, fl, phase, stream, true);
}

int pl_legendre_anal(pl_plan *p, int spin, const double *phase, double *alm, const double *fl, void *stream)
{
    if (!p) return fail("null plan");
    p->dots_armed = false;  // (a pending pl_plan_arm_post_dots is for the next pl_cg_fwd_* call only: any other analysis on the plan cancels it)
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int RG = rings_per_group(spin, p->P);
    const int ngroups = (p->P.npairs + RG - 1) / RG;
    if (spin == 0) {
        if (grow(p, &p->partial, &p->partial_cap, (int64_t)ngroups * p->P.nent0 * 4)) return 1;
        { ProfScope ps(p, PK_LEG_ANAL0, st); launch_anal0(p->P, phase, p->partial, fl, alm, st); }
    } else {
        if (ensure_spin(p, spin)) return 1;
        if (grow(p, &p->partial, &p->partial_cap, (int64_t)ngroups * p->nent[spin] * 4)) return 1;
        { ProfScope ps(p, PK_LEG_ANALS, st); launch_anals(p->P, p->S[spin], spin, p->nent[spin], phase, p->partial, fl, alm, st); }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_phase2map(pl_plan *p, int spin, const double *phase, double *map, void *stream)
{
    if (!p) return fail("null plan");
    if (spin && ensure_spin(p, spin)) return 1;
    ProfScope ps(p, PK_FFT_SYNTH, static_cast<hipStream_t>(stream));
    HIPCHK(launch_phase2map(p->P, p->F, p->fs, mlim_of(p, spin), ncomp_of(spin), phase, map, static_cast<hipStream_t>(stream)));
    return 0;
}

int pl_map2phase(pl_plan *p, int spin, const double *map, double *phase, void *stream)
{
    if (!p) return fail("null plan");
    if (spin && ensure_spin(p, spin)) return 1;
    ProfScope ps(p, PK_FFT_ANAL, static_cast<hipStream_t>(stream));
    HIPCHK(launch_map2phase(p->P, p->F, p->fs, mlim_of(p, spin), ncomp_of(spin), map, phase, static_cast<hipStream_t>(stream)));
    return 0;
}

// pl_map2alm on device arrays whose addresses are read from a table in DEVICE memory when the kernels run (ncomp entries: one per component, each
// an npix map anywhere in device memory).  A captured launch (HIP graph) can then be replayed on other inputs by rewriting the table -- no copy of the
// maps into fixed slots -- and the components of a spin transform need not be the rows of one array.  Same kernels, same arithmetic as pl_map2alm.
int pl_map2alm_ind(pl_plan *p, int spin, const double *const *maps_ind_dev, double *alm_dev, const double *fl_dev, void *stream)
{
    if (p) p->dots_armed = false;  // (see pl_legendre_anal)
    if (!p) return fail("null plan");
    if (spin < 0 || spin > kMaxSpin) return fail("spin must be 0..3");
    if (!maps_ind_dev || !alm_dev) return fail("pl_map2alm_ind: null pointer table / alm pointer");
    if (spin && ensure_spin(p, spin)) return 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (grow(p, &p->phase, &p->phase_cap, pl_plan_phase_doubles(p, spin))) return 1;
    {
        ProfScope ps(p, PK_FFT_ANAL, st);
        HIPCHK(launch_map2phase(p->P, p->F, p->fs, mlim_of(p, spin), ncomp_of(spin), nullptr, p->phase, st, nullptr, maps_ind_dev));
    }
    return pl_legendre_anal(p, spin, p->phase, alm_dev, fl_dev, stream);
}

static int stage_fl(pl_plan *p, const double *fl, int where, hipStream_t st, const double **fl_dev)
{
    *fl_dev = nullptr;
    if (!fl) return 0;
    if (where == PL_DEVICE) { *fl_dev = fl; return 0; }
    if (!p->h_fl) { HIPCHK(hipMalloc(reinterpret_cast<void **>(&p->h_fl), (p->P.lmax + 1) * sizeof(double))); p->bytes += (p->P.lmax + 1) * 8; }
    HIPCHK(hipMemcpyAsync(p->h_fl, fl, (p->P.lmax + 1) * sizeof(double), hipMemcpyHostToDevice, st));
    *fl_dev = p->h_fl;
    return 0;
}

static int alm2map_impl(pl_plan *p, int spin, const double *alm, double *map, const double *fl, int where, void *stream, bool gonly)
{
    if (!p) return fail("null plan");
    if (spin < 0 || spin > kMaxSpin) return fail("spin must be 0..3");
    if (gonly && spin == 0) return fail("gradient-only synthesis is a spin >= 1 transform");
    if (!alm || !map) return fail("null alm / map pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nc = ncomp_of(spin), nca = gonly ? 1 : nc;
    const double *alm_d = alm, *fl_d = nullptr;
    double *map_d = map;
    if (stage_fl(p, fl, where, st, &fl_d)) return 1;
    if (where == PL_HOST) {
        if (grow(p, &p->h_alm, &p->h_alm_cap, 2 * p->P.nalm * nc) || grow(p, &p->h_map, &p->h_map_cap, p->P.npix * nc)) return 1;
        HIPCHK(hipMemcpyAsync(p->h_alm, alm, 2 * p->P.nalm * nca * sizeof(double), hipMemcpyHostToDevice, st));
        alm_d = p->h_alm; map_d = p->h_map;
    }
    if (grow(p, &p->phase, &p->phase_cap, pl_plan_phase_doubles(p, spin))) return 1;
    if (legendre_synth_impl(p, spin, alm_d, fl_d, p->phase, stream, gonly)) return 1;
    if (pl_phase2map(p, spin, p->phase, map_d, stream)) return 1;
    if (where == PL_HOST) {
        HIPCHK(hipMemcpyAsync(map, map_d, p->P.npix * nc * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return 0;
}

int pl_alm2map(pl_plan *p, int spin, const double *alm, double *map, const double *fl, int where, void *stream)
{
    return alm2map_impl(p, spin, alm, map, fl, where, stream, false);
}

int pl_alm2map_grad(pl_plan *p, int spin, const double *almG, double *map, const double *fl, int where, void *stream)
{
    return alm2map_impl(p, spin, almG, map, fl, where, stream, true);
}

int pl_alm2map_pair(pl_plan *p, int spin, const double *alm_gc, const double *fl, const double *alm_g2, const double *fl2, double *maps4,
                    void *stream)
{
    if (!p) return fail("null plan");
    if (spin < 1 || spin > kMaxSpin) return fail("pl_alm2map_pair: spin must be 1..3");
    if (!alm_gc || !alm_g2 || !maps4) return fail("pl_alm2map_pair: null alm / map pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ensure_spin(p, spin)) return 1;
    if (grow(p, &p->prep, &p->prep_cap, p->nent[spin] * 4) || grow(p, &p->prep2, &p->prep2_cap, p->nent[spin] * 4) ||
        grow(p, &p->phase, &p->phase_cap, 2 * pl_plan_phase_doubles(p, spin)))
        return 1;
    launch_preps(p->P, p->S[spin], spin, alm_gc, fl, p->prep, st, false);
    launch_preps(p->P, p->S[spin], spin, alm_g2, fl2, p->prep2, st, true);
    { ProfScope ps(p, PK_LEG_SYNTHS_PAIR, st); launch_synths_pair(p->P, p->S[spin], spin, p->prep, p->prep2, p->phase, st); }
    HIPCHK(hipGetLastError());
    {
        ProfScope ps(p, PK_FFT_SYNTH, st);
        HIPCHK(launch_phase2map(p->P, p->F, p->fs, mlim_of(p, spin), 4, p->phase, maps4, st));
    }
    return 0;
}

int pl_alm2map_grad_pair(pl_plan *p, int spin, const double *alm_g1, const double *fl1, const double *alm_g2, const double *fl2, double *maps4, void *stream)
{
    if (!p) return fail("null plan");
    if (spin < 1 || spin > kMaxSpin) return fail("pl_alm2map_grad_pair: spin must be 1..3");
    if (!alm_g1 || !alm_g2 || !maps4) return fail("pl_alm2map_grad_pair: null alm / map pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ensure_spin(p, spin)) return 1;
    if (grow(p, &p->prep, &p->prep_cap, p->nent[spin] * 4) || grow(p, &p->prep2, &p->prep2_cap, p->nent[spin] * 4) ||
        grow(p, &p->phase, &p->phase_cap, 2 * pl_plan_phase_doubles(p, spin)))
        return 1;
    launch_preps(p->P, p->S[spin], spin, alm_g1, fl1, p->prep, st, true);
    launch_preps(p->P, p->S[spin], spin, alm_g2, fl2, p->prep2, st, true);
    { ProfScope ps(p, PK_LEG_SYNTHS_GRAD, st); launch_synths_gpair(p->P, p->S[spin], spin, p->prep, p->prep2, p->phase, st); }
    HIPCHK(hipGetLastError());
    {
        ProfScope ps(p, PK_FFT_SYNTH, st);
        HIPCHK(launch_phase2map(p->P, p->F, p->fs, mlim_of(p, spin), 4, p->phase, maps4, st));
    }
    return 0;
}

int pl_alm2map_batch2(pl_plan *p, int spin, const double *alm_gc_1, const double *alm_gc_2, const double *fl, double *maps4, void *stream)
{
    if (!p) return fail("null plan");
    if (spin < 0 || spin > kMaxSpin) return fail("pl_alm2map_batch2: spin must be 0..3");
    if (!alm_gc_1 || !alm_gc_2 || !maps4) return fail("pl_alm2map_batch2: null alm / map pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (spin == 0) {  // two scalar inputs (one alm array each) -> the two rows of maps4; fine grids: one recursion for both (k_leg_synth0<R, true>)
        if (grow(p, &p->prep, &p->prep_cap, p->P.nent0 * 4 * 2) || grow(p, &p->phase, &p->phase_cap, 2 * pl_plan_phase_doubles(p, 0))) return 1;
        launch_prep0(p->P, alm_gc_1, fl, p->prep, st, 1);
        launch_prep0(p->P, alm_gc_2, fl, p->prep + p->P.nent0 * 4, st, 1);
        { ProfScope ps(p, synth0_pairs(p->P, 2) ? PK_LEG_SYNTH0_PAIR : PK_LEG_SYNTH0, st); launch_synth0(p->P, p->prep, p->phase, st, 2); }
        HIPCHK(hipGetLastError());
        {
            ProfScope ps(p, PK_FFT_SYNTH, st);
            HIPCHK(launch_phase2map(p->P, p->F, p->fs, mlim_of(p, 0), 2, p->phase, maps4, st));
        }
        return 0;
    }
    if (ensure_spin(p, spin)) return 1;
    if (grow(p, &p->prep, &p->prep_cap, p->nent[spin] * 4) || grow(p, &p->prep2, &p->prep2_cap, p->nent[spin] * 4) ||
        grow(p, &p->phase, &p->phase_cap, 2 * pl_plan_phase_doubles(p, spin)))
        return 1;
    launch_preps(p->P, p->S[spin], spin, alm_gc_1, fl, p->prep, st, false);
    launch_preps(p->P, p->S[spin], spin, alm_gc_2, fl, p->prep2, st, false);
    { ProfScope ps(p, PK_LEG_SYNTHS_BATCH2, st); launch_synths_batch2(p->P, p->S[spin], spin, p->prep, p->prep2, p->phase, st); }
    HIPCHK(hipGetLastError());
    {
        ProfScope ps(p, PK_FFT_SYNTH, st);
        HIPCHK(launch_phase2map(p->P, p->F, p->fs, mlim_of(p, spin), 4, p->phase, maps4, st));
    }
    return 0;
}

int pl_map2alm(pl_plan *p, int spin, const double *map, double *alm, const double *fl, int where, void *stream)
{
    if (p) p->dots_armed = false;  // (see pl_legendre_anal)
    if (!p) return fail("null plan");
    if (spin < 0 || spin > kMaxSpin) return fail("spin must be 0..3");
    if (!alm || !map) return fail("null alm / map pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nc = ncomp_of(spin);
    const double *map_d = map, *fl_d = nullptr;
    double *alm_d = alm;
    if (stage_fl(p, fl, where, st, &fl_d)) return 1;
    if (where == PL_HOST) {
        if (grow(p, &p->h_alm, &p->h_alm_cap, 2 * p->P.nalm * nc) || grow(p, &p->h_map, &p->h_map_cap, p->P.npix * nc)) return 1;
        HIPCHK(hipMemcpyAsync(p->h_map, map, p->P.npix * nc * sizeof(double), hipMemcpyHostToDevice, st));
        map_d = p->h_map; alm_d = p->h_alm;
    }
    if (grow(p, &p->phase, &p->phase_cap, pl_plan_phase_doubles(p, spin))) return 1;
    if (pl_map2phase(p, spin, map_d, p->phase, stream)) return 1;
    if (pl_legendre_anal(p, spin, p->phase, alm_d, fl_d, stream)) return 1;
    if (where == PL_HOST) {
        HIPCHK(hipMemcpyAsync(alm, alm_d, 2 * p->P.nalm * nc * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return 0;
}

int pl_almxfl(int lmax, const double *alm_in, const double *fl, int nfl, double *alm_out, void *stream)
{
    launch_almxfl(lmax, alm_in, fl, nfl, alm_out, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_alm2cl(int lmax, const double *a, const double *b, double *cl, void *stream)
{
    launch_alm2cl(lmax, a, b ? b : a, cl, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_alm_copy(int lmax_in, const double *in, int lmax_out, double *out, void *stream)
{
    launch_alm_copy(lmax_in, in, lmax_out, out, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- batched forms: every alm / map argument holds nb arrays back to back; filters, noise maps and template matrices are shared ----
#define PL_NB_CHECK(name) if (nb < 1 || nb > PL_MAX_BATCH) return fail(name ": batch count out of range [1, PL_MAX_BATCH]")

int pl_almxfl_b(int lmax, int nb, const double *alm_in, const double *fl, int nfl, double *alm_out, void *stream)
{
    PL_NB_CHECK("pl_almxfl_b");
    launch_almxfl(lmax, alm_in, fl, nfl, alm_out, static_cast<hipStream_t>(stream), nb);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_alm_copy_b(int lmax_in, int nb, const double *in, int lmax_out, double *out, void *stream)
{
    PL_NB_CHECK("pl_alm_copy_b");
    launch_alm_copy(lmax_in, in, lmax_out, out, static_cast<hipStream_t>(stream), nb);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_alm_splice_b(int lmax_lo, int nb, const double *alm_lo, int lmax_hi, const double *alm_hi, const double *fl_hi, int lsplit, double *out, void *stream)
{
    PL_NB_CHECK("pl_alm_splice_b");
    if (lsplit > lmax_lo || lsplit > lmax_hi || lmax_hi < 0) return fail("pl_alm_splice_b: lsplit exceeds a band-limit");
    launch_alm_splice(lmax_lo, alm_lo, lmax_hi, alm_hi, lsplit, out, static_cast<hipStream_t>(stream), fl_hi, nb);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_almxfl_add_b(int lmax, int nb, const double *a, const double *b, const double *fl, int nfl, double *out, void *stream)
{
    PL_NB_CHECK("pl_almxfl_add_b");
    if (lmax < 0 || !a || !b || !fl || !out) return fail("pl_almxfl_add_b: bad arguments");
    launch_almxfl_add(lmax, a, b, fl, nfl, out, static_cast<hipStream_t>(stream), nb);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_alm_dot_b(int lmax, int lmin, int nb, const double *a, const double *b, int accumulate, double *parts_dev, void *stream)
{
    PL_NB_CHECK("pl_alm_dot_b");
    if (lmax < 0 || !a || !b || !parts_dev) return fail("pl_alm_dot_b: bad arguments");
    launch_alm_dot(lmax, lmin < 0 ? 0 : lmin, a, b, accumulate, parts_dev, static_cast<hipStream_t>(stream), nb);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_axpy_dev_b(int64_t n, int nb, const double *num_parts_dev, const double *den_parts_dev, double sign, const double *x, double *y, void *stream)
{
    PL_NB_CHECK("pl_axpy_dev_b");
    if (n < 0 || !num_parts_dev || !x || !y) return fail("pl_axpy_dev_b: bad arguments");
    if (n == 0) return 0;
    launch_axpy_dev(n, num_parts_dev, den_parts_dev, sign, x, y, static_cast<hipStream_t>(stream), nb);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_template_project_b(int64_t npix, int nmodes, int nb, double *tmap, const double *n_inv, const double *pmat, const double *rmat, double *scratch,
                          void *stream)
{
    PL_NB_CHECK("pl_template_project_b");
    if (npix <= 0 || nmodes < 1 || nmodes > PL_TEMPLATE_MAX_MODES || !tmap || !n_inv || !pmat || !rmat || !scratch)
        return fail("pl_template_project_b: bad arguments (1 <= nmodes <= PL_TEMPLATE_MAX_MODES)");
    launch_template_project(npix, nmodes, tmap, n_inv, pmat, rmat, scratch, static_cast<hipStream_t>(stream), nb);
    HIPCHK(hipGetLastError());
    return 0;
}

// y_b -= rmat^t (pmat x_b) for nb vectors of n doubles: a rank-nmodes update with the coefficient pass on one vector and the subtraction on
// another.  It is how the CG operators project templates out in harmonic space: with V = B^t Y^t N^-1 T (one alm per template mode),
//   B^t Y^t [N^-1 - N^-1 T (T^t N^-1 T)^-1 T^t N^-1] Y B x  =  B^t Y^t N^-1 Y B x  -  V (T^t N^-1 T)^-1 V^t x
// (plancklens/qcinv/opfilt_tt.py:196-205 applies the bracket in pixel space), pmat = V with the weights of the real scalar product folded in,
// rmat = (T^t N^-1 T)^-1 V, both as real (nmodes, 2 nalm) matrices.  The kernels are those of pl_template_project (fixed reduction trees).
int pl_lowrank_update_b(int64_t n, int nmodes, int nb, const double *x, double *y, const double *pmat, const double *rmat, double *scratch, void *stream)
{
    PL_NB_CHECK("pl_lowrank_update_b");
    if (n <= 0 || nmodes < 1 || nmodes > PL_TEMPLATE_MAX_MODES || !x || !y || !pmat || !rmat || !scratch)
        return fail("pl_lowrank_update_b: bad arguments (1 <= nmodes <= PL_TEMPLATE_MAX_MODES)");
    launch_template_project(n, nmodes, const_cast<double *>(x), nullptr, pmat, rmat, scratch, static_cast<hipStream_t>(stream), nb, y);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_gemv_b(int nrows, int ncols, int64_t lda, const double *A, int nb, const double *x, double *y, void *stream)
{
    PL_NB_CHECK("pl_gemv_b");
    if (nrows < 0 || ncols < 0 || lda < ncols || !A || !x || !y) return fail("pl_gemv_b: bad arguments");
    if (nrows == 0) return 0;
    launch_gemv_nb(nrows, ncols, lda, A, nb, x, y, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_axpy(int64_t n, double a, const double *x, const double *y, double *out, void *stream)
{
    launch_axpy(n, a, x, y, out, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_alm_dot(int lmax, int lmin, const double *a, const double *b, int accumulate, double *parts_dev, void *stream)
{
    if (lmax < 0 || !a || !b || !parts_dev) return fail("pl_alm_dot: bad arguments");
    launch_alm_dot(lmax, lmin < 0 ? 0 : lmin, a, b, accumulate, parts_dev, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_axpy_dev(int64_t n, const double *num_parts_dev, const double *den_parts_dev, double sign, const double *x, double *y, void *stream)
{
    if (n < 0 || !num_parts_dev || !x || !y) return fail("pl_axpy_dev: bad arguments");
    if (n == 0) return 0;
    launch_axpy_dev(n, num_parts_dev, den_parts_dev, sign, x, y, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

static int cg_dot_axpy_impl(int nb, int nf, const int *lmax, int lmin, const double *const *a, const double *const *b1, const double *const *b2,
                            double *parts1_dev, double *parts2_dev, const double *den_parts_dev, double *const *y1, const double *const *x1, double sign1,
                            double *const *y2, const double *const *x2, double sign2, unsigned *barrier_dev, const double *active_dev, void *stream)
{
    if (nf < 1 || nf > 3 || !lmax || !a || !b1 || !parts1_dev || !y1 || !x1) return fail("pl_cg_dot_axpy: bad arguments");
    if (!b2 && !den_parts_dev) return fail("pl_cg_dot_axpy: either a second scalar product (b2) or a denominator (den_parts_dev) is needed");
    if (b2 && (!parts2_dev || den_parts_dev)) return fail("pl_cg_dot_axpy: b2 needs parts2_dev and excludes den_parts_dev");
    if ((y2 == nullptr) != (x2 == nullptr)) return fail("pl_cg_dot_axpy: y2 and x2 come together");
    if ((sign1 != 1.0 && sign1 != -1.0) || (sign2 != 1.0 && sign2 != -1.0)) return fail("pl_cg_dot_axpy: signs are +1 or -1");
    for (int k = 0; k < nf; ++k)
        if (lmax[k] < 0 || !a[k] || !b1[k] || (b2 && !b2[k]) || !y1[k] || !x1[k] || (y2 && (!y2[k] || !x2[k]))) return fail("pl_cg_dot_axpy: null field");
    launch_cg_fused(nf, lmax, lmin < 0 ? 0 : lmin, a, b1, b2, parts1_dev, parts2_dev, den_parts_dev, y1, x1, sign1, y2, x2, sign2, barrier_dev,
                    static_cast<hipStream_t>(stream), nb, active_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_cg_dot_axpy(int nf, const int *lmax, int lmin, const double *const *a, const double *const *b1, const double *const *b2,
                   double *parts1_dev, double *parts2_dev, const double *den_parts_dev, double *const *y1, const double *const *x1, double sign1,
                   double *const *y2, const double *const *x2, double sign2, unsigned *barrier_dev, void *stream)
{
    return cg_dot_axpy_impl(1, nf, lmax, lmin, a, b1, b2, parts1_dev, parts2_dev, den_parts_dev, y1, x1, sign1, y2, x2, sign2, barrier_dev, nullptr, stream);
}

int pl_cg_dot_axpy_b(int nb, int nf, const int *lmax, int lmin, const double *const *a, const double *const *b1, const double *const *b2,
                     double *parts1_dev, double *parts2_dev, const double *den_parts_dev, double *const *y1, const double *const *x1, double sign1,
                     double *const *y2, const double *const *x2, double sign2, const double *active_dev, void *stream)
{
    PL_NB_CHECK("pl_cg_dot_axpy_b");
    return cg_dot_axpy_impl(nb, nf, lmax, lmin, a, b1, b2, parts1_dev, parts2_dev, den_parts_dev, y1, x1, sign1, y2, x2, sign2, nullptr, active_dev, stream);
}

int pl_post_dots_count(pl_plan *p)
{
    if (!p) return -1;
    return post_dots_count(p->P);
}

// One shot: the next pl_cg_fwd_tt* (nf = 1) / pl_cg_fwd_pp* (nf = 2) call on this plan also leaves, per batch entry, pl_post_dots_count(p) partial sums
// of <d, q> in pre1 and of <d, r> in pre2 (q its result; CG weights, entries l < lmin excluded) -- formed by the kernel that writes q.
int pl_plan_arm_post_dots(pl_plan *p, int nf, const double *const *d, const double *const *r, int lmin, double *pre1, double *pre2)
{
    if (!p) return fail("null plan");
    if ((nf != 1 && nf != 2) || !d || !r || !pre1 || !pre2) return fail("pl_plan_arm_post_dots: bad arguments");
    PostDots D;
    for (int k = 0; k < nf; ++k) {
        if (!d[k] || !r[k]) return fail("pl_plan_arm_post_dots: null field");
        D.d[k] = d[k]; D.r[k] = r[k];
    }
    D.s1 = pre1; D.s2 = pre2; D.lmin = lmin < 0 ? 0 : lmin;
    p->dots = D; p->dots_nf = nf; p->dots_armed = true;
    return 0;
}

// The vector updates of pl_cg_dot_axpy_b from scalar products that arrive as npre partial sums per batch entry (pl_plan_arm_post_dots, pl_gemv_split_dot,
// pl_alm_splice_dot): den given: y1 += sign1 sum(pre1) / sum(den) x1; else y1 += sign1 sum(pre2) / sum(pre1) x1 and (optional) y2 += sign2 (same) x2.
// parts1 / parts2 (optional) receive the totals as PL_DOT_PARTS-entry partial sums.  y1_assign: y1 = ... instead of y1 += ... (y1 is not read).
int pl_cg_axpy_pre_b(int nb, int nf, const int *lmax, int npre, const double *pre1, const double *pre2, const double *den_parts_dev, double *parts1_dev,
                     double *parts2_dev, double *const *y1, const double *const *x1, double sign1, double *const *y2, const double *const *x2, double sign2,
                     const double *active_dev, int y1_assign, void *stream)
{
    PL_NB_CHECK("pl_cg_axpy_pre_b");
    if (nf < 1 || nf > 3 || !lmax || npre < 1 || !pre1 || !y1 || !x1) return fail("pl_cg_axpy_pre_b: bad arguments");
    if (!pre2 && !den_parts_dev) return fail("pl_cg_axpy_pre_b: either a second scalar product (pre2) or a denominator (den_parts_dev) is needed");
    if (pre2 && den_parts_dev) return fail("pl_cg_axpy_pre_b: pre2 excludes den_parts_dev");
    if ((y2 == nullptr) != (x2 == nullptr)) return fail("pl_cg_axpy_pre_b: y2 and x2 come together");
    if ((sign1 != 1.0 && sign1 != -1.0) || (sign2 != 1.0 && sign2 != -1.0)) return fail("pl_cg_axpy_pre_b: signs are +1 or -1");
    for (int k = 0; k < nf; ++k)
        if (lmax[k] < 0 || !y1[k] || !x1[k] || (y2 && (!y2[k] || !x2[k]))) return fail("pl_cg_axpy_pre_b: null field");
    launch_cg_axpy_pre(nf, lmax, npre, pre1, pre2, den_parts_dev, parts1_dev, parts2_dev, y1, x1, sign1, y2, x2, sign2, static_cast<hipStream_t>(stream), nb,
                       active_dev, y1_assign);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_template_project(int64_t npix, int nmodes, double *tmap, const double *n_inv, const double *pmat, const double *rmat, double *scratch, void *stream)
{
    if (npix <= 0 || nmodes < 1 || nmodes > PL_TEMPLATE_MAX_MODES || !tmap || !n_inv || !pmat || !rmat || !scratch)
        return fail("pl_template_project: bad arguments (1 <= nmodes <= PL_TEMPLATE_MAX_MODES)");
    launch_template_project(npix, nmodes, tmap, n_inv, pmat, rmat, scratch, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// The temperature CG operator in one call (plancklens/qcinv/opfilt_tt.py:54-73 with :196-205 inside):
//   alm_out = fl_out * Y^t [N^-1 - N^-1 P (P^t N^-1 P)^-1 P^t N^-1] Y (fl_in * alm_in)  +  fl_add * alm_add.
// On grids whose rings all run in the generic ring-FFT kernel (the coarse levels of the multigrid chain) the weighting and the projection
// ride in the two FFT launches (NinvProj); on the finer grids they are the two pl_template_project launches between the transforms.
static bool cg_roundtrip_enabled()
{
    static const bool on = dbg_env_int("PLSHTS_CG_ROUNDTRIP", 1) != 0;  // (PLSHTS_DEBUG=1 PLSHTS_CG_ROUNDTRIP=0: the two-launch form, tests)
    return on;
}

// (PLSHTS_DEBUG=1 PLSHTS_LR_PROLOGUE=0: the coefficient pass of a low-rank update always as its own launch on a side stream, tests)
static bool lr_prologue_enabled()
{
    static const bool on = dbg_env_int("PLSHTS_LR_PROLOGUE", 1) != 0;
    return on;
}
// launch shape of the coefficient pass (elementwise.hip / tproj_device.h: 1024-thread workgroups on fine grids, 256-thread ones below)
static void tproj_coeffs_shape_host(int64_t n, int *nt, int *nparts)
{
    int np = 0, ps = 0;
    tproj_parts_layout(n, &np, &ps);
    *nparts = np;
    *nt = n >= (int64_t)ps * 4096 ? 1024 : 256;
}

struct LowRank { int nmodes = 0; const double *pm = nullptr, *rm = nullptr; double *scratch = nullptr; };

static int cg_fwd_tt_impl(pl_plan *p, int nb, const double *alm_in, const double *fl_in, const double *n_inv, int nmodes, const double *pmat,
                          const double *rmat, double *scratch, const double *alm_add, const double *fl_add, double *alm_out, const double *fl_out,
                          void *stream, const double *pinv_md = nullptr, const LowRank *lr = nullptr)
{
    if (!p) return fail("null plan");
    // scalar products of the result asked for by pl_plan_arm_post_dots (one field): the request is consumed HERE, before any early
    // return -- a call that fails must not leave the plan armed for a later one (whose `pre` buffers may be gone by then)
    PostDots dots_now;
    const bool want_dots = p->dots_armed;
    const int dots_nf = p->dots_nf;
    p->dots_armed = false;
    if (want_dots) dots_now = p->dots;
    if (want_dots && dots_nf != 1) return fail("pl_cg_fwd_tt: armed scalar products need one field");
    if (!alm_in || !alm_out || !n_inv) return fail("pl_cg_fwd_tt: null alm / n_inv pointer");
    // pinv_md: the templates are exactly (monopole, dipole) and evaluated from the ring geometry (k_tproj_md_*): nmodes = 4, no matrices
    if (pinv_md && (nmodes != 4 || !scratch)) return fail("pl_cg_fwd_tt_md: nmodes = 4 and a scratch buffer are required");
    if (nmodes < 0 || nmodes > PL_TEMPLATE_MAX_MODES || (nmodes > 0 && !pinv_md && (!pmat || !rmat || !scratch))) return fail("pl_cg_fwd_tt: bad template arguments");
    if ((alm_add == nullptr) != (fl_add == nullptr)) return fail("pl_cg_fwd_tt: alm_add and fl_add come together");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const DevPlan &P = p->P;
    if (grow(p, &p->phase, &p->phase_cap, pl_plan_phase_doubles(p, 0) * nb) || grow(p, &p->wmap, &p->wmap_cap, P.npix * nb)) return 1;
    // low-rank template update of the result (lr): its coefficient pass c = pm x reads the input only, so it runs beside the transforms
    // on a side stream of the plan (a parallel branch when the solve is replayed as a HIP graph) and leaves the critical path
    bool lr_forked = false;
    // every exit after the fork joins the side stream again (eager: nothing of this call is left running unordered with the caller's
    // stream; under graph capture: no unjoined branch, which would fail the capture with an unrelated message)
    struct JoinGuard {
        bool *forked; hipStream_t st; hipEvent_t ev;
        ~JoinGuard() { if (*forked) (void)hipStreamWaitEvent(st, ev, 0); }
    } join_guard{&lr_forked, st, p->fs.join[FftStreams::kN - 1]};
    // (coarse grids: the pass rides in the prologue launch of the synthesis instead -- no fork, the replayed solve stays one chain of kernels)
    int lr_nt = 0, lr_np = 0;
    tproj_coeffs_shape_host(2 * P.nalm, &lr_nt, &lr_np);
    const bool lr_in_prologue = lr && lr->nmodes > 0 && lr_nt == 256 && lr_prologue_enabled();
    if (lr && lr->nmodes > 0 && !lr_in_prologue) {
        hipStream_t side = p->fs.ok ? p->fs.s[FftStreams::kN - 1] : nullptr;
        if (side && hipEventRecord(p->fs.fork, st) == hipSuccess && hipStreamWaitEvent(side, p->fs.fork, 0) == hipSuccess) {
            launch_template_project(2 * P.nalm, lr->nmodes, const_cast<double *>(alm_in), nullptr, lr->pm, lr->rm, lr->scratch, side, nb, alm_out, 1);
            lr_forked = hipEventRecord(p->fs.join[FftStreams::kN - 1], side) == hipSuccess;
            if (!lr_forked) return fail("pl_cg_fwd_tt_lr: event record on the side stream failed");
        } else {
            launch_template_project(2 * P.nalm, lr->nmodes, const_cast<double *>(alm_in), nullptr, lr->pm, lr->rm, lr->scratch, st, nb, alm_out, 1);
        }
        HIPCHK(hipGetLastError());
    }
    if (lr_in_prologue) {
        bool done = false;
        if (legendre_synth_impl(p, 0, alm_in, fl_in, p->phase, stream, false, nb, lr->nmodes, lr->pm, lr->scratch, &done)) return 1;
        if (!done) {  // (cannot happen while the two shape rules agree; kept correct anyway)
            launch_template_project(2 * P.nalm, lr->nmodes, const_cast<double *>(alm_in), nullptr, lr->pm, lr->rm, lr->scratch, st, nb, alm_out, 1);
            HIPCHK(hipGetLastError());
        }
    } else if (legendre_synth_impl(p, 0, alm_in, fl_in, p->phase, stream, false, nb)) return 1;
    // the weighting always rides in the synthesis-side FFT kernels; the projection too where every ring runs in the generic kernel
    const bool fused = nmodes == 0 || (!pinv_md && fft_all_generic(P, p->F) && nmodes <= kFuseModes);
    NinvProj W;
    W.n_inv = n_inv;
    if (fused && nmodes > 0) {
        if (grow(p, &p->tparts, &p->tparts_cap, (int64_t)nmodes * P.npairs * nb)) return 1;
        W.nmodes = nmodes; W.nparts = P.npairs; W.parts = p->tparts; W.pm = pmat; W.rm = rmat;
    }
    // no projection in pixel space and every ring in the generic kernel (the coarse levels of the chains): the ring transforms both ways
    // and the weighting in one launch, no map written (PLSHTS_CG_ROUNDTRIP=0: the two launches)
    const bool roundtrip = nmodes == 0 && cg_roundtrip_enabled() && fft_all_generic(P, p->F);
    if (roundtrip) {
        ProfScope ps(p, PK_FFT_SYNTH, st);
        HIPCHK(launch_ring_roundtrip(P, p->F, mlim_of(p, 0), nb, p->phase, n_inv, st));
    } else {
        ProfScope ps(p, PK_FFT_SYNTH, st);
        HIPCHK(launch_phase2map(P, p->F, p->fs, mlim_of(p, 0), nb, p->phase, p->wmap, st, &W));
    }
    if (!fused) {  // the map arrives weighted: coefficients and projection only
        if (pinv_md) launch_template_project_md(P, nb, p->wmap, n_inv, 1, pinv_md, scratch, st);
        else launch_template_project(P.npix, nmodes, p->wmap, nullptr, pmat, rmat, scratch, st, nb);
        HIPCHK(hipGetLastError());
    }
    if (!roundtrip) {
        ProfScope ps(p, PK_FFT_ANAL, st);
        HIPCHK(launch_map2phase(P, p->F, p->fs, mlim_of(p, 0), nb, p->wmap, p->phase, st, (fused && nmodes > 0) ? &W : nullptr));
    }
    const int RG = rings_per_group(0, P);
    const int ngroups = (P.npairs + RG - 1) / RG;
    if (grow(p, &p->partial, &p->partial_cap, (int64_t)ngroups * P.nent0 * 4 * nb)) return 1;
    const bool lr_on = lr && lr->nmodes > 0;
    const PostDots *dots = want_dots ? &dots_now : nullptr;
    if (lr_on && lr_forked) {  // the coefficients are needed from here on
        lr_forked = false;     // (joined here: the guard has nothing left to do)
        HIPCHK(hipStreamWaitEvent(st, p->fs.join[FftStreams::kN - 1], 0));
    }
    if (lr_on) {  // alm_out -= rm^t c inside the post-processing of the analysis (k_post0), every batch entry from its own coefficients
        int nparts = 0, pstride = 0;
        tproj_parts_layout(2 * P.nalm, &nparts, &pstride);
        ProfScope ps(p, PK_LEG_ANAL0, st);
        launch_anal0(P, p->phase, p->partial, fl_out, alm_out, st, alm_add, fl_add, nb, lr->nmodes, nparts, pstride, lr->rm, lr->scratch, dots,
                     tproj_parts_bstride());
    } else {
        ProfScope ps(p, PK_LEG_ANAL0, st);
        launch_anal0(P, p->phase, p->partial, fl_out, alm_out, st, alm_add, fl_add, nb, 0, 0, 0, nullptr, nullptr, dots);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// pl_cg_fwd_tt_b with plain N^-1 weighting, followed by the low-rank update alm_out -= hrm^t (hpm alm_in) (the template projection in
// harmonic space, pl_lowrank_update_b) whose coefficient pass overlaps the transforms.  Same results as the two separate calls.
int pl_cg_fwd_tt_lr_b(pl_plan *p, int nb, const double *alm_in, const double *fl_in, const double *n_inv, int nmodes, const double *hpm, const double *hrm,
                      double *scratch, const double *alm_add, const double *fl_add, double *alm_out, const double *fl_out, void *stream)
{
    PL_NB_CHECK("pl_cg_fwd_tt_lr_b");
    if (nmodes < 1 || nmodes > PL_TEMPLATE_MAX_MODES || !hpm || !hrm || !scratch) return fail("pl_cg_fwd_tt_lr_b: bad low-rank arguments");
    if (alm_out == alm_in) return fail("pl_cg_fwd_tt_lr_b: alm_out must not alias alm_in");
    LowRank lr;
    lr.nmodes = nmodes; lr.pm = hpm; lr.rm = hrm; lr.scratch = scratch;
    return cg_fwd_tt_impl(p, nb, alm_in, fl_in, n_inv, 0, nullptr, nullptr, nullptr, alm_add, fl_add, alm_out, fl_out, stream, nullptr, &lr);
}

int pl_cg_fwd_tt(pl_plan *p, const double *alm_in, const double *fl_in, const double *n_inv, int nmodes, const double *pmat, const double *rmat,
                 double *scratch, const double *alm_add, const double *fl_add, double *alm_out, const double *fl_out, void *stream)
{
    return cg_fwd_tt_impl(p, 1, alm_in, fl_in, n_inv, nmodes, pmat, rmat, scratch, alm_add, fl_add, alm_out, fl_out, stream);
}

int pl_plan_fft_all_generic(const pl_plan *p) { return p ? (fft_all_generic(p->P, p->F) ? 1 : 0) : 0; }

int64_t pl_template_md_scratch_doubles(const pl_plan *p, int nb) { return p ? (int64_t)nb * 4 * (p->P.npairs + 1) : 0; }

int pl_template_project_md_b(pl_plan *p, int nb, double *tmap, const double *n_inv, const double *pinv_dev, double *scratch, void *stream)
{
    if (!p || !tmap || !n_inv || !pinv_dev || !scratch) return fail("pl_template_project_md_b: null pointer");
    PL_NB_CHECK("pl_template_project_md_b");
    launch_template_project_md(p->P, nb, tmap, n_inv, 0, pinv_dev, scratch, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_cg_fwd_tt_md_b(pl_plan *p, int nb, const double *alm_in, const double *fl_in, const double *n_inv, const double *pinv_dev, double *scratch,
                      const double *alm_add, const double *fl_add, double *alm_out, const double *fl_out, void *stream)
{
    PL_NB_CHECK("pl_cg_fwd_tt_md_b");
    if (!pinv_dev) return fail("pl_cg_fwd_tt_md_b: null pinv");
    return cg_fwd_tt_impl(p, nb, alm_in, fl_in, n_inv, 4, nullptr, nullptr, scratch, alm_add, fl_add, alm_out, fl_out, stream, pinv_dev);
}

int pl_cg_fwd_tt_b(pl_plan *p, int nb, const double *alm_in, const double *fl_in, const double *n_inv, int nmodes, const double *pmat, const double *rmat,
                   double *scratch, const double *alm_add, const double *fl_add, double *alm_out, const double *fl_out, void *stream)
{
    PL_NB_CHECK("pl_cg_fwd_tt_b");
    return cg_fwd_tt_impl(p, nb, alm_in, fl_in, n_inv, nmodes, pmat, rmat, scratch, alm_add, fl_add, alm_out, fl_out, stream);
}

// The polarization CG operator (fwd_op.calc, plancklens/qcinv/opfilt_pp.py:69-78, apply_alm :190-205 with the single-map apply_map
// :207-215): (E, B)_out = fl_out * Y2^t [n_inv * Y2 (fl_in * (E, B)_in)] + (fl_add_e E_add, fl_add_b B_add).  E and B are separate arrays.
// The weighting rides in the synthesis-side ring-FFT launches (every kernel class), the add terms in k_posts.
static int cg_fwd_pp_impl(pl_plan *p, int nb, const double *elm_in, const double *blm_in, const double *fl_in, const double *n_inv, const double *elm_add,
                          const double *blm_add, const double *fl_add_e, const double *fl_add_b, double *elm_out, double *blm_out, const double *fl_out,
                          void *stream, const double *n_qu = nullptr, const double *n_uu = nullptr)
{
    // n_qu / n_uu given: n_inv is the QQ map of a (QQ, QU, UU) noise model -- the weighting is then one pass of k_map_qu_weight between
    // the two ring-FFT stages instead of riding in the synthesis-side kernels
    if (!p) return fail("null plan");
    // armed scalar products (pl_plan_arm_post_dots, two fields): consumed before any early return, as in cg_fwd_tt_impl
    PostDots dots_now;
    const bool want_dots = p->dots_armed;
    const int dots_nf = p->dots_nf;
    p->dots_armed = false;
    if (want_dots) dots_now = p->dots;
    if (want_dots && dots_nf != 2) return fail("pl_cg_fwd_pp: armed scalar products need two fields");
    if ((n_qu == nullptr) != (n_uu == nullptr)) return fail("pl_cg_fwd_pp: n_qu and n_uu come together");
    if (!elm_in || !blm_in || !elm_out || !blm_out || !n_inv) return fail("pl_cg_fwd_pp: null alm / n_inv pointer");
    if ((elm_add == nullptr) != (blm_add == nullptr) || (elm_add && (!fl_add_e || !fl_add_b)))
        return fail("pl_cg_fwd_pp: elm_add, blm_add, fl_add_e and fl_add_b come together");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const DevPlan &P = p->P;
    const int spin = 2;
    if (ensure_spin(p, spin)) return 1;
    if (grow(p, &p->phase, &p->phase_cap, pl_plan_phase_doubles(p, spin) * nb) || grow(p, &p->wmap, &p->wmap_cap, 2 * P.npix * nb) ||
        grow(p, &p->prep, &p->prep_cap, p->nent[spin] * 4 * nb))
        return 1;
    launch_preps_gc(P, p->S[spin], spin, elm_in, blm_in, fl_in, p->prep, st, nb);
    // Block vectors on the grids where the kernel is bound by FMA issue: the entries go through the synthesis two at a time on one
    // recursion (k_leg_synths<R, false, 2>: 20 instead of 24 FMAs per step and pair, bit-identical maps); odd block sizes go unpaired.
    // On the coarse grids (launch latency, not FMA issue) every entry is a workgroup row of the ordinary kernel.
    static const int pair_min_nside = dbg_env_int("PLSHTS_CG_PAIR_NSIDE", 1024);
    if (nb >= 2 && (nb & 1) == 0 && P.nside >= pair_min_nside) {  // (odd block sizes take the unpaired route)
        ProfScope ps(p, PK_LEG_SYNTHS_BATCH2, st);
        launch_synths_batch2(P, p->S[spin], spin, p->prep, p->prep + p->nent[spin] * 4, p->phase, st, nb / 2);
    } else {
        ProfScope ps(p, PK_LEG_SYNTHS, st);
        launch_synths(P, p->S[spin], spin, p->prep, p->phase, st, false, nb);
    }
    HIPCHK(hipGetLastError());
    NinvProj W;
    W.n_inv = n_qu ? nullptr : n_inv;
    const bool roundtrip = !n_qu && cg_roundtrip_enabled() && fft_all_generic(P, p->F);  // as in cg_fwd_tt_impl: Q and U weighted alike
    if (roundtrip) {
        ProfScope ps(p, PK_FFT_SYNTH, st);
        HIPCHK(launch_ring_roundtrip(P, p->F, mlim_of(p, spin), 2 * nb, p->phase, n_inv, st));
    } else {
        ProfScope ps(p, PK_FFT_SYNTH, st);
        HIPCHK(launch_phase2map(P, p->F, p->fs, mlim_of(p, spin), 2 * nb, p->phase, p->wmap, st, &W));
    }
    if (n_qu) {
        launch_map_qu_weight(P.npix, p->wmap, p->wmap + P.npix, n_inv, n_qu, n_uu, st, nb, 2 * P.npix);
        HIPCHK(hipGetLastError());
    }
    if (!roundtrip) {
        ProfScope ps(p, PK_FFT_ANAL, st);
        HIPCHK(launch_map2phase(P, p->F, p->fs, mlim_of(p, spin), 2 * nb, p->wmap, p->phase, st));
    }
    const int RG = rings_per_group(spin, P);
    const int ngroups = (P.npairs + RG - 1) / RG;
    if (grow(p, &p->partial, &p->partial_cap, (int64_t)ngroups * p->nent[spin] * 4 * nb)) return 1;
    {
        ProfScope ps(p, PK_LEG_ANALS, st);
        launch_anals_gc(P, p->S[spin], spin, p->nent[spin], p->phase, p->partial, fl_out, elm_out, blm_out, st, elm_add, blm_add, fl_add_e,
                        fl_add_b, nb, want_dots ? &dots_now : nullptr);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_cg_fwd_pp(pl_plan *p, const double *elm_in, const double *blm_in, const double *fl_in, const double *n_inv, const double *elm_add,
                 const double *blm_add, const double *fl_add_e, const double *fl_add_b, double *elm_out, double *blm_out, const double *fl_out,
                 void *stream)
{
    return cg_fwd_pp_impl(p, 1, elm_in, blm_in, fl_in, n_inv, elm_add, blm_add, fl_add_e, fl_add_b, elm_out, blm_out, fl_out, stream);
}

int pl_cg_fwd_pp_qu_b(pl_plan *p, int nb, const double *elm_in, const double *blm_in, const double *fl_in, const double *n_qq, const double *n_qu,
                      const double *n_uu, const double *elm_add, const double *blm_add, const double *fl_add_e, const double *fl_add_b, double *elm_out,
                      double *blm_out, const double *fl_out, void *stream)
{
    PL_NB_CHECK("pl_cg_fwd_pp_qu_b");
    if (!n_qu || !n_uu) return fail("pl_cg_fwd_pp_qu_b: null noise map");
    return cg_fwd_pp_impl(p, nb, elm_in, blm_in, fl_in, n_qq, elm_add, blm_add, fl_add_e, fl_add_b, elm_out, blm_out, fl_out, stream, n_qu, n_uu);
}

int pl_cg_fwd_pp_b(pl_plan *p, int nb, const double *elm_in, const double *blm_in, const double *fl_in, const double *n_inv, const double *elm_add,
                   const double *blm_add, const double *fl_add_e, const double *fl_add_b, double *elm_out, double *blm_out, const double *fl_out,
                   void *stream)
{
    PL_NB_CHECK("pl_cg_fwd_pp_b");
    return cg_fwd_pp_impl(p, nb, elm_in, blm_in, fl_in, n_inv, elm_add, blm_add, fl_add_e, fl_add_b, elm_out, blm_out, fl_out, stream);
}

int pl_alm_splice(int lmax_lo, const double *alm_lo, int lmax_hi, const double *alm_hi, int lsplit, double *out, void *stream)
{
    if (lsplit > lmax_lo || lsplit > lmax_hi || lmax_hi < 0) return fail("pl_alm_splice: lsplit exceeds a band-limit");
    launch_alm_splice(lmax_lo, alm_lo, lmax_hi, alm_hi, lsplit, out, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_alm_splice_fl(int lmax_lo, const double *alm_lo, int lmax_hi, const double *alm_hi, const double *fl_hi, int lsplit, double *out, void *stream)
{
    if (lsplit > lmax_lo || lsplit > lmax_hi || lmax_hi < 0 || !fl_hi) return fail("pl_alm_splice_fl: lsplit exceeds a band-limit, or null filter");
    launch_alm_splice(lmax_lo, alm_lo, lmax_hi, alm_hi, lsplit, out, static_cast<hipStream_t>(stream), fl_hi);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_almxfl_add(int lmax, const double *a, const double *b, const double *fl, int nfl, double *out, void *stream)
{
    if (lmax < 0 || !a || !b || !fl || !out) return fail("pl_almxfl_add: bad arguments");
    launch_almxfl_add(lmax, a, b, fl, nfl, out, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_alm_lincomb(int lmax, int nout, const int *nterm, const double *const *alm, const double *const *fl, double *const *out, void *stream)
{
    if (lmax < 0 || nout < 1 || nout > 2 || !nterm || !alm || !fl || !out) return fail("pl_alm_lincomb: bad arguments");
    for (int k = 0; k < nout; ++k) {
        if (nterm[k] < 1 || nterm[k] > 2 || !out[k]) return fail("pl_alm_lincomb: one or two terms per output");
        for (int t = 0; t < nterm[k]; ++t) {
            if (!alm[2 * k + t] || !fl[2 * k + t]) return fail("pl_alm_lincomb: null term");
            if (alm[2 * k + t] == out[k] && t > 0) return fail("pl_alm_lincomb: an output may alias its first term only");
        }
    }
    launch_alm_lincomb(lmax, nout, nterm, alm, fl, out, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_gemv(int nrows, int ncols, int64_t lda, const double *A, const double *x, double *y, void *stream)
{
    if (nrows < 0 || ncols < 0 || lda < ncols || !A || !x || !y) return fail("pl_gemv: bad arguments");
    if (nrows == 0) return 0;
    launch_gemv(nrows, ncols, lda, A, x, y, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// pre_op_split with a dense low block and a diagonal high part as one launch (k_gemv_split), nf = 1 (temperature) or 2 (E, B) fields:
// alm_out[f] (lmax_hi) = [rows of field f of A ([alm_hi[0] | alm_hi[1]] truncated to lmax_lo) | fl_hi[f] alm_hi[f] above lmax_lo].  A: the
// (nf 2 nalm_lo)^2 flat matrix of pre_op_dense (row stride lda, 16-byte aligned); map_dev: nalm_lo int32 indices of the lmax_lo entries in the
// lmax_hi layout.  No alm_out may alias an alm_hi.
int pl_gemv_split(int nf, int lmax_lo, int lmax_hi, int64_t lda, const double *A, const double *const *alm_hi, const int *map_dev, const double *const *fl_hi,
                  double *const *alm_out, void *stream)
{
    if (nf < 1 || nf > 2 || lmax_lo < 0 || lmax_hi <= lmax_lo || !A || !alm_hi || !map_dev || !fl_hi || !alm_out) return fail("pl_gemv_split: bad arguments");
    for (int f = 0; f < nf; ++f) {
        if (!alm_hi[f] || !fl_hi[f] || !alm_out[f]) return fail("pl_gemv_split: null field pointer");
        for (int g = 0; g < nf; ++g) if (alm_out[f] == alm_hi[g]) return fail("pl_gemv_split: output aliases an input");
    }
    const int nrows = nf * (lmax_lo + 1) * (lmax_lo + 2);
    if (lda < nrows || (lda & 1) || (reinterpret_cast<uintptr_t>(A) & 15)) return fail("pl_gemv_split: the matrix must be 16-byte aligned with an even row stride");
    launch_gemv_split(nf, lda, A, alm_hi, map_dev, lmax_lo, lmax_hi, fl_hi, alm_out, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_gemv_split_dot_count(int nf, int lmax_lo, int lmax_hi) { return gemv_split_dot_count(nf, lmax_lo, lmax_hi); }
int pl_alm_splice_dot_count(int lmax_hi) { return alm_splice_dot_count(lmax_hi); }

// pl_gemv_split that also leaves pl_gemv_split_dot_count(nf, lmax_lo, lmax_hi) partial sums of sum_f <alm_out[f], q[f]> in pre_dev
int pl_gemv_split_dot(int nf, int lmax_lo, int lmax_hi, int64_t lda, const double *A, const double *const *alm_hi, const int *map_dev,
                      const double *const *fl_hi, double *const *alm_out, const double *const *q, int lmin, double *pre_dev, void *stream)
{
    if (nf < 1 || nf > 2 || lmax_lo < 0 || lmax_hi <= lmax_lo || !A || !alm_hi || !map_dev || !fl_hi || !alm_out || !q || !pre_dev)
        return fail("pl_gemv_split_dot: bad arguments");
    for (int f = 0; f < nf; ++f) {
        if (!alm_hi[f] || !fl_hi[f] || !alm_out[f] || !q[f]) return fail("pl_gemv_split_dot: null field pointer");
        for (int g = 0; g < nf; ++g) if (alm_out[f] == alm_hi[g]) return fail("pl_gemv_split_dot: output aliases an input");
    }
    const int nrows = nf * (lmax_lo + 1) * (lmax_lo + 2);
    if (lda < nrows || (lda & 1) || (reinterpret_cast<uintptr_t>(A) & 15)) return fail("pl_gemv_split_dot: the matrix must be 16-byte aligned with an even row stride");
    launch_gemv_split(nf, lda, A, alm_hi, map_dev, lmax_lo, lmax_hi, fl_hi, alm_out, static_cast<hipStream_t>(stream), q, lmin < 0 ? 0 : lmin, pre_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

// pl_alm_splice_b (fl_hi may be NULL) that also leaves, per batch entry, pl_alm_splice_dot_count(lmax_hi) partial sums of <out, q> in pre_dev
int pl_alm_splice_dot_b(int lmax_lo, int nb, const double *alm_lo, int lmax_hi, const double *alm_hi, const double *fl_hi, int lsplit, double *out,
                        const double *q, int lmin, double *pre_dev, void *stream)
{
    PL_NB_CHECK("pl_alm_splice_dot_b");
    if (lsplit > lmax_lo || lsplit > lmax_hi || lmax_hi < 0) return fail("pl_alm_splice_dot_b: lsplit exceeds a band-limit");
    if (!q || !pre_dev) return fail("pl_alm_splice_dot_b: null q / pre_dev");
    launch_alm_splice(lmax_lo, alm_lo, lmax_hi, alm_hi, lsplit, out, static_cast<hipStream_t>(stream), fl_hi, nb, q, lmin < 0 ? 0 : lmin, pre_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_copy_slim(const double *src_dev, double *dst, int64_t ndoubles, int nblocks, void *stream)
{
    if (!src_dev || !dst || ndoubles < 0) return fail("pl_copy_slim: bad arguments");
    if (((reinterpret_cast<uintptr_t>(src_dev) | reinterpret_cast<uintptr_t>(dst)) & 15) != 0) return fail("pl_copy_slim: pointers must be 16-byte aligned");
    if (ndoubles == 0) return 0;
    launch_copy_slim(src_dev, dst, ndoubles, nblocks, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// The table of pl_map2alm_ind: n <= 8 device addresses (given by value: `addrs` is a host array read before the call returns) stored to table_dev by a
// one-workgroup kernel on `stream` -- ordered after whatever still reads the old entries, before whatever reads the new ones.
int pl_store_addresses(int n, const unsigned long long *addrs, unsigned long long *table_dev, void *stream)
{
    if (n < 1 || n > 8 || !addrs || !table_dev) return fail("pl_store_addresses: 1 <= n <= 8 addresses, non-null pointers");
    launch_store_addresses(n, addrs, table_dev, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_map_mul(int64_t n, const double *a, const double *b, double *out, void *stream)
{
    launch_map_mul(n, a, b, out, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_map_qu_weight(int64_t n, double *qmap, double *umap, const double *nqq, const double *nqu, const double *nuu, void *stream)
{
    if (n < 0 || !qmap || !umap || !nqq || !nqu || !nuu) return fail("pl_map_qu_weight: bad arguments");
    if (n == 0) return 0;
    launch_map_qu_weight(n, qmap, umap, nqq, nqu, nuu, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_map_cmul(int64_t n, const double *ar, const double *ai, double s1, const double *br, const double *bi, double s2, double sign,
                double *outr, double *outi, int accumulate, void *stream)
{
    launch_map_cmul(n, ar, ai, s1, br, bi, s2, sign, outr, outi, accumulate, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_qe_lens_product(int64_t n, const double *tmap, const double *gt, const double *ct, const double *rep, const double *imp,
                       const double *g3, const double *c3, const double *g1, const double *c1, double *out_re, double *out_im, void *stream)
{
    if (!out_re || !out_im) return fail("null output");
    if (!tmap && !rep) return fail("neither the temperature nor the polarization part given");
    if ((tmap && (!gt || !ct)) || (rep && (!imp || !g3 || !c3 || !g1 || !c1))) return fail("incomplete set of leg maps");
    launch_qe_lens_product(n, tmap, gt, ct, rep, imp, g3, c3, g1, c1, out_re, out_im, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_map_add_normal(int64_t n, const double *map_in, double *map_out, double sigma, uint64_t key, void *stream)
{
    if (n < 0 || !map_out) return fail("pl_map_add_normal: bad arguments");
    if (((reinterpret_cast<uintptr_t>(map_in) | reinterpret_cast<uintptr_t>(map_out)) & 15) != 0) return fail("pl_map_add_normal: pointers must be 16-byte aligned");
    if (n == 0) return 0;
    launch_map_add_normal(n, map_in, map_out, sigma, key, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int pl_alm_unit_phases(int lmax, double *alm_out, uint64_t key, void *stream)
{
    if (lmax < 0 || !alm_out) return fail("pl_alm_unit_phases: bad arguments");
    launch_alm_unit_phases(lmax, alm_out, key, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

double pl_fma64_rate_tflops(int mode, int iters, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *out = nullptr;
    if (mode < 0 || mode > 2) return -1.0;
    if (hipMalloc(reinterpret_cast<void **>(&out), 8) != hipSuccess) return -1.0;
    const int nblk = 256 * 8;
    hipEvent_t e0, e1;
    bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    launch_fma_peak(mode, iters / 8 + 1, out, nblk, st);
    ok = ok && hipEventRecord(e0, st) == hipSuccess;
    launch_fma_peak(mode, iters, out, nblk, st);
    ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
    float ms = 0.f;
    ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(out);
    if (!ok || ms <= 0.f) return -1.0;
    const double per_wave = 16.0 * 128.0;  // per trip and wave: 16 v_fma_f64 of 128 flop each
    const double flops = per_wave * (double)iters * 4.0 * nblk;
    return flops / (ms * 1e-3) / 1e12;
}

double pl_fma64_peak_tflops(int iters, void *stream) { return pl_fma64_rate_tflops(1, iters, stream); }

}  // extern "C"
