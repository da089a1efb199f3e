// Legendre stage of the HEALPix SHTs for gfx950 (MI355X): alm <-> per-ring Fourier coefficients F_m(ring).
//
// Mapping (DESIGN.md "Legendre kernels"): one wavefront = one azimuthal order m; its 64 lanes own
// R ring pairs each (north ring + mirror south ring share every recursion value by parity).  For fixed m
// the recursion coefficients and the (pre-transformed) a_lm depend on l only, so they are wave-uniform:
// they are fetched with scalar loads and the vector ALU stream is pure v_fma_f64 with one scalar operand.
// A 256-thread workgroup takes 4 consecutive m for the same 64 R ring pairs and transposes its results
// through LDS so that the ring-major phase array [pair][m] is written / read in 128-byte (spin 0) or
// 256-byte (spin s) contiguous pieces.
//
// Bound: FP64 FMA issue; no MFMA (a recurrence, not a contraction).
//
// Batches (several right-hand sides of the conjugate-gradient filter through one launch, api.hip pl_cg_fwd_*_b): the Legendre
// kernels take the batch index from blockIdx.y, the prep / post kernels from blockIdx.z.  Entry b reads / writes the b-th
// consecutive alm, prep and partial-sum array; in the phase array [ring pair][component][m][4] the components of entry b follow
// those of entry b - 1 (component count = gridDim.y x components of one transform), which is the layout the ring-FFT kernels
// take for any number of components.  With a batch of one every address is what it was.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "device_plan.h"
#include "plshts_internal.h"
#include "legendre_math.h"
#include "tproj_device.h"


namespace plshts {

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ bool wave_all(bool p) { return __all(p) != 0; }
__device__ __forceinline__ bool wave_any(bool p) { return __any(p) != 0; }

// Wave-uniform coefficient loads.  The l loops keep two coefficient sets in flight (issued one set ahead of their use,
// fenced with sched_barrier + an explicit lgkmcnt(0) wait).  An asm-pinned s_load variant of the same idea was removed:
// the register allocator does not know that an asm output is written asynchronously and may copy it before the wait
// (NaNs in the analysis kernels under SGPR pressure).
typedef double d2v_t __attribute__((ext_vector_type(2)));
typedef double d4v_t __attribute__((ext_vector_type(4)));
typedef double d8v_t __attribute__((ext_vector_type(8)));
// Wave-uniform table loads through the constant address space: the tables are read-only for the lifetime of the
// kernel, and an address_space(4) load with a uniform address is always selected as s_load (a plain global load
// degrades to a per-lane VMEM load as soon as the kernel also stores to global memory: 64 VGPRs of coefficients).
template <class T>
__device__ __forceinline__ T ldc(const void *p)
{
    typedef const T __attribute__((address_space(4))) *cptr_t;
    return *(cptr_t)(unsigned long long)p;
}
__device__ __forceinline__ d8v_t ld8(const double2 *p) { return ldc<d8v_t>(p); }

// L2 prefetch of the wave-uniform coefficient streams.  The scalar loads of the recursion loops are waited for with
// lgkmcnt(0) one compute section (~50 FMA) after their issue; that hides an L2 hit but not an L2 miss (the tables do not
// fit the 4 MB L2 of an XCD: every line is fetched from Infinity Cache / HBM once per XCD).  A vector load of one dword
// per 64 B line, one line per lane, issued ~64 l ahead pulls the lines into L2.  Its result is only consumed (summed
// into a dummy) when the next window is issued, 32 l later, so the vmcnt wait in front of that use never stalls.
struct StreamPrefetch {
    float pending = 0.f, acc = 0.f;
    int next = 0;  // next stream position at which to issue
    static constexpr int kSpan = 32, kAhead = 64;
    // streams a (SA bytes per step) and b (SB bytes per step) of n steps each; SPAN steps starting at lwin
    template <int SA, int SB, int SPAN>
    __device__ __forceinline__ void issue(const void *pa, const void *pb, int lwin, int n, int lane)
    {
        constexpr int LA = SPAN * SA / 64, LB = SPAN * SB / 64;  // lines per window
        static_assert(LA + LB <= 64 && (64 % SA == 0) && (64 % SB == 0), "window does not fit one wave");
        const bool isb = lane >= LA;
        const int li = lwin + (isb ? min(lane - LA, LB - 1) * (64 / SB) : lane * (64 / SA));
        const char *p = isb ? static_cast<const char *>(pb) + (int64_t)min(li, n - 1) * SB
                            : static_cast<const char *>(pa) + (int64_t)min(li, n - 1) * SA;
        acc += pending;
        pending = *reinterpret_cast<const volatile float *>(p);
    }
    template <int SA, int SB>
    __device__ __forceinline__ void start(const void *pa, const void *pb, int i, int n, int lane)
    {
        next = i;
        if (i < n) issue<SA, SB, kAhead>(pa, pb, i, n, lane);
    }
    // call once per loop trip with the current position i
    template <int SA, int SB>
    __device__ __forceinline__ void step(const void *pa, const void *pb, int i, int n, int lane)
    {
        if (i >= next) {
            if (next + kAhead < n) issue<SA, SB, kSpan>(pa, pb, next + kAhead, n, lane);
            next += kSpan;
        }
    }
    __device__ __forceinline__ void drain()
    {
        acc += pending;
        asm volatile("" : : "v"(acc));  // keeps the loads alive; nothing is stored
    }
};

// -----------------------------------------------------------------------------------------------------
// alm -> recursion-basis coefficients (fused hp.almxfl)
// -----------------------------------------------------------------------------------------------------
// spin 0: prep[e] = {c_re, c_im, d_re, d_im}
__device__ __forceinline__ void prep0_wg(const DevPlan &P, const double2 *__restrict__ alm_, const double *__restrict__ fl, double4 *__restrict__ prep_)
{
    const double2 *__restrict__ alm = alm_ + (int64_t)blockIdx.z * P.nalm;
    double4 *__restrict__ prep = prep_ + (int64_t)blockIdx.z * P.nent0;
    const int m = blockIdx.y;
    const int nil = (P.lmax - m) / 2 + 1;
    const int64_t base = P.off0[m];
    const int64_t abase = (int64_t)m * (2 * P.lmax + 1 - m) / 2;  // index of l = 0 of this m
    for (int il = blockIdx.x * blockDim.x + threadIdx.x; il < nil; il += gridDim.x * blockDim.x) {
        const int l = m + 2 * il;
        const int64_t e = base + il;
        double2 a0 = alm[abase + l];
        double2 a1 = make_double2(0., 0.), a2 = make_double2(0., 0.);
        if (l + 1 <= P.lmax) a1 = alm[abase + l + 1];
        if (l + 2 <= P.lmax) a2 = alm[abase + l + 2];
        if (fl) {
            double f0 = fl[l], f1 = (l + 1 <= P.lmax) ? fl[l + 1] : 0., f2 = (l + 2 <= P.lmax) ? fl[l + 2] : 0.;
            a0.x *= f0; a0.y *= f0; a1.x *= f1; a1.y *= f1; a2.x *= f2; a2.y *= f2;
        }
        const double al = P.alpha0[e], e1 = P.eps0[2 * e], e2 = P.eps0[2 * e + 1];
        double4 o;
        o.x = al * (e1 * a0.x + e2 * a2.x);
        o.y = al * (e1 * a0.y + e2 * a2.y);
        o.z = al * a1.x;
        o.w = al * a1.y;
        prep[e] = o;
    }
}
__global__ void k_prep0(DevPlan P, const double2 *__restrict__ alm_, const double *__restrict__ fl, double4 *__restrict__ prep_)
{
    prep0_wg(P, alm_, fl, prep_);
}
// The same launch with the coefficient pass of a low-rank update on the SAME input riding along (the temperature CG operator with its template
// projection in harmonic space, pl_cg_fwd_tt_lr_b: c = pm alm_in reads the input only): workgroups y <= mmax are k_prep0's, the workgroups
// behind them are those of k_tproj_coeffs<256> / k_tproj_coeffs_b<256> on the 2 nalm doubles of the input (tproj_device.h: identical partial sums).
// On the coarse multigrid levels this replaces a forked side-stream launch: the replayed solve stays one chain of kernels on one hardware queue.
struct PrepLowRank { int nmodes, nparts, nb; const double *pm; double *parts; };
template <bool BATCH>
__global__ __launch_bounds__(256) void k_prep0_lr(DevPlan P, const double2 *__restrict__ alm_, const double *__restrict__ fl, double4 *__restrict__ prep_,
                                                  PrepLowRank L)
{
    if ((int)blockIdx.y <= P.mmax) { prep0_wg(P, alm_, fl, prep_); return; }
    const int part = ((int)blockIdx.y - (P.mmax + 1)) * (int)gridDim.x + (int)blockIdx.x;
    if (part >= L.nparts) return;
    double *t = const_cast<double *>(reinterpret_cast<const double *>(alm_));  // (read only: n_inv null)
    if constexpr (BATCH) {
        if ((int)blockIdx.z * kProjChunk >= L.nb) return;
        tproj_coeffs_b_wg<256>(2 * P.nalm, L.nmodes, L.nb, t, nullptr, L.pm, L.parts, part, L.nparts, blockIdx.z);
    } else {
        tproj_coeffs_wg<256>(2 * P.nalm, L.nmodes, t + (int64_t)blockIdx.z * 2 * P.nalm, nullptr, L.pm,
                             L.parts + (int64_t)blockIdx.z * (kProjMaxModes * kProjParts), part, L.nparts);
    }
}

// spin s: prep[e] = {An_re, An_im, Ap_re, Ap_im}
__global__ void k_preps(DevPlan P, DevSpinTab S, int spin, const double2 *__restrict__ almG_, const double2 *__restrict__ almC_,
                        const double *__restrict__ fl, double4 *__restrict__ prep_)
{
    const double2 *__restrict__ almG = almG_ + (int64_t)blockIdx.z * P.nalm;
    const double2 *__restrict__ almC = almC_ ? almC_ + (int64_t)blockIdx.z * P.nalm : nullptr;
    double4 *__restrict__ prep = prep_ + (int64_t)blockIdx.z * S.off[P.mmax + 1];
    const int m = blockIdx.y;
    const int l0 = m > spin ? m : spin;
    const int nl = P.lmax - l0 + 1;
    if (nl <= 0) return;
    const int64_t base = S.off[m];
    const int64_t abase = (int64_t)m * (2 * P.lmax + 1 - m) / 2;
    const double sg = (spin & 1) ? -1.0 : 1.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += gridDim.x * blockDim.x) {
        const int l = l0 + i;
        const int64_t e = base + i;
        const double2 g = almG[abase + l], c = almC ? almC[abase + l] : make_double2(0., 0.);
        double f = -0.5 * S.beta[e];
        if (fl) f *= fl[l];
        double4 o;
        // G + iC = (g.x - c.y) + i (g.y + c.x);  G - iC = (g.x + c.y) + i (g.y - c.x)
        o.x = f * sg * (g.x - c.y);
        o.y = f * sg * (g.y + c.x);
        o.z = f * (g.x + c.y);
        o.w = f * (g.y - c.x);
        prep[e] = o;
    }
}

// -----------------------------------------------------------------------------------------------------
// seed tables (device_plan.h DevSeedTab): phase A of a kernel family run once per plan
// -----------------------------------------------------------------------------------------------------
// One wave = (m, ring group of 64 R pairs) as in the Legendre kernels; the loop below is their phase A -- the same steps on the same
// coefficients in the same order, so the stored state is the state they would have computed -- with the family's activation threshold
// `thr` and check interval `gran` (8: the synthesis kernels' blocks; 16: the analysis kernels' tiles, two blocks each).
template <int R>
__global__ __launch_bounds__(256) void k_seed_gen0(DevPlan P, double thr, int gran, int *__restrict__ il_out, double2 *__restrict__ st_out,
                                                   int *__restrict__ sc_out)
{
    constexpr int RG = 64 * R;
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const int ngroups = (P.npairs + RG - 1) / RG;
    const int g = blockIdx.x % ngroups, m = 4 * (blockIdx.x / ngroups) + wave;
    if (m > P.mmax) return;
    Rec0 r[R];
    const double seed = P.seed0[m];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int ip = g * RG + k * 64 + lane;
        const int ipc = min(ip, P.npairs - 1);
        rec0_init(r[k], seed, m, P.cth[ipc], P.sth[ipc], ip < P.npairs && m <= P.mlim0[ipc]);
    }
    const int nil = (P.lmax - m) / 2 + 1;
    const double2 *__restrict__ ab = reinterpret_cast<const double2 *>(P.ab0) + P.off0[m];
    int il = 0;
    while (il + gran <= nil) {
        bool act = false;
#pragma unroll
        for (int k = 0; k < R; ++k) act = act || rec0_counts(r[k], thr);
        if (wave_any(act)) break;
        for (int b = 0; b < gran; b += 8) {
            for (int t = 0; t < 8; ++t) {
                const double2 c = ab[il + b + t];
#pragma unroll
                for (int k = 0; k < R; ++k) rec0_step_fast(r[k], c.x, c.y);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) rec0_renorm_up(r[k]);
        }
        il += gran;
    }
    if (lane == 0) il_out[m * ngroups + g] = il;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int64_t e = (int64_t)m * (ngroups * RG) + g * RG + k * 64 + lane;
        st_out[e] = make_double2(r[k].p0, r[k].p1);
        sc_out[e] = r[k].sc;
    }
}

template <int R>
__global__ __launch_bounds__(256) void k_seed_gens(DevPlan P, DevSpinTab S, int spin, double thr, int gran, int *__restrict__ il_out,
                                                   double4 *__restrict__ st_out, int2 *__restrict__ sc_out)
{
    constexpr int RG = 64 * R;
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const int ngroups = (P.npairs + RG - 1) / RG;
    const int g = blockIdx.x % ngroups, m = 4 * (blockIdx.x / ngroups) + wave;
    if (m > P.mmax) return;
    const int l0 = m > spin ? m : spin;
    RecS r[R];
    int i = 0;
    if (l0 <= P.lmax) {
        const double fn = S.seedfac_n[m], fp = S.seedfac_p[m];
        const int psin = S.psin[m], phalf = S.phalf[m], ucn = S.usecos_n[m], ucp = S.usecos_p[m];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int ip = g * RG + k * 64 + lane;
            const int ipc = min(ip, P.npairs - 1);
            recs_init(r[k], fn, fp, psin, phalf, ucn, ucp, P.cth[ipc], P.sth[ipc], P.chalf[ipc], P.shalf[ipc], ip < P.npairs && m <= S.mlim[ipc]);
        }
        const int nl = P.lmax - l0 + 1;
        const double2 *__restrict__ ab = reinterpret_cast<const double2 *>(S.ab) + S.off[m];
        while (i + gran <= nl) {
            bool act = false;
#pragma unroll
            for (int k = 0; k < R; ++k) act = act || recs_counts(r[k], thr);
            if (wave_any(act)) break;
            for (int b = 0; b < gran; b += 8) {
                for (int t = 0; t < 8; ++t) {
                    const double2 c = ab[i + b + t];
#pragma unroll
                    for (int k = 0; k < R; ++k) recs_step_fast(r[k], c.x, c.y);
                }
#pragma unroll
                for (int k = 0; k < R; ++k) recs_renorm_up(r[k]);
            }
            i += gran;
        }
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) { r[k].n0 = r[k].n1 = r[k].p0 = r[k].p1 = 0.0; r[k].x = 0.0; r[k].scn = r[k].scp = kNeverActive; }
    }
    if (lane == 0) il_out[m * ngroups + g] = i;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int64_t e = (int64_t)m * (ngroups * RG) + g * RG + k * 64 + lane;
        st_out[e] = make_double4(r[k].n0, r[k].n1, r[k].p0, r[k].p1);
        sc_out[e] = make_int2(r[k].scn, r[k].scp);
    }
}

// state of ring k of this lane from the table of its family (T.rg == 64 R checked by the caller)
__device__ __forceinline__ void seed_load0(const DevSeedTab &T, int m, int ip, double x, Rec0 &r)
{
    const int64_t e = (int64_t)m * T.npad + ip;
    const double2 v = reinterpret_cast<const double2 *>(T.st)[e];
    r.x2 = x * x; r.p0 = v.x; r.p1 = v.y; r.sc = T.sc[e];
}
__device__ __forceinline__ void seed_loads(const DevSeedTab &T, int m, int ip, double x, RecS &r)
{
    const int64_t e = (int64_t)m * T.npad + ip;
    const double4 v = reinterpret_cast<const double4 *>(T.st)[e];
    const int2 c = reinterpret_cast<const int2 *>(T.sc)[e];
    r.x = x; r.n0 = v.x; r.n1 = v.y; r.p0 = v.z; r.p1 = v.w; r.scn = c.x; r.scp = c.y;
}

// -----------------------------------------------------------------------------------------------------
// synthesis, spin 0
// -----------------------------------------------------------------------------------------------------
// IN2 (round 6): two inputs on one recursion -- batch entries (2 y, 2 y + 1) of the prep block share every recursion value (it depends on m and the
// ring only): 2 + 2 x 4 = 10 FMAs per two-l step and ring for two maps instead of 2 x (2 + 4) = 12.  The sums of each input are formed in the same
// order from the same values as by the single-input kernel: bit-identical phase values.  Phase components 2 y and 2 y + 1 of 2 gridDim.y.
template <int R, bool IN2 = false>
__global__ __launch_bounds__(256) void k_leg_synth0(DevPlan P, const double4 *__restrict__ prep_, double *__restrict__ phase)
{
    constexpr int NI = IN2 ? 2 : 1;
    const double4 *__restrict__ prep = prep_ + (int64_t)(NI * blockIdx.y) * P.nent0;
    constexpr int RG = 64 * R;
    __shared__ double tile[NI * RG * 16];  // [input][ring][m_local 4][4]
    const int wave = wave_id();
    const int lane = threadIdx.x & 63;
    const int ngroups = (P.npairs + RG - 1) / RG;
    // workgroups are dealt round-robin over the 8 XCDs: rotate the ring group with the m group so that every XCD
    // sees all latitudes (polar groups are light, equatorial ones heavy) -- otherwise the XCDs finish unevenly
    const int mg = P.mg0 + P.mgstride * (blockIdx.x / ngroups), g = (blockIdx.x % ngroups + mg) % ngroups;  // (m-group shard of the plan: device_plan.h)
    const int last = min(P.npairs - 1, g * RG + RG - 1);
    if (P.mlim0[last] < 4 * mg) return;  // every ring of the group is pruned for every m of the group
    const int m = 4 * mg + wave;

    double cr[NI][R], ci[NI][R], dr[NI][R], di[NI][R];
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int k = 0; k < R; ++k) cr[n][k] = ci[n][k] = dr[n][k] = di[n][k] = 0.0;

    if (m <= P.mmax) {
        Rec0 r[R];
        const double seed = P.seed0[m];
        const bool seeded = P.seed_syn0.rg == RG;  // the plan's table of states at the end of phase A (device_plan.h DevSeedTab)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int ip = g * RG + k * 64 + lane;
            const bool ok = ip < P.npairs && m <= P.mlim0[min(ip, P.npairs - 1)];
            const int ipc = min(ip, P.npairs - 1);
            if (seeded) seed_load0(P.seed_syn0, m, ip, P.cth[ipc], r[k]);
            else rec0_init(r[k], seed, m, P.cth[ipc], P.sth[ipc], ok);
        }
        const int nil = (P.lmax - m) / 2 + 1;
        const int64_t base = P.off0[m];
        const double2 *__restrict__ ab = reinterpret_cast<const double2 *>(P.ab0) + base;
        const double4 *__restrict__ cd = prep + base;
        const double4 *__restrict__ cd2 = cd + (IN2 ? P.nent0 : 0);
        const d2v_t *__restrict__ abv = reinterpret_cast<const d2v_t *>(ab);
        const d4v_t *__restrict__ cdv = reinterpret_cast<const d4v_t *>(cd);
        const d4v_t *__restrict__ cdv2 = reinterpret_cast<const d4v_t *>(cd2);
        int il = seeded ? __builtin_amdgcn_readfirstlane(P.seed_syn0.il[m * ngroups + g]) : 0;
        bool all_done = false;
        {
            bool live = false;
#pragma unroll
            for (int k = 0; k < R; ++k) live = live || (r[k].sc != kNeverActive);
            if (!wave_any(live)) il = nil;  // every ring of this wave is pruned for this m
        }
        StreamPrefetch pf, pf2;
        pf.start<32, 16>(cd, ab, il, nil, lane);
        if constexpr (IN2) pf2.start<32, 16>(cd2, ab, il, nil, lane);
        // phase A: no lane of the wave has reached the IEEE range yet -- recursion only; the rescale check is deferred
        // to the end of each block of 8 il (see rec0_renorm_up)
        while (il + 8 <= nil) {
            bool act = false;
#pragma unroll
            for (int k = 0; k < R; ++k) act = act || rec0_counts(r[k], kActSynth0);
            if (wave_any(act)) break;
            pf.step<32, 16>(cd, ab, il, nil, lane);
            if constexpr (IN2) pf2.step<32, 16>(cd2, ab, il, nil, lane);
            const d8v_t c0 = ld8(ab + il), c1 = ld8(ab + il + 4);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double cA = t < 4 ? c0[2 * t] : c1[2 * (t - 4)], cB = t < 4 ? c0[2 * t + 1] : c1[2 * (t - 4) + 1];
#pragma unroll
                for (int k = 0; k < R; ++k) rec0_step_fast(r[k], cA, cB);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) rec0_renorm_up(r[k]);
            il += 8;
        }
        // phase B: some lanes active, some still scaled -- fast steps, each ring's terms multiplied by its 0/1 mask,
        // masks and scales refreshed every 8 il
        while (il + 8 <= nil) {
            bool done = true;
#pragma unroll
            for (int k = 0; k < R; ++k) done = done && (r[k].sc == 0 || r[k].sc == kNeverActive);
            if (wave_all(done)) { all_done = true; break; }
            double mk[R];
#pragma unroll
            for (int k = 0; k < R; ++k) mk[k] = r[k].sc == 0 ? 1.0 : 0.0;
            pf.step<32, 16>(cd, ab, il, nil, lane);
            if constexpr (IN2) pf2.step<32, 16>(cd2, ab, il, nil, lane);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const d8v_t c = ld8(ab + il + 4 * h);
                const d8v_t q0 = ldc<d8v_t>(cdv + il + 4 * h), q1 = ldc<d8v_t>(cdv + il + 4 * h + 2);
                d8v_t p0 = q0, p1 = q1;
                if constexpr (IN2) { p0 = ldc<d8v_t>(cdv2 + il + 4 * h); p1 = ldc<d8v_t>(cdv2 + il + 4 * h + 2); }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int o = 4 * (t & 1);
                    const double qx = t < 2 ? q0[o] : q1[o], qy = t < 2 ? q0[o + 1] : q1[o + 1];
                    const double qz = t < 2 ? q0[o + 2] : q1[o + 2], qw = t < 2 ? q0[o + 3] : q1[o + 3];
                    const double px = t < 2 ? p0[o] : p1[o], py = t < 2 ? p0[o + 1] : p1[o + 1];
                    const double pz = t < 2 ? p0[o + 2] : p1[o + 2], pw = t < 2 ? p0[o + 3] : p1[o + 3];
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const double v = r[k].p1 * mk[k];
                        cr[0][k] = fma(v, qx, cr[0][k]); ci[0][k] = fma(v, qy, ci[0][k]);
                        dr[0][k] = fma(v, qz, dr[0][k]); di[0][k] = fma(v, qw, di[0][k]);
                        if constexpr (IN2) {
                            cr[1][k] = fma(v, px, cr[1][k]); ci[1][k] = fma(v, py, ci[1][k]);
                            dr[1][k] = fma(v, pz, dr[1][k]); di[1][k] = fma(v, pw, di[1][k]);
                        }
                        rec0_step_fast(r[k], c[2 * t], c[2 * t + 1]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < R; ++k) rec0_renorm_up(r[k]);
            il += 8;
        }
        // phase C: every live lane is in the IEEE range -- pure FMA stream, coefficient loads one trip ahead
        if (all_done) {
            auto one_step = [&](const d2v_t &c_ab, const d4v_t &c, const d4v_t &c2) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const double v = r[k].p1;
                    cr[0][k] = fma(v, c.x, cr[0][k]); ci[0][k] = fma(v, c.y, ci[0][k]);
                    dr[0][k] = fma(v, c.z, dr[0][k]); di[0][k] = fma(v, c.w, di[0][k]);
                    if constexpr (IN2) {
                        cr[1][k] = fma(v, c2.x, cr[1][k]); ci[1][k] = fma(v, c2.y, ci[1][k]);
                        dr[1][k] = fma(v, c2.z, dr[1][k]); di[1][k] = fma(v, c2.w, di[1][k]);
                    }
                    rec0_step_fast(r[k], c_ab.x, c_ab.y);
                }
            };
            // two coefficient sets of two steps each, loaded one set ahead (see k_leg_synths)
            d2v_t A0 = ldc<d2v_t>(abv + il), A1 = ldc<d2v_t>(abv + min(il + 1, nil - 1));
            d4v_t Ac0 = ldc<d4v_t>(cdv + il), Ac1 = ldc<d4v_t>(cdv + min(il + 1, nil - 1));
            d4v_t Ad0 = Ac0, Ad1 = Ac1;
            if constexpr (IN2) { Ad0 = ldc<d4v_t>(cdv2 + il); Ad1 = ldc<d4v_t>(cdv2 + min(il + 1, nil - 1)); }
            while (il + 3 < nil) {
                pf.step<32, 16>(cd, ab, il, nil, lane);
                if constexpr (IN2) pf2.step<32, 16>(cd2, ab, il, nil, lane);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                const d2v_t B0 = ldc<d2v_t>(abv + il + 2), B1 = ldc<d2v_t>(abv + il + 3);
                const d4v_t Bc0 = ldc<d4v_t>(cdv + il + 2), Bc1 = ldc<d4v_t>(cdv + il + 3);
                d4v_t Bd0 = Bc0, Bd1 = Bc1;
                if constexpr (IN2) { Bd0 = ldc<d4v_t>(cdv2 + il + 2); Bd1 = ldc<d4v_t>(cdv2 + il + 3); }
                __builtin_amdgcn_sched_barrier(0);
                one_step(A0, Ac0, Ad0); one_step(A1, Ac1, Ad1);
                __builtin_amdgcn_sched_barrier(0);
                const int ip = min(il + 4, nil - 2);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                A0 = ldc<d2v_t>(abv + ip); A1 = ldc<d2v_t>(abv + ip + 1); Ac0 = ldc<d4v_t>(cdv + ip); Ac1 = ldc<d4v_t>(cdv + ip + 1);
                if constexpr (IN2) { Ad0 = ldc<d4v_t>(cdv2 + ip); Ad1 = ldc<d4v_t>(cdv2 + ip + 1); }
                __builtin_amdgcn_sched_barrier(0);
                one_step(B0, Bc0, Bd0); one_step(B1, Bc1, Bd1);
                __builtin_amdgcn_sched_barrier(0);
                il += 4;
            }
            if (il + 1 < nil) { one_step(A0, Ac0, Ad0); one_step(A1, Ac1, Ad1); il += 2; }
        }
        // tail (at most 7 il, any mix of active and scaled lanes): one careful step at a time
        for (; il < nil; ++il) {
            const double2 c_ab = ab[il];
            const double4 c = cd[il];
            const double4 c2 = cd2[il];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double v = rec0_value(r[k]);
                cr[0][k] = fma(v, c.x, cr[0][k]); ci[0][k] = fma(v, c.y, ci[0][k]);
                dr[0][k] = fma(v, c.z, dr[0][k]); di[0][k] = fma(v, c.w, di[0][k]);
                if constexpr (IN2) {
                    cr[1][k] = fma(v, c2.x, cr[1][k]); ci[1][k] = fma(v, c2.y, ci[1][k]);
                    dr[1][k] = fma(v, c2.z, dr[1][k]); di[1][k] = fma(v, c2.w, di[1][k]);
                }
                rec0_step_careful(r[k], c_ab.x, c_ab.y);
            }
        }
        pf.drain();
        if constexpr (IN2) pf2.drain();
    }
    // F_north = C + x D, F_south = C - x D  -> LDS tile -> ring-major global phase array
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int rl = k * 64 + lane;
            const int ip = min(g * RG + rl, P.npairs - 1);
            const double x = P.cth[ip];
            double *t = tile + n * (RG * 16) + rl * 16 + wave * 4;
            t[0] = fma(x, dr[n][k], cr[n][k]); t[1] = fma(x, di[n][k], ci[n][k]);
            t[2] = fma(-x, dr[n][k], cr[n][k]); t[3] = fma(-x, di[n][k], ci[n][k]);
        }
    __syncthreads();
    const int ncomp = NI * (int)gridDim.y;
    for (int c = threadIdx.x; c < NI * RG * 8; c += 256) {
        const int n = c / (RG * 8), cc = c - n * (RG * 8);
        const int rl = cc >> 3, part = cc & 7;
        const int ip = g * RG + rl;
        if (ip < P.npairs) {
            double2 v = *reinterpret_cast<const double2 *>(tile + n * (RG * 16) + rl * 16 + part * 2);
            *reinterpret_cast<double2 *>(phase + (((int64_t)ip * ncomp + NI * blockIdx.y + n) * P.mstride + 4 * mg) * 4 + part * 2) = v;
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// synthesis, spin s: LDS tile entry = {Q_N re, im, Q_S re, im, U_N re, im, U_S re, im}; global phase array [pair][Q | U][m][4]
// -----------------------------------------------------------------------------------------------------
// GONLY: the curl input is identically zero (gradient legs of the temperature estimators): An = sg Ap, so the four sums
// are combinations of Sn Ap and Sp Ap split by the parity of l -- 4 accumulation FMAs per step instead of 8.
// IN2 = 1 ("pair"): a second, gradient-only input (prep2) rides on the same recursion: its phase values are components 2 and 3
// of a four-component phase array.  16 instead of 12 + 8 accumulation + recursion FMAs per step for the two transforms (the
// spin-1 legs of the MV estimator: gradient of T^WF and the spin-1 leg of P^WF).
// IN2 = 2 ("batch"): the second input is a general one (G2, C2) -- the same transform of a second simulation: 4 + 8 + 8 = 20 FMAs
// per step for two maps instead of 24 (batching independent maps through one recursion).  The sums of each
// input are formed in the same order as by the single-input kernel: the results are bit-identical.
template <int R, bool GONLY, int IN2 = 0>
__global__ __launch_bounds__(256) void k_leg_synths(DevPlan P, DevSpinTab S, int spin, const double4 *__restrict__ prep_,
                                                    double *__restrict__ phase, const double4 *__restrict__ prep2_ = nullptr,
                                                    int bstride = 1)
{
    // batch entry blockIdx.y: prep arrays bstride entries apart (2 when the second input of a paired launch is the next entry of
    // the same block vector: entries 2 y and 2 y + 1 share the recursion)
    const double4 *__restrict__ prep = prep_ + (int64_t)blockIdx.y * bstride * S.off[P.mmax + 1];
    const double4 *__restrict__ prep2 = prep2_ ? prep2_ + (int64_t)blockIdx.y * bstride * S.off[P.mmax + 1] : nullptr;
    static_assert(!(GONLY && IN2 == 2), "the batch form takes two general inputs");  // GONLY with IN2 = 1: two gradient-only inputs (4 + 4 + 4 FMAs per step)
    constexpr bool PAIR = IN2 == 1, BATCH = IN2 == 2;
    constexpr int EST = IN2 ? 16 : 8;  // 4 doubles x number of components of the phase array
    constexpr int RG = 64 * R;
    __shared__ double tile[RG * 32];  // [ring][m_local 4][8]
    const int wave = wave_id();
    const int lane = threadIdx.x & 63;
    const int ngroups = (P.npairs + RG - 1) / RG;
    // workgroups are dealt round-robin over the 8 XCDs: rotate the ring group with the m group so that every XCD
    // sees all latitudes (polar groups are light, equatorial ones heavy) -- otherwise the XCDs finish unevenly
    const int mg = P.mg0 + P.mgstride * (blockIdx.x / ngroups), g = (blockIdx.x % ngroups + mg) % ngroups;  // (m-group shard of the plan: device_plan.h)
    const int last = min(P.npairs - 1, g * RG + RG - 1);
    if (S.mlim[last] < 4 * mg) return;
    const int m = 4 * mg + wave;
    const int l0 = m > spin ? m : spin;

    // X_N = sum Sn An, Y_N = sum Sp Ap, X_S = sum sigma Sp An, Y_S = sum sigma Sn Ap, sigma_l = (-1)^(l+m).
    // The mirror sums are kept per parity of i = l - l0 (xe / xo, ye / yo) so that no sign multiply is needed.
    double xn_r[R], xn_i[R], yn_r[R], yn_i[R];
    double xe_r[R], xe_i[R], ye_r[R], ye_i[R], xo_r[R], xo_i[R], yo_r[R], yo_i[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        xn_r[k] = xn_i[k] = yn_r[k] = yn_i[k] = 0.0;
        xe_r[k] = xe_i[k] = ye_r[k] = ye_i[k] = xo_r[k] = xo_i[k] = yo_r[k] = yo_i[k] = 0.0;
    }
    // PAIR: sums of the second input, Sp Ap2 (gx) and Sn Ap2 (gy) by the parity of l, as in the GONLY kernel
    double gxe_r[R], gxe_i[R], gye_r[R], gye_i[R], gxo_r[R], gxo_i[R], gyo_r[R], gyo_i[R];
#pragma unroll
    for (int k = 0; k < R; ++k) gxe_r[k] = gxe_i[k] = gye_r[k] = gye_i[k] = gxo_r[k] = gxo_i[k] = gyo_r[k] = gyo_i[k] = 0.0;
    // BATCH: the twelve sums of the second general input
    double zn_r[R], zn_i[R], wn_r[R], wn_i[R], ze_r[R], ze_i[R], we_r[R], we_i[R], zo_r[R], zo_i[R], wo_r[R], wo_i[R];
#pragma unroll
    for (int k = 0; k < R; ++k) zn_r[k] = zn_i[k] = wn_r[k] = wn_i[k] = ze_r[k] = ze_i[k] = we_r[k] = we_i[k] = zo_r[k] = zo_i[k] = wo_r[k] = wo_i[k] = 0.0;
    double sig0 = 1.0;

    if (m <= P.mmax && l0 <= P.lmax) {
        RecS r[R];
        sig0 = ((l0 + m) & 1) ? -1.0 : 1.0;
        const double fn = S.seedfac_n[m], fp = S.seedfac_p[m];
        const int psin = S.psin[m], phalf = S.phalf[m], ucn = S.usecos_n[m], ucp = S.usecos_p[m];
        const bool seeded = S.seed_syn.rg == RG;  // the table of states at the end of phase A (device_plan.h DevSeedTab)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int ip = g * RG + k * 64 + lane;
            const int ipc = min(ip, P.npairs - 1);
            const bool ok = ip < P.npairs && m <= S.mlim[ipc];
            if (seeded) seed_loads(S.seed_syn, m, ip, P.cth[ipc], r[k]);
            else recs_init(r[k], fn, fp, psin, phalf, ucn, ucp, P.cth[ipc], P.sth[ipc], P.chalf[ipc], P.shalf[ipc], ok);
        }
        const int nl = P.lmax - l0 + 1;
        const int64_t base = S.off[m];
        const double2 *__restrict__ ab = reinterpret_cast<const double2 *>(S.ab) + base;
        const double4 *__restrict__ aa = prep + base;
        const d2v_t *__restrict__ abv = reinterpret_cast<const d2v_t *>(ab);
        const d4v_t *__restrict__ aav = reinterpret_cast<const d4v_t *>(aa);
        const double4 *__restrict__ aa2 = IN2 ? prep2 + base : nullptr;  // pair: gradient-only prep {sg Ap2, Ap2}, only Ap2 is read; batch: {An2, Ap2}
        const d4v_t *__restrict__ aav2 = reinterpret_cast<const d4v_t *>(aa2);
        int i = seeded ? __builtin_amdgcn_readfirstlane(S.seed_syn.il[m * ngroups + g]) : 0;
        bool all_done = false;
        {
            bool live = false;
#pragma unroll
            for (int k = 0; k < R; ++k) live = live || r[k].scn != kNeverActive || r[k].scp != kNeverActive;
            if (!wave_any(live)) i = nl;  // every ring of this wave is pruned for this m
        }
        StreamPrefetch pf;
        pf.start<32, 16>(aa, ab, i, nl, lane);
        double mn[R], mp[R];  // phase B: 0/1 masks of the (n, p) recursions of each ring
        // two consecutive l (even i, odd i); i stays even through phases A, B and C
        // g0, g1: the entries of the second input for the two l -- pair: (Ap2.re, Ap2.im, -, -); batch: (An2, Ap2) like a0, a1
        auto pair_step = [&](auto masked, double ca0, double cb0, double ca1, double cb1, const d4v_t &a0, const d4v_t &a1,
                             const d4v_t &g0 = d4v_t{0., 0., 0., 0.}, const d4v_t &g1 = d4v_t{0., 0., 0., 0.}) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                double vn = r[k].n1, vp = r[k].p1;
                if constexpr (decltype(masked)::value) { vn *= mn[k]; vp *= mp[k]; }
                if constexpr (GONLY) {  // xe / xo hold sum Sp Ap, ye / yo sum Sn Ap (even / odd l)
                    xe_r[k] = fma(vp, a0.z, xe_r[k]); xe_i[k] = fma(vp, a0.w, xe_i[k]);
                } else {
                    xn_r[k] = fma(vn, a0.x, xn_r[k]); xn_i[k] = fma(vn, a0.y, xn_i[k]);
                    yn_r[k] = fma(vp, a0.z, yn_r[k]); yn_i[k] = fma(vp, a0.w, yn_i[k]);
                    xe_r[k] = fma(vp, a0.x, xe_r[k]); xe_i[k] = fma(vp, a0.y, xe_i[k]);
                }
                ye_r[k] = fma(vn, a0.z, ye_r[k]); ye_i[k] = fma(vn, a0.w, ye_i[k]);
                if constexpr (PAIR) {
                    gxe_r[k] = fma(vp, g0.x, gxe_r[k]); gxe_i[k] = fma(vp, g0.y, gxe_i[k]);
                    gye_r[k] = fma(vn, g0.x, gye_r[k]); gye_i[k] = fma(vn, g0.y, gye_i[k]);
                }
                if constexpr (BATCH) {
                    zn_r[k] = fma(vn, g0.x, zn_r[k]); zn_i[k] = fma(vn, g0.y, zn_i[k]);
                    wn_r[k] = fma(vp, g0.z, wn_r[k]); wn_i[k] = fma(vp, g0.w, wn_i[k]);
                    ze_r[k] = fma(vp, g0.x, ze_r[k]); ze_i[k] = fma(vp, g0.y, ze_i[k]);
                    we_r[k] = fma(vn, g0.z, we_r[k]); we_i[k] = fma(vn, g0.w, we_i[k]);
                }
                recs_step_fast(r[k], ca0, cb0);
                vn = r[k].n1; vp = r[k].p1;
                if constexpr (decltype(masked)::value) { vn *= mn[k]; vp *= mp[k]; }
                if constexpr (GONLY) {
                    xo_r[k] = fma(vp, a1.z, xo_r[k]); xo_i[k] = fma(vp, a1.w, xo_i[k]);
                } else {
                    xn_r[k] = fma(vn, a1.x, xn_r[k]); xn_i[k] = fma(vn, a1.y, xn_i[k]);
                    yn_r[k] = fma(vp, a1.z, yn_r[k]); yn_i[k] = fma(vp, a1.w, yn_i[k]);
                    xo_r[k] = fma(vp, a1.x, xo_r[k]); xo_i[k] = fma(vp, a1.y, xo_i[k]);
                }
                yo_r[k] = fma(vn, a1.z, yo_r[k]); yo_i[k] = fma(vn, a1.w, yo_i[k]);
                if constexpr (PAIR) {
                    gxo_r[k] = fma(vp, g1.x, gxo_r[k]); gxo_i[k] = fma(vp, g1.y, gxo_i[k]);
                    gyo_r[k] = fma(vn, g1.x, gyo_r[k]); gyo_i[k] = fma(vn, g1.y, gyo_i[k]);
                }
                if constexpr (BATCH) {
                    zn_r[k] = fma(vn, g1.x, zn_r[k]); zn_i[k] = fma(vn, g1.y, zn_i[k]);
                    wn_r[k] = fma(vp, g1.z, wn_r[k]); wn_i[k] = fma(vp, g1.w, wn_i[k]);
                    zo_r[k] = fma(vp, g1.x, zo_r[k]); zo_i[k] = fma(vp, g1.y, zo_i[k]);
                    wo_r[k] = fma(vn, g1.z, wo_r[k]); wo_i[k] = fma(vn, g1.w, wo_i[k]);
                }
                recs_step_fast(r[k], ca1, cb1);
            }
        };
        // phase A: no lane active yet -- recursion only, rescale check deferred to the end of each block of 8 l
        while (i + 8 <= nl) {
            bool act = false;
#pragma unroll
            for (int k = 0; k < R; ++k) act = act || recs_counts(r[k], kActSynthS);
            if (wave_any(act)) break;
            pf.step<32, 16>(aa, ab, i, nl, lane);
            const d8v_t c0 = ld8(ab + i), c1 = ld8(ab + i + 4);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double ca = t < 4 ? c0[2 * t] : c1[2 * (t - 4)], cb = t < 4 ? c0[2 * t + 1] : c1[2 * (t - 4) + 1];
#pragma unroll
                for (int k = 0; k < R; ++k) recs_step_fast(r[k], ca, cb);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) recs_renorm_up(r[k]);
            i += 8;
        }
        // phase B: mixed -- fast steps with per-ring 0/1 masks, masks and scales refreshed every 8 l
        while (i + 8 <= nl) {
            bool done = true;
#pragma unroll
            for (int k = 0; k < R; ++k)
                done = done && (r[k].scn == 0 || r[k].scn == kNeverActive) && (r[k].scp == 0 || r[k].scp == kNeverActive);
            if (wave_all(done)) { all_done = true; break; }
#pragma unroll
            for (int k = 0; k < R; ++k) { mn[k] = r[k].scn == 0 ? 1.0 : 0.0; mp[k] = r[k].scp == 0 ? 1.0 : 0.0; }
            pf.step<32, 16>(aa, ab, i, nl, lane);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const d8v_t c = ld8(ab + i + 4 * h);
                const d8v_t q0 = ldc<d8v_t>(aav + i + 4 * h), q1 = ldc<d8v_t>(aav + i + 4 * h + 2);
                d8v_t p0 = q0, p1 = q1;  // the same four entries of the second input (pair: only z, w are used)
                if constexpr (IN2 != 0) { p0 = ldc<d8v_t>(aav2 + i + 4 * h); p1 = ldc<d8v_t>(aav2 + i + 4 * h + 2); }
                auto in2 = [](const d8v_t &p, int o) -> d4v_t {  // entry o (0 or 4) of p in the form pair_step takes
                    if constexpr (BATCH) return d4v_t{p[o], p[o + 1], p[o + 2], p[o + 3]};
                    else return d4v_t{p[o + 2], p[o + 3], 0., 0.};
                };
                pair_step(std::true_type(), c[0], c[1], c[2], c[3], __builtin_shufflevector(q0, q0, 0, 1, 2, 3),
                          __builtin_shufflevector(q0, q0, 4, 5, 6, 7), in2(p0, 0), in2(p0, 4));
                pair_step(std::true_type(), c[4], c[5], c[6], c[7], __builtin_shufflevector(q1, q1, 0, 1, 2, 3),
                          __builtin_shufflevector(q1, q1, 4, 5, 6, 7), in2(p1, 0), in2(p1, 4));
            }
#pragma unroll
            for (int k = 0; k < R; ++k) recs_renorm_up(r[k]);
            i += 8;
        }
        // phase C: pure FMA stream, two l per pair_step
        if (all_done && i + 1 < nl) {
            // Software pipeline with two coefficient sets (A, B), two l per set: the wave-uniform loads of the next
            // pair are issued before the FMA stream of the current pair.  SMEM returns out of order, so every set is
            // waited for with lgkmcnt(0) before the other one is issued -- after ~50 v_fma_f64, i.e. hidden.
            // The sched_barriers keep hipcc from sinking the loads to their first use.
            d2v_t A0 = ldc<d2v_t>(abv + i), A1 = ldc<d2v_t>(abv + i + 1);
            // gradient-only: only the second half (Ap) of an entry is used -- load just that, so that the register
            // allocator does not overlap the dead halves of two loads (which costs a wait between their issues)
            auto lda = [&](int idx) -> d4v_t {
                if constexpr (GONLY) {
                    const d2v_t h = ldc<d2v_t>(reinterpret_cast<const d2v_t *>(aav + idx) + 1);
                    return d4v_t{0., 0., h.x, h.y};
                } else {
                    return ldc<d4v_t>(aav + idx);
                }
            };
            auto ldg = [&](int idx) -> d4v_t {  // entry idx of the second input: pair (Ap2, -), batch (An2, Ap2)
                if constexpr (PAIR) {
                    const d2v_t h = ldc<d2v_t>(reinterpret_cast<const d2v_t *>(aav2 + idx) + 1);
                    return d4v_t{h.x, h.y, 0., 0.};
                } else if constexpr (BATCH) {
                    return ldc<d4v_t>(aav2 + idx);
                } else {
                    return d4v_t{0., 0., 0., 0.};
                }
            };
            d4v_t Aa0 = lda(i), Aa1 = lda(i + 1);
            d4v_t Ag0 = ldg(i), Ag1 = ldg(i + 1);
            while (i + 3 < nl) {
                pf.step<32, 16>(aa, ab, i, nl, lane);
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): set A has landed (it was issued one half trip ago) ...
                const d2v_t B0 = ldc<d2v_t>(abv + i + 2), B1 = ldc<d2v_t>(abv + i + 3);  // ... so that B can be issued without A's uses waiting on it
                const d4v_t Ba0 = lda(i + 2), Ba1 = lda(i + 3);
                const d4v_t Bg0 = ldg(i + 2), Bg1 = ldg(i + 3);
                __builtin_amdgcn_sched_barrier(0);
                pair_step(std::false_type(), A0.x, A0.y, A1.x, A1.y, Aa0, Aa1, Ag0, Ag1);
                __builtin_amdgcn_sched_barrier(0);
                const int ip = min(i + 4, nl - 2);  // clamped: the last prefetch re-reads valid entries
                __builtin_amdgcn_s_waitcnt(0xC07F);
                A0 = ldc<d2v_t>(abv + ip); A1 = ldc<d2v_t>(abv + ip + 1); Aa0 = lda(ip); Aa1 = lda(ip + 1); Ag0 = ldg(ip); Ag1 = ldg(ip + 1);
                __builtin_amdgcn_sched_barrier(0);
                pair_step(std::false_type(), B0.x, B0.y, B1.x, B1.y, Ba0, Ba1, Bg0, Bg1);
                __builtin_amdgcn_sched_barrier(0);
                i += 4;
            }
            if (i + 1 < nl) {
                pair_step(std::false_type(), A0.x, A0.y, A1.x, A1.y, Aa0, Aa1, Ag0, Ag1);
                i += 2;
            }
        }
        // tail (at most 7 l, any mix of active and scaled lanes): one careful step at a time
        for (; i < nl; ++i) {
            const double2 c_ab = ab[i];
            const double4 a = aa[i];
            const bool odd = (i & 1) != 0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double vn = recs_value_n(r[k]), vp = recs_value_p(r[k]);
                const double px = GONLY ? a.z : a.x, py = GONLY ? a.w : a.y;  // what Sp multiplies in the parity sums
                if constexpr (!GONLY) {
                    xn_r[k] = fma(vn, a.x, xn_r[k]); xn_i[k] = fma(vn, a.y, xn_i[k]);
                    yn_r[k] = fma(vp, a.z, yn_r[k]); yn_i[k] = fma(vp, a.w, yn_i[k]);
                }
                if (!odd) {
                    xe_r[k] = fma(vp, px, xe_r[k]); xe_i[k] = fma(vp, py, xe_i[k]);
                    ye_r[k] = fma(vn, a.z, ye_r[k]); ye_i[k] = fma(vn, a.w, ye_i[k]);
                } else {
                    xo_r[k] = fma(vp, px, xo_r[k]); xo_i[k] = fma(vp, py, xo_i[k]);
                    yo_r[k] = fma(vn, a.z, yo_r[k]); yo_i[k] = fma(vn, a.w, yo_i[k]);
                }
                if constexpr (PAIR) {
                    const double4 a2 = aa2[i];
                    if (!odd) {
                        gxe_r[k] = fma(vp, a2.z, gxe_r[k]); gxe_i[k] = fma(vp, a2.w, gxe_i[k]);
                        gye_r[k] = fma(vn, a2.z, gye_r[k]); gye_i[k] = fma(vn, a2.w, gye_i[k]);
                    } else {
                        gxo_r[k] = fma(vp, a2.z, gxo_r[k]); gxo_i[k] = fma(vp, a2.w, gxo_i[k]);
                        gyo_r[k] = fma(vn, a2.z, gyo_r[k]); gyo_i[k] = fma(vn, a2.w, gyo_i[k]);
                    }
                }
                if constexpr (BATCH) {
                    const double4 a2 = aa2[i];
                    zn_r[k] = fma(vn, a2.x, zn_r[k]); zn_i[k] = fma(vn, a2.y, zn_i[k]);
                    wn_r[k] = fma(vp, a2.z, wn_r[k]); wn_i[k] = fma(vp, a2.w, wn_i[k]);
                    if (!odd) {
                        ze_r[k] = fma(vp, a2.x, ze_r[k]); ze_i[k] = fma(vp, a2.y, ze_i[k]);
                        we_r[k] = fma(vn, a2.z, we_r[k]); we_i[k] = fma(vn, a2.w, we_i[k]);
                    } else {
                        zo_r[k] = fma(vp, a2.x, zo_r[k]); zo_i[k] = fma(vp, a2.y, zo_i[k]);
                        wo_r[k] = fma(vn, a2.z, wo_r[k]); wo_i[k] = fma(vn, a2.w, wo_i[k]);
                    }
                }
                recs_step_careful(r[k], c_ab.x, c_ab.y);
            }
        }
        pf.drain();
    }
    // Q = X + Y, U = i (Y - X) -> LDS tile -> ring-major global phase array (entries of EST doubles, this transform at `off`)
    auto emit = [&](int k, double xnr, double xni, double ynr, double yni, double xer, double xei, double xor_, double xoi,
                    double yer, double yei, double yor, double yoi) {
        double *t = tile + (k * 64 + lane) * 32 + wave * 8;
        const double xs_r = sig0 * (xer - xor_), xs_i = sig0 * (xei - xoi);
        const double ys_r = sig0 * (yer - yor), ys_i = sig0 * (yei - yoi);
        t[0] = xnr + ynr; t[1] = xni + yni;
        t[2] = xs_r + ys_r; t[3] = xs_i + ys_i;
        t[4] = -(yni - xni); t[5] = ynr - xnr;
        t[6] = -(ys_i - xs_i); t[7] = ys_r - xs_r;
    };
    auto store = [&](int off) {
        __syncthreads();
        for (int c = threadIdx.x; c < RG * 16; c += 256) {
            const int rl = c >> 4, part = c & 15;
            const int ip = g * RG + rl;
            if (ip < P.npairs) {
                const double2 v = *reinterpret_cast<const double2 *>(tile + rl * 32 + part * 2);
                // part = 4 m_local + (pair of doubles inside the 8-double block)
                // phase array [ring pair][component][m][4]: components (off / 4) + 0 (doubles 0-3 of the block) and + 1 (4-7)
                *reinterpret_cast<double2 *>(phase + (((((int64_t)ip * gridDim.y + blockIdx.y) * (EST / 4) + (off >> 2) + ((part >> 1) & 1)) * P.mstride) + 4 * mg + (part >> 2)) * 4 +
                                             (part & 1) * 2) = v;
            }
        }
    };
    const double sg = (spin & 1) ? -1.0 : 1.0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if constexpr (GONLY)  // An = sg Ap: X_N = sg sum Sn Ap, Y_N = sum Sp Ap, X_S = sg sum sigma Sp Ap, Y_S = sum sigma Sn Ap
            emit(k, sg * (ye_r[k] + yo_r[k]), sg * (ye_i[k] + yo_i[k]), xe_r[k] + xo_r[k], xe_i[k] + xo_i[k],
                 sg * xe_r[k], sg * xe_i[k], sg * xo_r[k], sg * xo_i[k], ye_r[k], ye_i[k], yo_r[k], yo_i[k]);
        else
            emit(k, xn_r[k], xn_i[k], yn_r[k], yn_i[k], xe_r[k], xe_i[k], xo_r[k], xo_i[k], ye_r[k], ye_i[k], yo_r[k], yo_i[k]);
    }
    store(0);
    if constexpr (PAIR) {  // the gradient-only input, same formulas as GONLY, into components 2 and 3 of the phase array
        __syncthreads();
#pragma unroll
        for (int k = 0; k < R; ++k)
            emit(k, sg * (gye_r[k] + gyo_r[k]), sg * (gye_i[k] + gyo_i[k]), gxe_r[k] + gxo_r[k], gxe_i[k] + gxo_i[k],
                 sg * gxe_r[k], sg * gxe_i[k], sg * gxo_r[k], sg * gxo_i[k], gye_r[k], gye_i[k], gyo_r[k], gyo_i[k]);
        store(8);
    }
    if constexpr (BATCH) {  // the second general input, same formulas as the first, into components 2 and 3
        __syncthreads();
#pragma unroll
        for (int k = 0; k < R; ++k)
            emit(k, zn_r[k], zn_i[k], wn_r[k], wn_i[k], ze_r[k], ze_i[k], zo_r[k], zo_i[k], we_r[k], we_i[k], wo_r[k], wo_i[k]);
        store(8);
    }
}

// -----------------------------------------------------------------------------------------------------
// cross-lane transpose-reduce: every lane holds v[0..63]; on return lane L holds sum over lanes of v[L]
// -----------------------------------------------------------------------------------------------------
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

// gfx950 v_permlane32_swap / v_permlane16_swap: exchange the upper half (odd 16-lane rows) of `a` with the lower
// half (even rows) of `b`.  Afterwards a + b is, in the lower lanes, own a + partner's a and, in the upper lanes,
// partner's b + own b: one transpose-reduce step without any select.
__device__ __forceinline__ double swap_add32(double a, double b)
{
    const v2u_t lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const v2u_t hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi.x, lo.x) + __hiloint2double(hi.y, lo.y);
}
__device__ __forceinline__ double swap_add16(double a, double b)
{
    const v2u_t lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const v2u_t hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi.x, lo.x) + __hiloint2double(hi.y, lo.y);
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int HALF>
__device__ __forceinline__ void reduce_step(double *v, int lane)
{
    const bool up = (lane & HALF) != 0;  // the lane-id bit handled by this step equals the half size
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const double keep = up ? v[i + HALF] : v[i];
        const double send = up ? v[i] : v[i + HALF];
        double recv;
        if (HALF == 8) recv = dpp_move<0x128>(send);       // row_ror:8  = lane ^ 8 inside a row of 16
        else if (HALF == 2) recv = dpp_move<0x4E>(send);   // quad_perm [2,3,0,1] = lane ^ 2
        else if (HALF == 1) recv = dpp_move<0xB1>(send);   // quad_perm [1,0,3,2] = lane ^ 1
        else recv = __shfl_xor(send, HALF, 64);
        v[i] = keep + recv;
    }
}

// The transpose-reduce of the analysis kernels, in two parts so that only one double per l stays live:
// fold4: the 4 per-lane sums (a0..a3) of one l are folded over the 4 rows of 16 lanes as soon as they exist;
//        afterwards row q (lanes 16 q .. 16 q + 15) holds component {a0, a2, a1, a3}[q], still spread over its 16 lanes.
// reduce16: 16 such values (one per l of the tile) are transposed-reduced inside the rows; on return lane 16 q + j
//        holds the wave total of component {0, 2, 1, 3}[q] of the j-th l of the tile.
__device__ __forceinline__ double fold4(double a0, double a1, double a2, double a3)
{
    return swap_add16(swap_add32(a0, a1), swap_add32(a2, a3));
}
__device__ __forceinline__ double reduce16(double *v, int lane)
{
    reduce_step<8>(v, lane);
    reduce_step<4>(v, lane);
    reduce_step<2>(v, lane);
    reduce_step<1>(v, lane);
    return v[0];
}
// Same result as reduce16 through a wave-private LDS transpose: every lane stores its 16 values, then lane 16 q + j adds
// the 16 entries of value j that belong to lane row q.  15 v_add_f64 instead of ~105 VALU instructions of selects and DPP
// moves.  Layout (conflict-free for both the ds_write_b64 stores and the ds_read_b128 loads, checked with
// SQ_LDS_BANK_CONFLICT): value t of lane L at row (t & 7) of 130 doubles, column 32 (L >> 4) + 16 (t >> 3) + (L & 15).
constexpr int kRedStride = 130;
constexpr int kRedDoubles = 8 * kRedStride;  // per wave
__device__ __forceinline__ double reduce16_lds(const double *v, int lane, double *scratch)
{
    double *w = scratch + 32 * (lane >> 4) + (lane & 15);
#pragma unroll
    for (int t = 0; t < 16; ++t) w[(t & 7) * kRedStride + 16 * (t >> 3)] = v[t];
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): own stores landed (wave-private region, no barrier needed)
    __builtin_amdgcn_wave_barrier();
    const int j = lane & 15;
    const double2 *row = reinterpret_cast<const double2 *>(scratch + (j & 7) * kRedStride + 32 * (lane >> 4) + 16 * (j >> 3));
    double2 s = row[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) { const double2 u = row[k]; s.x += u.x; s.y += u.y; }
    __builtin_amdgcn_wave_barrier();  // all reads issued before the next tile overwrites the region
    return s.x + s.y;
}
__device__ __forceinline__ int fold4_component(int lane) { const int q = lane >> 4; return ((q & 1) << 1) | (q >> 1); }

// -----------------------------------------------------------------------------------------------------
// analysis, spin 0: partial[g][entry] = {C_re, C_im, D_re, D_im} summed over the ring pairs of group g
// -----------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_leg_anal0(DevPlan P, const double *__restrict__ phase, double *__restrict__ partial_)
{
    constexpr int RG = 64 * R;
    constexpr int T = 16;
    constexpr int kTile = RG * 16 > 4 * kRedDoubles ? RG * 16 : 4 * kRedDoubles;
    __shared__ __attribute__((aligned(16))) double tile[kTile];  // phase values of the ring group; reused as the reduce scratch of the 4 waves
    const int wave = wave_id();
    const int lane = threadIdx.x & 63;
    const int ngroups = (P.npairs + RG - 1) / RG;
    double *__restrict__ partial = partial_ + (int64_t)blockIdx.y * ngroups * P.nent0 * 4;
    // workgroups are dealt round-robin over the 8 XCDs: rotate the ring group with the m group so that every XCD
    // sees all latitudes (polar groups are light, equatorial ones heavy) -- otherwise the XCDs finish unevenly
    const int mg = P.mg0 + P.mgstride * (blockIdx.x / ngroups), g = (blockIdx.x % ngroups + mg) % ngroups;  // (m-group shard of the plan: device_plan.h)
    const int last = min(P.npairs - 1, g * RG + RG - 1);
    if (P.mlim0[last] < 4 * mg) return;
    const int m = 4 * mg + wave;

    for (int c = threadIdx.x; c < RG * 8; c += 256) {
        const int rl = c >> 3, part = c & 7;
        const int ip = g * RG + rl;
        double2 v = make_double2(0., 0.);
        if (ip < P.npairs) v = *reinterpret_cast<const double2 *>(phase + (((int64_t)ip * gridDim.y + blockIdx.y) * P.mstride + 4 * mg) * 4 + part * 2);
        *reinterpret_cast<double2 *>(tile + rl * 16 + part * 2) = v;
    }
    __syncthreads();
    if (m > P.mmax) return;

    Rec0 r[R];
    double er[R], ei[R], orr[R], oi[R];
    const double seed = P.seed0[m];
    const bool seeded = P.seed_ana0.rg == RG;  // the plan's table of states at the end of phase A (device_plan.h DevSeedTab)
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int rl = k * 64 + lane;
        const int ip = g * RG + rl;
        const int ipc = min(ip, P.npairs - 1);
        const bool ok = ip < P.npairs && m <= P.mlim0[ipc];
        const double x = P.cth[ipc];
        if (seeded) seed_load0(P.seed_ana0, m, ip, x, r[k]);
        else rec0_init(r[k], seed, m, x, P.sth[ipc], ok);
        const double *t = tile + rl * 16 + wave * 4;
        const double nr = ok ? t[0] : 0., ni = ok ? t[1] : 0., sr = ok ? t[2] : 0., si = ok ? t[3] : 0.;
        er[k] = nr + sr; ei[k] = ni + si;
        orr[k] = (nr - sr) * x; oi[k] = (ni - si) * x;
    }
    __syncthreads();  // every wave has taken its phase values: the tile becomes reduce scratch
    double *scratch = tile + wave * kRedDoubles;
    const int nil = (P.lmax - m) / 2 + 1;
    const int64_t base = P.off0[m];
    const double2 *__restrict__ ab = reinterpret_cast<const double2 *>(P.ab0) + base;
    double *__restrict__ out = partial + ((int64_t)g * P.nent0 + base) * 4;
    bool all_active = false, any_active = false;
    int pfa = -1;  // tile start whose first-half coefficients are already in SA
    d8v_t SA0, SA1, SB0, SB1;
    double acc[16];  // one folded value per l of the tile (see fold4)
    // one half tile (8 consecutive il) with the coefficient set (c0, c1) = 8 (A, B) pairs
    auto half = [&](auto hc, const d8v_t &c0, const d8v_t &c1, int ib) {
        constexpr int h = decltype(hc)::value;
        if (all_active && ib + 8 <= nil) {  // every lane in the IEEE range, all 8 il in range: one branch-free FMA block
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                const int t = 8 * h + tt;
                const double cA = tt < 4 ? c0[2 * tt] : c1[2 * (tt - 4)];
                const double cB = tt < 4 ? c0[2 * tt + 1] : c1[2 * (tt - 4) + 1];
                double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const double v = r[k].p1;
                    if (k == 0) { a0 = v * er[0]; a1 = v * ei[0]; a2 = v * orr[0]; a3 = v * oi[0]; }
                    else { a0 = fma(v, er[k], a0); a1 = fma(v, ei[k], a1); a2 = fma(v, orr[k], a2); a3 = fma(v, oi[k], a3); }
                    rec0_step_fast(r[k], cA, cB);
                }
                acc[t] = fold4(a0, a1, a2, a3);
            }
        } else if (ib + 8 <= nil) {  // mixed: fast steps, each ring's terms times its 0/1 mask; rescale check after the 8 il
            double mk[R];
#pragma unroll
            for (int k = 0; k < R; ++k) mk[k] = r[k].sc == 0 ? 1.0 : 0.0;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                const int t = 8 * h + tt;
                const double cA = tt < 4 ? c0[2 * tt] : c1[2 * (tt - 4)];
                const double cB = tt < 4 ? c0[2 * tt + 1] : c1[2 * (tt - 4) + 1];
                double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const double v = r[k].p1 * mk[k];
                    if (k == 0) { a0 = v * er[0]; a1 = v * ei[0]; a2 = v * orr[0]; a3 = v * oi[0]; }
                    else { a0 = fma(v, er[k], a0); a1 = fma(v, ei[k], a1); a2 = fma(v, orr[k], a2); a3 = fma(v, oi[k], a3); }
                    rec0_step_fast(r[k], cA, cB);
                }
                acc[t] = fold4(a0, a1, a2, a3);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) rec0_renorm_up(r[k]);
        } else {
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                const int t = 8 * h + tt;
                const int il = ib + tt;
                double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
                if (il < nil) {
                    const double cA = tt < 4 ? c0[2 * tt] : c1[2 * (tt - 4)];
                    const double cB = tt < 4 ? c0[2 * tt + 1] : c1[2 * (tt - 4) + 1];
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const double v = rec0_value(r[k]);
                        if (k == 0) { a0 = v * er[0]; a1 = v * ei[0]; a2 = v * orr[0]; a3 = v * oi[0]; }
                    else { a0 = fma(v, er[k], a0); a1 = fma(v, ei[k], a1); a2 = fma(v, orr[k], a2); a3 = fma(v, oi[k], a3); }
                        rec0_step_careful(r[k], cA, cB);
                    }
                }
                acc[t] = fold4(a0, a1, a2, a3);
            }
        }
    };
    {
        bool live = false;
#pragma unroll
        for (int k = 0; k < R; ++k) live = live || (r[k].sc != kNeverActive);
        if (!wave_any(live)) {  // every ring of this wave is pruned for this m: the partial sums are zero
            for (int il0 = 0; il0 < nil; il0 += T)
                if (il0 + (lane >> 2) < nil) out[(int64_t)il0 * 4 + lane] = 0.0;
            return;
        }
    }
    // tiles that the seed table skips (recursion only, no ring of the wave counted yet): their partial sums are zero
    const int il_start = seeded ? __builtin_amdgcn_readfirstlane(P.seed_ana0.il[m * ngroups + g]) : 0;
    for (int il0 = 0; il0 < il_start; il0 += T)
        if (il0 + (lane >> 2) < nil) out[(int64_t)il0 * 4 + lane] = 0.0;
    for (int il0 = il_start; il0 < nil; il0 += T) {
        if (!any_active) {
            // no lane has reached the IEEE range: recursion only.  A lane that activates inside this tile is
            // at 2^-256 then and cannot grow past ~2^-70 within the tile, so the tile's sums are exactly
            // representable as zero at double precision.
            bool act = false;
#pragma unroll
            for (int k = 0; k < R; ++k) act = act || rec0_counts(r[k], kActAnal0);
            any_active = wave_any(act);
            if (!any_active) {
                const int nt = min(T, nil - il0);
                if (nt == T) {  // two blocks of 8 fast steps, rescale check after each (see rec0_renorm_up)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const d8v_t c0 = ld8(ab + il0 + 8 * h), c1 = ld8(ab + il0 + 8 * h + 4);
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const double cA = t < 4 ? c0[2 * t] : c1[2 * (t - 4)], cB = t < 4 ? c0[2 * t + 1] : c1[2 * (t - 4) + 1];
#pragma unroll
                            for (int k = 0; k < R; ++k) rec0_step_fast(r[k], cA, cB);
                        }
#pragma unroll
                        for (int k = 0; k < R; ++k) rec0_renorm_up(r[k]);
                    }
                } else {
                    for (int t = 0; t < nt; ++t) {
                        const double2 c_ab = ab[il0 + t];
#pragma unroll
                        for (int k = 0; k < R; ++k) rec0_step_careful(r[k], c_ab.x, c_ab.y);
                    }
                }
                if (il0 + (lane >> 2) < nil) out[(int64_t)il0 * 4 + lane] = 0.0;
                continue;
            }
        }
        // Two coefficient sets: SA (first 8 il of the tile) and SB (last 8); each is loaded while the other is being
        // consumed, so the wave-uniform load latency hides under ~100 v_fma_f64 (tables are padded: reads past nil are unused)
        if (pfa != il0) { SA0 = ld8(ab + il0); SA1 = ld8(ab + il0 + 4); }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): SA has landed before SB is issued
        SB0 = ld8(ab + il0 + 8); SB1 = ld8(ab + il0 + 12);
        __builtin_amdgcn_sched_barrier(0);
        half(std::integral_constant<int, 0>(), SA0, SA1, il0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        SA0 = ld8(ab + il0 + 16); SA1 = ld8(ab + il0 + 20);
        pfa = il0 + 16;
        __builtin_amdgcn_sched_barrier(0);
        half(std::integral_constant<int, 1>(), SB0, SB1, il0 + 8);
        __builtin_amdgcn_sched_barrier(0);
        if (!all_active) {
            bool done = true;
#pragma unroll
            for (int k = 0; k < R; ++k) done = done && (r[k].sc == 0 || r[k].sc == kNeverActive);
            all_active = wave_all(done);
        }
        const double tot = reduce16_lds(acc, lane, scratch);
        if (il0 + (lane & 15) < nil) out[(int64_t)(il0 + (lane & 15)) * 4 + fold4_component(lane)] = tot;
    }
}

// reduce partials over ring groups and convert (C, D) -> a_lm (fused hp.almxfl)
// add / fl_add (optional): alm = fl * (analysis) + fl_add * add, the S^-1 x term of the CG operator folded in
// lr (optional): the low-rank template update alm -= rm^t c folded in -- c_k = the sum of the nparts partial sums of mode k
// (left by k_tproj_coeffs on the operator's input, summed exactly as k_tproj_apply sums them), subtracted from every entry right after it is formed,
// mode by mode in k_tproj_apply's order: bit-identical to that kernel run afterwards
// the two partial sums of a workgroup of 256 threads -> PostDots (fixed tree: bit-reproducible)
__device__ __forceinline__ void post_dots_emit(double t1, double t2, const PostDots &dots, double *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { t1 += __shfl_down(t1, off, 64); t2 += __shfl_down(t2, off, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave] = t1; red[4 + wave] = t2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t w = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        dots.s1[w] = (red[0] + red[1]) + (red[2] + red[3]);
        dots.s2[w] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}
__device__ __forceinline__ void post_dots_add(double &t1, double &t2, double w, const double2 q, const double2 d, const double2 r)
{
    t1 = fma(w, d.x * q.x + d.y * q.y, t1);
    t2 = fma(w, d.x * r.x + d.y * r.y, t2);
}
struct PostLowRank { int nmodes = 0, nparts = 0, pstride = 0; int64_t bstride = 0; const double *rm = nullptr, *parts = nullptr; };  // bstride: between the partial sums of batch entries

__global__ void k_post0(DevPlan P, int RG, const double4 *__restrict__ partial_, const double *__restrict__ fl, double2 *__restrict__ alm_,
                        const double2 *__restrict__ add_, const double *__restrict__ fl_add, PostLowRank lr, PostDots dots)
{
    __shared__ double lrc[16];
    __shared__ double dred[8];
    double t1 = 0., t2 = 0.;
    if (lr.nmodes > 0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int k = wave; k < lr.nmodes; k += 4) {
            double v = 0.0;
            for (int j = lane; j < lr.nparts; j += 64) v += lr.parts[(int64_t)blockIdx.z * lr.bstride + k * lr.pstride + j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) lrc[k] = v;
        }
        __syncthreads();
    }
    const int m = blockIdx.y;
    const int nil = (P.lmax - m) / 2 + 1;
    const int64_t base = P.off0[m];
    const int64_t abase = (int64_t)m * (2 * P.lmax + 1 - m) / 2;
    const int ngroups = (P.npairs + RG - 1) / RG;
    const double4 *__restrict__ partial = partial_ + (int64_t)blockIdx.z * ngroups * P.nent0;
    double2 *__restrict__ alm = alm_ + (int64_t)blockIdx.z * P.nalm;
    const double2 *__restrict__ add = add_ ? add_ + (int64_t)blockIdx.z * P.nalm : nullptr;
    const int mg4 = 4 * (m / 4);
    // the ring groups this m-group keeps (the analysis kernel wrote partial sums for exactly these): mlim does not decrease towards the equator, so they
    // are the groups from the first kept one on
    int g0 = 0;
    while (g0 < ngroups && P.mlim0[min(P.npairs - 1, g0 * RG + RG - 1)] < mg4) ++g0;
    for (int il = blockIdx.x * blockDim.x + threadIdx.x; il < nil; il += gridDim.x * blockDim.x) {
        const int64_t e = base + il;
        // (the vectors of the scalar products are fetched ahead of the partial sums: their latency hides behind that loop)
        const int l_ = m + 2 * il;
        const int64_t idot = (int64_t)blockIdx.z * P.nalm + abase + l_;
        double2 dd0 = make_double2(0., 0.), dd1 = dd0, rr0 = dd0, rr1 = dd0;
        if (dots.s1) {
            dd0 = reinterpret_cast<const double2 *>(dots.d[0])[idot]; rr0 = reinterpret_cast<const double2 *>(dots.r[0])[idot];
            if (l_ + 1 <= P.lmax) { dd1 = reinterpret_cast<const double2 *>(dots.d[0])[idot + 1]; rr1 = reinterpret_cast<const double2 *>(dots.r[0])[idot + 1]; }
        }
        double c0r = 0., c0i = 0., c1r = 0., c1i = 0., dr = 0., di = 0.;
#pragma unroll 4
        for (int g = g0; g < ngroups; ++g) {  // (loads of four groups in flight; added in group order as before)
            const double4 v = partial[(int64_t)g * P.nent0 + e];
            c0r += v.x; c0i += v.y; dr += v.z; di += v.w;
            if (il > 0) {
                const double4 w = partial[(int64_t)g * P.nent0 + e - 1];
                c1r += w.x; c1i += w.y;
            }
        }
        const int l = m + 2 * il;
        const double al = P.alpha0[e], e1 = P.eps0[2 * e];
        double f0 = e1 * al, f1 = 0.;
        if (il > 0) f1 = P.eps0[2 * (e - 1) + 1] * P.alpha0[e - 1];  // eps_l alpha_{l-2}
        double2 a;
        a.x = f0 * c0r + f1 * c1r;
        a.y = f0 * c0i + f1 * c1i;
        if (fl) { a.x *= fl[l]; a.y *= fl[l]; }
        if (add) { const double2 t = add[abase + l]; a.x = fma(fl_add[l], t.x, a.x); a.y = fma(fl_add[l], t.y, a.y); }
        if (lr.nmodes > 0) {
            const int64_t i = 2 * (abase + l), n2 = 2 * P.nalm;
            for (int k = 0; k < lr.nmodes; ++k) { a.x = fma(-lr.rm[(int64_t)k * n2 + i], lrc[k], a.x); a.y = fma(-lr.rm[(int64_t)k * n2 + i + 1], lrc[k], a.y); }
        }
        alm[abase + l] = a;
        const double wdot = m == 0 ? 1.0 : 2.0;
        if (dots.s1 && l >= dots.lmin) post_dots_add(t1, t2, wdot, a, dd0, rr0);
        if (l + 1 <= P.lmax) {
            double2 b;
            b.x = al * dr; b.y = al * di;
            if (fl) { b.x *= fl[l + 1]; b.y *= fl[l + 1]; }
            if (add) { const double2 t = add[abase + l + 1]; b.x = fma(fl_add[l + 1], t.x, b.x); b.y = fma(fl_add[l + 1], t.y, b.y); }
            if (lr.nmodes > 0) {
                const int64_t i = 2 * (abase + l + 1), n2 = 2 * P.nalm;
                for (int k = 0; k < lr.nmodes; ++k) { b.x = fma(-lr.rm[(int64_t)k * n2 + i], lrc[k], b.x); b.y = fma(-lr.rm[(int64_t)k * n2 + i + 1], lrc[k], b.y); }
            }
            alm[abase + l + 1] = b;
            if (dots.s1 && l + 1 >= dots.lmin) post_dots_add(t1, t2, wdot, b, dd1, rr1);
        }
    }
    if (dots.s1) post_dots_emit(t1, t2, dots, dred);
}

// -----------------------------------------------------------------------------------------------------
// analysis, spin s: partial[g][entry] = {G'_re, G'_im, C'_re, C'_im}
// -----------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_leg_anals(DevPlan P, DevSpinTab S, int spin, const double *__restrict__ phase,
                                                   double *__restrict__ partial_, int64_t nent)
{
    constexpr int RG = 64 * R;
    constexpr int T = 16;
    constexpr int kTile = RG * 32 > 4 * kRedDoubles ? RG * 32 : 4 * kRedDoubles;
    __shared__ __attribute__((aligned(16))) double tile[kTile];  // phase values of the ring group; reused as the reduce scratch of the 4 waves
    const int wave = wave_id();
    const int lane = threadIdx.x & 63;
    const int ngroups = (P.npairs + RG - 1) / RG;
    double *__restrict__ partial = partial_ + (int64_t)blockIdx.y * ngroups * nent * 4;
    // workgroups are dealt round-robin over the 8 XCDs: rotate the ring group with the m group so that every XCD
    // sees all latitudes (polar groups are light, equatorial ones heavy) -- otherwise the XCDs finish unevenly
    const int mg = P.mg0 + P.mgstride * (blockIdx.x / ngroups), g = (blockIdx.x % ngroups + mg) % ngroups;  // (m-group shard of the plan: device_plan.h)
    const int last = min(P.npairs - 1, g * RG + RG - 1);
    if (S.mlim[last] < 4 * mg) return;
    const int m = 4 * mg + wave;
    const int l0 = m > spin ? m : spin;

    for (int c = threadIdx.x; c < RG * 16; c += 256) {
        const int rl = c >> 4, part = c & 15;
        const int ip = g * RG + rl;
        double2 v = make_double2(0., 0.);
        if (ip < P.npairs)  // phase array [ring pair][component 2][m][4] -> tile [ring][m_local 4][Q 4 | U 4]
            v = *reinterpret_cast<const double2 *>(phase + (((((int64_t)ip * gridDim.y + blockIdx.y) * 2 + ((part >> 1) & 1)) * P.mstride) + 4 * mg + (part >> 2)) * 4 + (part & 1) * 2);
        *reinterpret_cast<double2 *>(tile + rl * 32 + part * 2) = v;
    }
    __syncthreads();
    if (m > P.mmax || l0 > P.lmax) return;

    RecS r[R];
    // With Wp = Q + iU, Wm = Q - iU and sigma_l = (-1)^(l+m) (mirror ring: Sn <-> sigma Sp):
    //   G'_l = sum Sn (sg Wp_N + sigma Wm_S) + Sp (Wm_N + sg sigma Wp_S)
    //   C'_l = sum Sn (sg Wp_N - sigma Wm_S) - Sp (Wm_N - sg sigma Wp_S)
    // ae = sg Wp_N + Wm_S, ao = sg Wp_N - Wm_S (multiply Sn); be = Wm_N + sg Wp_S, bo = Wm_N - sg Wp_S (multiply Sp)
    //   sigma = +1: G' += Sn ae + Sp be, C' += Sn ao - Sp bo;   sigma = -1: G' += Sn ao + Sp bo, C' += Sn ae - Sp be
    double aer[R], aei[R], aor[R], aoi[R], ber[R], bei[R], bor[R], boi[R];
    const double sg = (spin & 1) ? -1.0 : 1.0;
    const double fn = S.seedfac_n[m], fp = S.seedfac_p[m];
    const int psin = S.psin[m], phalf = S.phalf[m], ucn = S.usecos_n[m], ucp = S.usecos_p[m];
    const bool seeded = S.seed_ana.rg == RG;  // the table of states at the end of phase A (device_plan.h DevSeedTab)
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int rl = k * 64 + lane;
        const int ip = g * RG + rl;
        const int ipc = min(ip, P.npairs - 1);
        const bool ok = ip < P.npairs && m <= S.mlim[ipc];
        if (seeded) seed_loads(S.seed_ana, m, ip, P.cth[ipc], r[k]);
        else recs_init(r[k], fn, fp, psin, phalf, ucn, ucp, P.cth[ipc], P.sth[ipc], P.chalf[ipc], P.shalf[ipc], ok);
        const double *t = tile + rl * 32 + wave * 8;
        // select, not multiply: entries of pruned (m, ring) are never written by the FFT stage and may hold anything
        const double qnr = ok ? t[0] : 0., qni = ok ? t[1] : 0., qsr = ok ? t[2] : 0., qsi = ok ? t[3] : 0.;
        const double unr = ok ? t[4] : 0., uni = ok ? t[5] : 0., usr = ok ? t[6] : 0., usi = ok ? t[7] : 0.;
        // Wp = Q + iU = (q.re - u.im) + i (q.im + u.re);  Wm = Q - iU = (q.re + u.im) + i (q.im - u.re)
        const double wpn_r = sg * (qnr - uni), wpn_i = sg * (qni + unr), wmn_r = qnr + uni, wmn_i = qni - unr;
        const double wps_r = sg * (qsr - usi), wps_i = sg * (qsi + usr), wms_r = qsr + usi, wms_i = qsi - usr;
        aer[k] = wpn_r + wms_r; aei[k] = wpn_i + wms_i; aor[k] = wpn_r - wms_r; aoi[k] = wpn_i - wms_i;
        ber[k] = wmn_r + wps_r; bei[k] = wmn_i + wps_i; bor[k] = wmn_r - wps_r; boi[k] = wmn_i - wps_i;
    }
    if ((l0 + m) & 1) {  // sigma_{l0} = -1: swap the even / odd roles once
#pragma unroll
        for (int k = 0; k < R; ++k) {
            double t_;
            t_ = aer[k]; aer[k] = aor[k]; aor[k] = t_; t_ = aei[k]; aei[k] = aoi[k]; aoi[k] = t_;
            t_ = ber[k]; ber[k] = bor[k]; bor[k] = t_; t_ = bei[k]; bei[k] = boi[k]; boi[k] = t_;
        }
    }
    __syncthreads();  // every wave has taken its phase values: the tile becomes reduce scratch
    double *scratch = tile + wave * kRedDoubles;
    const int nl = P.lmax - l0 + 1;
    const int64_t base = S.off[m];
    const double2 *__restrict__ ab = reinterpret_cast<const double2 *>(S.ab) + base;
    double *__restrict__ out = partial + ((int64_t)g * nent + base) * 4;
    bool all_active = false, any_active = false;
    int pfa = -1;  // tile start whose first coefficient set is already in QA
    d8v_t QA, QB;
    double acc[16];  // one folded value per l of the tile (see fold4)
    // the 8 FMAs of one (l, ring): sigma_l = +1 for even t (after the swap above), -1 for odd t
    // ring k = 0 starts the four sums with a multiply instead of an FMA onto zero (no zero-initialising moves)
    auto accum = [&](int t, int k, double vn, double vp, double &a0, double &a1, double &a2, double &a3) {
        const double *pa0 = (t & 1) == 0 ? aer : aor, *pa1 = (t & 1) == 0 ? aei : aoi, *pa2 = (t & 1) == 0 ? aor : aer, *pa3 = (t & 1) == 0 ? aoi : aei;
        const double *pb0 = (t & 1) == 0 ? ber : bor, *pb1 = (t & 1) == 0 ? bei : boi, *pb2 = (t & 1) == 0 ? bor : ber, *pb3 = (t & 1) == 0 ? boi : bei;
        if (k == 0) { a0 = vn * pa0[0]; a1 = vn * pa1[0]; a2 = vn * pa2[0]; a3 = vn * pa3[0]; }
        else { a0 = fma(vn, pa0[k], a0); a1 = fma(vn, pa1[k], a1); a2 = fma(vn, pa2[k], a2); a3 = fma(vn, pa3[k], a3); }
        a0 = fma(vp, pb0[k], a0); a1 = fma(vp, pb1[k], a1); a2 = fma(-vp, pb2[k], a2); a3 = fma(-vp, pb3[k], a3);
    };
    auto fold = [&](double a0, double a1, double a2, double a3) -> double { return fold4(a0, a1, a2, a3); };
    // one quarter tile: 4 consecutive l with the coefficient set c = 4 (a, b) pairs
    auto quarter = [&](auto qc, const d8v_t &c, int ib) {
        constexpr int h = decltype(qc)::value;
        if (all_active && ib + 4 <= nl) {  // every lane in the IEEE range, all 4 l in range: one branch-free FMA block
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = 4 * h + tt;
                const double cA = c[2 * tt];
                const double cB = c[2 * tt + 1];
                double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    accum(t, k, r[k].n1, r[k].p1, a0, a1, a2, a3);
                    recs_step_fast(r[k], cA, cB);
                }
                acc[t] = fold(a0, a1, a2, a3);
            }
        } else if (ib + 4 <= nl) {  // mixed: fast steps, each ring's terms times its 0/1 masks; rescale check after the 4 l
            double mn[R], mp[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { mn[k] = r[k].scn == 0 ? 1.0 : 0.0; mp[k] = r[k].scp == 0 ? 1.0 : 0.0; }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = 4 * h + tt;
                const double cA = c[2 * tt];
                const double cB = c[2 * tt + 1];
                double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    accum(t, k, r[k].n1 * mn[k], r[k].p1 * mp[k], a0, a1, a2, a3);
                    recs_step_fast(r[k], cA, cB);
                }
                acc[t] = fold(a0, a1, a2, a3);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) recs_renorm_up(r[k]);
        } else {  // last, partial tile: one careful step at a time
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = 4 * h + tt;
                const int i = ib + tt;
                double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
                if (i < nl) {
                    const double cA = c[2 * tt];
                    const double cB = c[2 * tt + 1];
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        accum(t, k, recs_value_n(r[k]), recs_value_p(r[k]), a0, a1, a2, a3);
                        recs_step_careful(r[k], cA, cB);
                    }
                }
                acc[t] = fold(a0, a1, a2, a3);
            }
        }
    };
    {
        bool live = false;
#pragma unroll
        for (int k = 0; k < R; ++k) live = live || r[k].scn != kNeverActive || r[k].scp != kNeverActive;
        if (!wave_any(live)) {  // every ring of this wave is pruned for this m: the partial sums are zero
            for (int i0 = 0; i0 < nl; i0 += T)
                if (i0 + (lane >> 2) < nl) out[(int64_t)i0 * 4 + lane] = 0.0;
            return;
        }
    }
    // tiles that the seed table skips (recursion only, no ring of the wave counted yet): their partial sums are zero
    const int i_start = seeded ? __builtin_amdgcn_readfirstlane(S.seed_ana.il[m * ngroups + g]) : 0;
    for (int i0 = 0; i0 < i_start; i0 += T)
        if (i0 + (lane >> 2) < nl) out[(int64_t)i0 * 4 + lane] = 0.0;
    for (int i0 = i_start; i0 < nl; i0 += T) {
        if (!any_active) {
            bool act = false;
#pragma unroll
            for (int k = 0; k < R; ++k) act = act || recs_counts(r[k], kActAnalS);
            any_active = wave_any(act);
            if (!any_active) {
                const int nt = min(T, nl - i0);
                if (nt == T) {  // two blocks of 8 fast steps, rescale check after each (see rec0_renorm_up)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const d8v_t c0 = ld8(ab + i0 + 8 * h), c1 = ld8(ab + i0 + 8 * h + 4);
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const double ca = t < 4 ? c0[2 * t] : c1[2 * (t - 4)], cb = t < 4 ? c0[2 * t + 1] : c1[2 * (t - 4) + 1];
#pragma unroll
                            for (int k = 0; k < R; ++k) recs_step_fast(r[k], ca, cb);
                        }
#pragma unroll
                        for (int k = 0; k < R; ++k) recs_renorm_up(r[k]);
                    }
                } else {
                    for (int t = 0; t < nt; ++t) {
                        const double2 c_ab = ab[i0 + t];
#pragma unroll
                        for (int k = 0; k < R; ++k) recs_step_careful(r[k], c_ab.x, c_ab.y);
                    }
                }
                if (i0 + (lane >> 2) < nl) out[(int64_t)i0 * 4 + lane] = 0.0;
                continue;
            }
        }
        // Two coefficient sets of 4 l (QA, QB), each loaded while the other is consumed (~240 VALU instructions): with
        // sets of 8 l the four live sets took 64 SGPRs and 50 scalars were spilled to VGPR lanes (v_readlane /
        // v_writelane in the hot loop).  The tables are padded: reads past nl are unused.
        if (pfa != i0) QA = ld8(ab + i0);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): QA has landed before QB is issued
        QB = ld8(ab + i0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        quarter(std::integral_constant<int, 0>(), QA, i0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        QA = ld8(ab + i0 + 8);
        __builtin_amdgcn_sched_barrier(0);
        quarter(std::integral_constant<int, 1>(), QB, i0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        QB = ld8(ab + i0 + 12);
        __builtin_amdgcn_sched_barrier(0);
        quarter(std::integral_constant<int, 2>(), QA, i0 + 8);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        QA = ld8(ab + i0 + 16);
        pfa = i0 + 16;
        __builtin_amdgcn_sched_barrier(0);
        quarter(std::integral_constant<int, 3>(), QB, i0 + 12);
        __builtin_amdgcn_sched_barrier(0);
        if (!all_active) {
            bool done = true;
#pragma unroll
            for (int k = 0; k < R; ++k)
                done = done && (r[k].scn == 0 || r[k].scn == kNeverActive) && (r[k].scp == 0 || r[k].scp == kNeverActive);
            all_active = wave_all(done);
        }
        const double tot = reduce16_lds(acc, lane, scratch);
        if (i0 + (lane & 15) < nl) out[(int64_t)(i0 + (lane & 15)) * 4 + fold4_component(lane)] = tot;
    }
}

// G_l = -1/2 beta_l G'_l,  C_l = i/2 beta_l C'_l
// addG / addC with flG / flC (optional): almG += flG * addG, almC += flC * addC -- the S^-1 x term of the CG operator folded in
__global__ void k_posts(DevPlan P, DevSpinTab S, int spin, int RG, int64_t nent, const double4 *__restrict__ partial_,
                        const double *__restrict__ fl, double2 *__restrict__ almG_, double2 *__restrict__ almC_,
                        const double2 *__restrict__ addG_, const double2 *__restrict__ addC_, const double *__restrict__ flG,
                        const double *__restrict__ flC, PostDots dots)
{
    __shared__ double dred[8];
    double t1 = 0., t2 = 0.;
    const int64_t bz = blockIdx.z;
    const double4 *__restrict__ partial = partial_ + bz * ((P.npairs + RG - 1) / RG) * nent;
    double2 *__restrict__ almG = almG_ + bz * P.nalm, *__restrict__ almC = almC_ + bz * P.nalm;
    const double2 *__restrict__ addG = addG_ ? addG_ + bz * P.nalm : nullptr, *__restrict__ addC = addC_ ? addC_ + bz * P.nalm : nullptr;
    const int m = blockIdx.y;
    const int l0 = m > spin ? m : spin;
    const int64_t abase = (int64_t)m * (2 * P.lmax + 1 - m) / 2;
    const double wdot = m == 0 ? 1.0 : 2.0;
    // entries below the spin are zero
    for (int l = m + blockIdx.x * blockDim.x + threadIdx.x; l < l0 && l <= P.lmax; l += gridDim.x * blockDim.x) {
        double2 g = make_double2(0., 0.), c = g;
        if (addG) {
            const double2 tg = addG[abase + l], tc = addC[abase + l];
            g = make_double2(flG[l] * tg.x, flG[l] * tg.y);
            c = make_double2(flC[l] * tc.x, flC[l] * tc.y);
        }
        almG[abase + l] = g;
        almC[abase + l] = c;
        if (dots.s1 && l >= dots.lmin) {
            const int64_t i = bz * P.nalm + abase + l;
            post_dots_add(t1, t2, wdot, g, reinterpret_cast<const double2 *>(dots.d[0])[i], reinterpret_cast<const double2 *>(dots.r[0])[i]);
            post_dots_add(t1, t2, wdot, c, reinterpret_cast<const double2 *>(dots.d[1])[i], reinterpret_cast<const double2 *>(dots.r[1])[i]);
        }
    }
    const int nl = P.lmax - l0 + 1;
    const int64_t base = S.off[m];
    const int ngroups = (P.npairs + RG - 1) / RG;
    const int mg4 = 4 * (m / 4);
    // the ring groups this m-group keeps (the analysis kernel wrote partial sums for exactly these): mlim does not decrease towards the equator, so they
    // are the groups from the first kept one on
    int g0 = 0;
    while (g0 < ngroups && S.mlim[min(P.npairs - 1, g0 * RG + RG - 1)] < mg4) ++g0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += gridDim.x * blockDim.x) {
        const int64_t e = base + i;
        const int64_t ii = bz * P.nalm + abase + l0 + i;
        double2 dg = make_double2(0., 0.), dc = dg, rg = dg, rc = dg;  // fetched ahead of the partial sums (latency)
        if (dots.s1) {
            dg = reinterpret_cast<const double2 *>(dots.d[0])[ii]; rg = reinterpret_cast<const double2 *>(dots.r[0])[ii];
            dc = reinterpret_cast<const double2 *>(dots.d[1])[ii]; rc = reinterpret_cast<const double2 *>(dots.r[1])[ii];
        }
        double gr = 0., gi = 0., cr = 0., ci = 0.;
#pragma unroll 4
        for (int g = g0; g < ngroups; ++g) {  // (loads of four groups in flight; added in group order as before)
            const double4 v = partial[(int64_t)g * nent + e];
            gr += v.x; gi += v.y; cr += v.z; ci += v.w;
        }
        const int l = l0 + i;
        double f = 0.5 * S.beta[e];
        if (fl) f *= fl[l];
        double2 g = make_double2(-f * gr, -f * gi);   // G = -1/2 beta G'
        double2 c = make_double2(-f * ci, f * cr);    // C = i/2 beta C'
        if (addG) {
            const double2 tg = addG[abase + l], tc = addC[abase + l];
            g.x = fma(flG[l], tg.x, g.x); g.y = fma(flG[l], tg.y, g.y);
            c.x = fma(flC[l], tc.x, c.x); c.y = fma(flC[l], tc.y, c.y);
        }
        almG[abase + l] = g;
        almC[abase + l] = c;
        if (dots.s1 && l >= dots.lmin) {
            post_dots_add(t1, t2, wdot, g, dg, rg);
            post_dots_add(t1, t2, wdot, c, dc, rc);
        }
    }
    if (dots.s1) post_dots_emit(t1, t2, dots, dred);
}

// -----------------------------------------------------------------------------------------------------
// host launchers
// -----------------------------------------------------------------------------------------------------
// ring pairs per lane (tunable at run time for experiments, under PLSHTS_DEBUG=1: PLSHTS_R0 / PLSHTS_RS in the environment)
static int env_int(const char *name, int dflt) { return dbg_env_int(name, dflt); }
// Ring pairs per lane (R): more rings per lane amortise the per-l overhead (coefficient fetch, cross-lane reduce) but
// coarsen the polar pruning and shrink the grid.  The defaults are the measured optima at nside = lmax = 2048; smaller
// transforms step R down until the grid has at least ~4 workgroups per CU.  PLSHTS_R0 / RS / R0A / RSA override.
// m-groups (4 consecutive orders) this plan's Legendre launches cover: all of them, or every mgstride-th from mg0 on a shard plan
static int own_mgroups(const DevPlan &P)
{
    const int all = (P.mmax + 4) / 4;
    return P.mg0 >= all ? 0 : (all - P.mg0 + P.mgstride - 1) / P.mgstride;
}
static int pick_r(const char *env, int dflt, int rmax, const DevPlan &P)
{
    const int e = env_int(env, 0);
    if (e >= 1 && e <= rmax) return e;
    int r = dflt;
    const int nmg = own_mgroups(P);
    while (r > 1 && (int64_t)((P.npairs + 64 * r - 1) / (64 * r)) * nmg < 1024) --r;
    return r;
}
static int r0_synth(const DevPlan &P) { return pick_r("PLSHTS_R0", 3, 6, P); }
static int rs_synth(const DevPlan &P) { return pick_r("PLSHTS_RS", 2, 4, P); }
// (spin-0 analysis: 6 rings per lane.  Before the seed tables 8 won at nside >= 4096 (8.31 against 8.99 ms at nside = lmax = 4096); with them every ring of a
// wave still runs from the wave's first step, and the wider group of R = 8 pays more for that than it saves: 10.3 against 8.8 ms, tools/r_sweep.sh)
static int r0_anal(const DevPlan &P) { return pick_r("PLSHTS_R0A", 6, 8, P); }
static int rs_anal(const DevPlan &P) { return pick_r("PLSHTS_RSA", 4, 4, P); }  // (5, 6, 8 rings per lane -- one wave per SIMD -- measured 5.30 / 6.27 / 10.0 ms against 4.53 ms: round 3)

// partial sums per batch entry that k_post0 / k_posts leave in a PostDots (= their workgroups per entry)
int post_dots_count(const DevPlan &P) { return 4 * (P.mmax + 1); }
int rings_per_group(int spin, const DevPlan &P) { return 64 * (spin == 0 ? r0_anal(P) : rs_anal(P)); }

// ---- seed tables (device_plan.h DevSeedTab) of a kernel family: fam 0 = synthesis, 1 = analysis ------------------------------------------
// ring pairs per wave of the family on this plan (what its launcher will pick)
int seed_family_rg(const DevPlan &P, int spin, int fam)
{
    return 64 * (spin == 0 ? (fam == 0 ? r0_synth(P) : r0_anal(P)) : (fam == 0 ? rs_synth(P) : rs_anal(P)));
}
template <int R>
static void seed_gen0_r(const DevPlan &P, double thr, int gran, int *il, double *st, int *sc, hipStream_t s)
{
    const int ngroups = (P.npairs + 64 * R - 1) / (64 * R), nmg = (P.mmax + 4) / 4;
    hipLaunchKernelGGL(k_seed_gen0<R>, dim3(ngroups * nmg), dim3(256), 0, s, P, thr, gran, il, reinterpret_cast<double2 *>(st), sc);
}
template <int R>
static void seed_gens_r(const DevPlan &P, const DevSpinTab &S, int spin, double thr, int gran, int *il, double *st, int *sc, hipStream_t s)
{
    const int ngroups = (P.npairs + 64 * R - 1) / (64 * R), nmg = (P.mmax + 4) / 4;
    hipLaunchKernelGGL(k_seed_gens<R>, dim3(ngroups * nmg), dim3(256), 0, s, P, S, spin, thr, gran, il, reinterpret_cast<double4 *>(st),
                       reinterpret_cast<int2 *>(sc));
}
// il: (mmax + 1) x ngroups ints; st: (mmax + 1) x npad x (2 | 4) doubles; sc: (mmax + 1) x npad x (1 | 2) ints, npad = ngroups x rg
bool launch_seed_gen(const DevPlan &P, const DevSpinTab *S, int spin, int fam, int rg, int *il, double *st, int *sc, hipStream_t s)
{
    const int gran = fam == 0 ? 8 : 16;  // check interval of the family's kernels, in recursion steps
    const double thr = spin == 0 ? (fam == 0 ? kActSynth0 : kActAnal0) : (fam == 0 ? kActSynthS : kActAnalS);
    const int R = rg / 64;
    if (spin == 0) {
        switch (R) {
        case 1: seed_gen0_r<1>(P, thr, gran, il, st, sc, s); break;
        case 2: seed_gen0_r<2>(P, thr, gran, il, st, sc, s); break;
        case 3: seed_gen0_r<3>(P, thr, gran, il, st, sc, s); break;
        case 4: seed_gen0_r<4>(P, thr, gran, il, st, sc, s); break;
        case 5: seed_gen0_r<5>(P, thr, gran, il, st, sc, s); break;
        case 6: seed_gen0_r<6>(P, thr, gran, il, st, sc, s); break;
        case 7: seed_gen0_r<7>(P, thr, gran, il, st, sc, s); break;
        case 8: seed_gen0_r<8>(P, thr, gran, il, st, sc, s); break;
        default: return false;
        }
    } else {
        switch (R) {
        case 1: seed_gens_r<1>(P, *S, spin, thr, gran, il, st, sc, s); break;
        case 2: seed_gens_r<2>(P, *S, spin, thr, gran, il, st, sc, s); break;
        case 3: seed_gens_r<3>(P, *S, spin, thr, gran, il, st, sc, s); break;
        case 4: seed_gens_r<4>(P, *S, spin, thr, gran, il, st, sc, s); break;
        default: return false;
        }
    }
    return true;
}

void launch_prep0(const DevPlan &P, const double *alm, const double *fl, double *prep, hipStream_t st, int nb)
{
    dim3 grid(4, P.mmax + 1, nb);
    hipLaunchKernelGGL(k_prep0, grid, dim3(256), 0, st, P, reinterpret_cast<const double2 *>(alm), fl,
                       reinterpret_cast<double4 *>(prep));
}

// launch_prep0 + the coefficient pass c = pm x of launch_template_project(2 nalm, nmodes, alm, null, pm, ., parts, ., nb, ., phase 1) in one launch;
// false (nothing launched) where that pass would not run in 256-thread workgroups (fine grids): the caller launches the two separately
bool launch_prep0_lowrank(const DevPlan &P, const double *alm, const double *fl, double *prep, hipStream_t st, int nb, int nmodes, const double *pm,
                          double *parts)
{
    int nt = 0, nparts = 0;
    tproj_coeffs_shape(2 * P.nalm, &nt, &nparts);
    if (nt != 256 || nmodes < 1 || nmodes > kProjMaxModes) return false;
    PrepLowRank L = {nmodes, nparts, nb, pm, parts};
    dim3 grid(4, P.mmax + 1 + (nparts + 3) / 4, nb);
    if (nb > 1 && nmodes <= kFuseModesB)
        hipLaunchKernelGGL(k_prep0_lr<true>, grid, dim3(256), 0, st, P, reinterpret_cast<const double2 *>(alm), fl, reinterpret_cast<double4 *>(prep), L);
    else
        hipLaunchKernelGGL(k_prep0_lr<false>, grid, dim3(256), 0, st, P, reinterpret_cast<const double2 *>(alm), fl, reinterpret_cast<double4 *>(prep), L);
    return true;
}

// gradient and curl coefficients as two arrays (almC null: gradient only)
void launch_preps_gc(const DevPlan &P, const DevSpinTab &S, int spin, const double *almG, const double *almC, const double *fl, double *prep,
                     hipStream_t st, int nb)
{
    dim3 grid(4, P.mmax + 1, nb);
    hipLaunchKernelGGL(k_preps, grid, dim3(256), 0, st, P, S, spin, reinterpret_cast<const double2 *>(almG),
                       reinterpret_cast<const double2 *>(almC), fl, reinterpret_cast<double4 *>(prep));
}

void launch_preps(const DevPlan &P, const DevSpinTab &S, int spin, const double *alm, const double *fl, double *prep, hipStream_t st, bool gonly)
{
    launch_preps_gc(P, S, spin, alm, gonly ? nullptr : alm + 2 * P.nalm, fl, prep, st, 1);
}

template <int R>
static void launch_synth0_r(const DevPlan &P, const double *prep, double *phase, hipStream_t st, int nb, bool pair)
{
    constexpr int RG = 64 * R;
    const int ngroups = (P.npairs + RG - 1) / RG, nmg = own_mgroups(P);
    if (nmg == 0) return;
    if (pair) hipLaunchKernelGGL((k_leg_synth0<R, true>), dim3(ngroups * nmg, nb / 2), dim3(256), 0, st, P, reinterpret_cast<const double4 *>(prep), phase);
    else hipLaunchKernelGGL((k_leg_synth0<R, false>), dim3(ngroups * nmg, nb), dim3(256), 0, st, P, reinterpret_cast<const double4 *>(prep), phase);
}

// An even block of inputs on fine grids goes through the recursion two at a time (k_leg_synth0<R, true>: bit-identical phase values, 10 instead of
// 12 FMAs per two-l step for two maps).  PLSHTS_DEBUG=1 PLSHTS_S0_PAIR_NSIDE: smallest nside (default 1024; a large value switches it off).
bool synth0_pairs(const DevPlan &P, int nb)
{
    static const int nside_min = env_int("PLSHTS_S0_PAIR_NSIDE", 1024);
    const int r = r0_synth(P);
    return nb >= 2 && nb % 2 == 0 && P.nside >= nside_min && (r == 3 || r == 4);
}

void launch_synth0(const DevPlan &P, const double *prep, double *phase, hipStream_t st, int nb)
{
    const bool pair = synth0_pairs(P, nb);
    switch (r0_synth(P)) {
    case 1: launch_synth0_r<1>(P, prep, phase, st, nb, false); break;
    case 2: launch_synth0_r<2>(P, prep, phase, st, nb, false); break;
    case 5: launch_synth0_r<5>(P, prep, phase, st, nb, false); break;
    case 6: launch_synth0_r<6>(P, prep, phase, st, nb, false); break;
    case 4: launch_synth0_r<4>(P, prep, phase, st, nb, pair); break;
    default: launch_synth0_r<3>(P, prep, phase, st, nb, pair); break;
    }
}

template <int R, bool GONLY>
static void launch_synths_r(const DevPlan &P, const DevSpinTab &S, int spin, const double *prep, double *phase, hipStream_t st, int nb)
{
    constexpr int RG = 64 * R;
    const int ngroups = (P.npairs + RG - 1) / RG, nmg = own_mgroups(P);
    if (nmg == 0) return;
    hipLaunchKernelGGL((k_leg_synths<R, GONLY, 0>), dim3(ngroups * nmg, nb), dim3(256), 0, st, P, S, spin,
                       reinterpret_cast<const double4 *>(prep), phase);
}

// general input (prep) + gradient-only input (prep2) on one recursion; phase entries of 16 doubles (four components)
void launch_synths_pair(const DevPlan &P, const DevSpinTab &S, int spin, const double *prep, const double *prep2, double *phase, hipStream_t st)
{
    const int ngroups1 = (P.npairs + 63) / 64, ngroups2 = (P.npairs + 127) / 128, nmg = own_mgroups(P);
    if (rs_synth(P) == 1)
        hipLaunchKernelGGL((k_leg_synths<1, false, 1>), dim3(ngroups1 * nmg), dim3(256), 0, st, P, S, spin,
                           reinterpret_cast<const double4 *>(prep), phase, reinterpret_cast<const double4 *>(prep2));
    else
        hipLaunchKernelGGL((k_leg_synths<2, false, 1>), dim3(ngroups2 * nmg), dim3(256), 0, st, P, S, spin,
                           reinterpret_cast<const double4 *>(prep), phase, reinterpret_cast<const double4 *>(prep2));
}

// two gradient-only inputs (the gradient legs of the temperature estimator of two simulations) on one recursion: 12 FMAs per step for the two
// transforms instead of 2 x 8; phase entries of 16 doubles as in launch_synths_pair.  Sums formed as by the gradient-only kernel: bit-identical maps.
void launch_synths_gpair(const DevPlan &P, const DevSpinTab &S, int spin, const double *prep, const double *prep2, double *phase, hipStream_t st)
{
    const int ngroups1 = (P.npairs + 63) / 64, ngroups2 = (P.npairs + 127) / 128, nmg = own_mgroups(P);
    if (nmg == 0) return;
    if (rs_synth(P) == 1)
        hipLaunchKernelGGL((k_leg_synths<1, true, 1>), dim3(ngroups1 * nmg), dim3(256), 0, st, P, S, spin,
                           reinterpret_cast<const double4 *>(prep), phase, reinterpret_cast<const double4 *>(prep2));
    else
        hipLaunchKernelGGL((k_leg_synths<2, true, 1>), dim3(ngroups2 * nmg), dim3(256), 0, st, P, S, spin,
                           reinterpret_cast<const double4 *>(prep), phase, reinterpret_cast<const double4 *>(prep2));
}

// two general inputs (two simulations) on one recursion; phase entries of 16 doubles: components (Q1, U1, Q2, U2).
// PLSHTS_RSB overrides the rings per lane (1 or 2)
// npairs_b > 1: that many pairs of inputs in one launch (blockIdx.y), prep / prep2 of pair y at 2 y prep arrays from the given
// pointers -- the entries (2 y, 2 y + 1) of a block of prep arrays when prep2 = prep + one array; phase components 4 y ... 4 y + 3
void launch_synths_batch2(const DevPlan &P, const DevSpinTab &S, int spin, const double *prep, const double *prep2, double *phase, hipStream_t st,
                          int npairs_b)
{
    const int ngroups1 = (P.npairs + 63) / 64, ngroups2 = (P.npairs + 127) / 128, nmg = own_mgroups(P);
    if (nmg == 0) return;
    int r = env_int("PLSHTS_RSB", 0);
    if (r != 1 && r != 2) r = rs_synth(P) == 1 ? 1 : 2;
    const int bstride = npairs_b > 1 ? 2 : 1;
    if (r == 1)
        hipLaunchKernelGGL((k_leg_synths<1, false, 2>), dim3(ngroups1 * nmg, npairs_b), dim3(256), 0, st, P, S, spin,
                           reinterpret_cast<const double4 *>(prep), phase, reinterpret_cast<const double4 *>(prep2), bstride);
    else
        hipLaunchKernelGGL((k_leg_synths<2, false, 2>), dim3(ngroups2 * nmg, npairs_b), dim3(256), 0, st, P, S, spin,
                           reinterpret_cast<const double4 *>(prep), phase, reinterpret_cast<const double4 *>(prep2), bstride);
}

void launch_synths(const DevPlan &P, const DevSpinTab &S, int spin, const double *prep, double *phase, hipStream_t st, bool gonly, int nb)
{
    const int r = rs_synth(P);
    if (gonly) {
        switch (r) {
        case 1: launch_synths_r<1, true>(P, S, spin, prep, phase, st, nb); break;
        case 3: launch_synths_r<3, true>(P, S, spin, prep, phase, st, nb); break;
        case 4: launch_synths_r<4, true>(P, S, spin, prep, phase, st, nb); break;
        default: launch_synths_r<2, true>(P, S, spin, prep, phase, st, nb); break;
        }
        return;
    }
    switch (r) {
    case 1: launch_synths_r<1, false>(P, S, spin, prep, phase, st, nb); break;
    case 3: launch_synths_r<3, false>(P, S, spin, prep, phase, st, nb); break;
    case 4: launch_synths_r<4, false>(P, S, spin, prep, phase, st, nb); break;
    default: launch_synths_r<2, false>(P, S, spin, prep, phase, st, nb); break;
    }
}

template <int R>
static void launch_anal0_r(const DevPlan &P, const double *phase, double *partial, const double *fl, double *alm, hipStream_t st,
                           const double *add, const double *fl_add, int nb, const PostLowRank &lr, const PostDots &dots)
{
    constexpr int RG = 64 * R;
    const int ngroups = (P.npairs + RG - 1) / RG, nmg = own_mgroups(P);
    if (nmg > 0) hipLaunchKernelGGL(k_leg_anal0<R>, dim3(ngroups * nmg, nb), dim3(256), 0, st, P, phase, partial);
    dim3 grid(4, P.mmax + 1, nb);
    hipLaunchKernelGGL(k_post0, grid, dim3(256), 0, st, P, RG, reinterpret_cast<const double4 *>(partial), fl,
                       reinterpret_cast<double2 *>(alm), reinterpret_cast<const double2 *>(add), fl_add, lr, dots);
}

void launch_anal0(const DevPlan &P, const double *phase, double *partial, const double *fl, double *alm, hipStream_t st, const double *add,
                  const double *fl_add, int nb, int lr_nmodes, int lr_nparts, int lr_pstride, const double *lr_rm, const double *lr_parts,
                  const PostDots *dots_, int64_t lr_bstride)
{
    const PostDots dots = dots_ ? *dots_ : PostDots();
    PostLowRank lr;
    if (lr_nmodes > 0) { lr.nmodes = lr_nmodes; lr.nparts = lr_nparts; lr.pstride = lr_pstride; lr.bstride = lr_bstride; lr.rm = lr_rm; lr.parts = lr_parts; }
    switch (r0_anal(P)) {
    case 1: launch_anal0_r<1>(P, phase, partial, fl, alm, st, add, fl_add, nb, lr, dots); break;
    case 2: launch_anal0_r<2>(P, phase, partial, fl, alm, st, add, fl_add, nb, lr, dots); break;
    case 3: launch_anal0_r<3>(P, phase, partial, fl, alm, st, add, fl_add, nb, lr, dots); break;
    case 5: launch_anal0_r<5>(P, phase, partial, fl, alm, st, add, fl_add, nb, lr, dots); break;
    case 6: launch_anal0_r<6>(P, phase, partial, fl, alm, st, add, fl_add, nb, lr, dots); break;
    case 7: launch_anal0_r<7>(P, phase, partial, fl, alm, st, add, fl_add, nb, lr, dots); break;
    case 8: launch_anal0_r<8>(P, phase, partial, fl, alm, st, add, fl_add, nb, lr, dots); break;
    default: launch_anal0_r<4>(P, phase, partial, fl, alm, st, add, fl_add, nb, lr, dots); break;
    }
}

template <int R>
static void launch_anals_r(const DevPlan &P, const DevSpinTab &S, int spin, int64_t nent, const double *phase, double *partial,
                           const double *fl, double *almG, double *almC, hipStream_t st, const double *addG, const double *addC,
                           const double *flG, const double *flC, int nb, const PostDots &dots)
{
    constexpr int RG = 64 * R;
    const int ngroups = (P.npairs + RG - 1) / RG, nmg = own_mgroups(P);
    if (nmg > 0) hipLaunchKernelGGL(k_leg_anals<R>, dim3(ngroups * nmg, nb), dim3(256), 0, st, P, S, spin, phase, partial, nent);
    dim3 grid(4, P.mmax + 1, nb);
    hipLaunchKernelGGL(k_posts, grid, dim3(256), 0, st, P, S, spin, RG, nent, reinterpret_cast<const double4 *>(partial), fl,
                       reinterpret_cast<double2 *>(almG), reinterpret_cast<double2 *>(almC), reinterpret_cast<const double2 *>(addG),
                       reinterpret_cast<const double2 *>(addC), flG, flC, dots);
}

void launch_anals_gc(const DevPlan &P, const DevSpinTab &S, int spin, int64_t nent, const double *phase, double *partial, const double *fl,
                     double *almG, double *almC, hipStream_t st, const double *addG, const double *addC, const double *flG, const double *flC,
                     int nb = 1, const PostDots *dots = nullptr);

void launch_anals(const DevPlan &P, const DevSpinTab &S, int spin, int64_t nent, const double *phase, double *partial,
                  const double *fl, double *alm, hipStream_t st)
{
    launch_anals_gc(P, S, spin, nent, phase, partial, fl, alm, alm + 2 * P.nalm, st, nullptr, nullptr, nullptr, nullptr);
}

// gradient / curl outputs as two arrays, with the optional add terms of k_posts
void launch_anals_gc(const DevPlan &P, const DevSpinTab &S, int spin, int64_t nent, const double *phase, double *partial, const double *fl,
                     double *almG, double *almC, hipStream_t st, const double *addG, const double *addC, const double *flG, const double *flC,
                     int nb, const PostDots *dots_)
{
    const PostDots dots = dots_ ? *dots_ : PostDots();
    switch (rs_anal(P)) {
    case 1: launch_anals_r<1>(P, S, spin, nent, phase, partial, fl, almG, almC, st, addG, addC, flG, flC, nb, dots); break;
    case 2: launch_anals_r<2>(P, S, spin, nent, phase, partial, fl, almG, almC, st, addG, addC, flG, flC, nb, dots); break;
    case 4: launch_anals_r<4>(P, S, spin, nent, phase, partial, fl, almG, almC, st, addG, addC, flG, flC, nb, dots); break;
    default: launch_anals_r<3>(P, S, spin, nent, phase, partial, fl, almG, almC, st, addG, addC, flG, flC, nb, dots); break;
    }
}

}  // namespace plshts
