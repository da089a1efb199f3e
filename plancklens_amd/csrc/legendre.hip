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
#include <hip/hip_runtime.h>

#include "device_plan.h"
#include "legendre_math.h"

namespace plshts {

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ bool wave_all(bool p) { return __all(p) != 0; }

// -----------------------------------------------------------------------------------------------------
// alm -> recursion-basis coefficients (fused hp.almxfl)
// -----------------------------------------------------------------------------------------------------
// spin 0: prep[e] = {c_re, c_im, d_re, d_im}
__global__ void k_prep0(DevPlan P, const double2 *__restrict__ alm, const double *__restrict__ fl, double4 *__restrict__ prep)
{
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

// spin s: prep[e] = {An_re, An_im, Ap_re, Ap_im}
__global__ void k_preps(DevPlan P, DevSpinTab S, int spin, const double2 *__restrict__ almG, const double2 *__restrict__ almC,
                        const double *__restrict__ fl, double4 *__restrict__ prep)
{
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
        double2 g = almG[abase + l], c = almC[abase + l];
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
// synthesis, spin 0
// -----------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_leg_synth0(DevPlan P, const double4 *__restrict__ prep, double *__restrict__ phase)
{
    constexpr int RG = 64 * R;
    __shared__ double tile[RG * 16];  // [ring][m_local 4][4]
    const int wave = wave_id();
    const int lane = threadIdx.x & 63;
    const int ngroups = (P.npairs + RG - 1) / RG;
    const int mg = blockIdx.x / ngroups, g = blockIdx.x % ngroups;
    const int last = min(P.npairs - 1, g * RG + RG - 1);
    if (P.mlim0[last] < 4 * mg) return;  // every ring of the group is pruned for every m of the group
    const int m = 4 * mg + wave;

    double cr[R], ci[R], dr[R], di[R];
#pragma unroll
    for (int k = 0; k < R; ++k) cr[k] = ci[k] = dr[k] = di[k] = 0.0;

    if (m <= P.mmax) {
        Rec0 r[R];
        const double seed = P.seed0[m];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int ip = g * RG + k * 64 + lane;
            const bool ok = ip < P.npairs && m <= P.mlim0[min(ip, P.npairs - 1)];
            const int ipc = min(ip, P.npairs - 1);
            rec0_init(r[k], seed, m, P.cth[ipc], P.sth[ipc], ok);
        }
        const int nil = (P.lmax - m) / 2 + 1;
        const int64_t base = P.off0[m];
        const double2 *__restrict__ ab = reinterpret_cast<const double2 *>(P.ab0) + base;
        const double4 *__restrict__ cd = prep + base;
        int il = 0;
        // scaled phase: some lane has not yet reached the IEEE range
        for (; il < nil; ++il) {
            bool done = true;
#pragma unroll
            for (int k = 0; k < R; ++k) done = done && (r[k].sc == 0 || r[k].sc == kNeverActive);
            if (wave_all(done)) break;
            const double2 c_ab = ab[il];
            const double4 c = cd[il];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double v = rec0_value(r[k]);
                cr[k] = fma(v, c.x, cr[k]); ci[k] = fma(v, c.y, ci[k]);
                dr[k] = fma(v, c.z, dr[k]); di[k] = fma(v, c.w, di[k]);
                rec0_step_careful(r[k], c_ab.x, c_ab.y);
            }
        }
        // IEEE phase: pure FMA stream
#pragma unroll 2
        for (; il < nil; ++il) {
            const double2 c_ab = ab[il];
            const double4 c = cd[il];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double v = r[k].p1;
                cr[k] = fma(v, c.x, cr[k]); ci[k] = fma(v, c.y, ci[k]);
                dr[k] = fma(v, c.z, dr[k]); di[k] = fma(v, c.w, di[k]);
                rec0_step_fast(r[k], c_ab.x, c_ab.y);
            }
        }
    }
    // F_north = C + x D, F_south = C - x D  -> LDS tile -> ring-major global phase array
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int rl = k * 64 + lane;
        const int ip = min(g * RG + rl, P.npairs - 1);
        const double x = P.cth[ip];
        double *t = tile + rl * 16 + wave * 4;
        t[0] = fma(x, dr[k], cr[k]); t[1] = fma(x, di[k], ci[k]);
        t[2] = fma(-x, dr[k], cr[k]); t[3] = fma(-x, di[k], ci[k]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < RG * 8; c += 256) {
        const int rl = c >> 3, part = c & 7;
        const int ip = g * RG + rl;
        if (ip < P.npairs) {
            double2 v = *reinterpret_cast<const double2 *>(tile + rl * 16 + part * 2);
            *reinterpret_cast<double2 *>(phase + ((int64_t)ip * P.mstride + 4 * mg) * 4 + part * 2) = v;
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// synthesis, spin s: phase entry = {Q_N re, im, Q_S re, im, U_N re, im, U_S re, im}
// -----------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_leg_synths(DevPlan P, DevSpinTab S, int spin, const double4 *__restrict__ prep,
                                                    double *__restrict__ phase)
{
    constexpr int RG = 64 * R;
    __shared__ double tile[RG * 32];  // [ring][m_local 4][8]
    const int wave = wave_id();
    const int lane = threadIdx.x & 63;
    const int ngroups = (P.npairs + RG - 1) / RG;
    const int mg = blockIdx.x / ngroups, g = blockIdx.x % ngroups;
    const int last = min(P.npairs - 1, g * RG + RG - 1);
    if (S.mlim[last] < 4 * mg) return;
    const int m = 4 * mg + wave;
    const int l0 = m > spin ? m : spin;

    // X_N = sum Sn An, Y_N = sum Sp Ap, X_S = sum sigma Sp An, Y_S = sum sigma Sn Ap
    double xn_r[R], xn_i[R], yn_r[R], yn_i[R], xs_r[R], xs_i[R], ys_r[R], ys_i[R];
#pragma unroll
    for (int k = 0; k < R; ++k) xn_r[k] = xn_i[k] = yn_r[k] = yn_i[k] = xs_r[k] = xs_i[k] = ys_r[k] = ys_i[k] = 0.0;

    if (m <= P.mmax && l0 <= P.lmax) {
        RecS r[R];
        const double fn = S.seedfac_n[m], fp = S.seedfac_p[m];
        const int psin = S.psin[m], phalf = S.phalf[m], ucn = S.usecos_n[m], ucp = S.usecos_p[m];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int ip = g * RG + k * 64 + lane;
            const int ipc = min(ip, P.npairs - 1);
            const bool ok = ip < P.npairs && m <= S.mlim[ipc];
            recs_init(r[k], fn, fp, psin, phalf, ucn, ucp, P.cth[ipc], P.sth[ipc], P.chalf[ipc], P.shalf[ipc], ok);
        }
        const int nl = P.lmax - l0 + 1;
        const int64_t base = S.off[m];
        const double2 *__restrict__ ab = reinterpret_cast<const double2 *>(S.ab) + base;
        const double4 *__restrict__ aa = prep + base;
        double sig = ((l0 + m) & 1) ? -1.0 : 1.0;  // sigma_l = (-1)^(l + m)
        int i = 0;
        for (; i < nl; ++i) {
            bool done = true;
#pragma unroll
            for (int k = 0; k < R; ++k)
                done = done && (r[k].scn == 0 || r[k].scn == kNeverActive) && (r[k].scp == 0 || r[k].scp == kNeverActive);
            if (wave_all(done)) break;
            const double2 c_ab = ab[i];
            const double4 a = aa[i];
            const double sar = sig * a.x, sai = sig * a.y, spr = sig * a.z, spi = sig * a.w;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double vn = recs_value_n(r[k]), vp = recs_value_p(r[k]);
                xn_r[k] = fma(vn, a.x, xn_r[k]); xn_i[k] = fma(vn, a.y, xn_i[k]);
                yn_r[k] = fma(vp, a.z, yn_r[k]); yn_i[k] = fma(vp, a.w, yn_i[k]);
                xs_r[k] = fma(vp, sar, xs_r[k]); xs_i[k] = fma(vp, sai, xs_i[k]);
                ys_r[k] = fma(vn, spr, ys_r[k]); ys_i[k] = fma(vn, spi, ys_i[k]);
                recs_step_careful(r[k], c_ab.x, c_ab.y);
            }
            sig = -sig;
        }
        // IEEE phase, two l per trip so that sigma is a compile-time sign
        for (; i + 1 < nl; i += 2) {
            const double2 c_ab0 = ab[i], c_ab1 = ab[i + 1];
            const double4 a0 = aa[i], a1 = aa[i + 1];
            const double s0 = sig;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                double vn = r[k].n1, vp = r[k].p1;
                const double wn = s0 * vn, wp = s0 * vp;
                xn_r[k] = fma(vn, a0.x, xn_r[k]); xn_i[k] = fma(vn, a0.y, xn_i[k]);
                yn_r[k] = fma(vp, a0.z, yn_r[k]); yn_i[k] = fma(vp, a0.w, yn_i[k]);
                xs_r[k] = fma(wp, a0.x, xs_r[k]); xs_i[k] = fma(wp, a0.y, xs_i[k]);
                ys_r[k] = fma(wn, a0.z, ys_r[k]); ys_i[k] = fma(wn, a0.w, ys_i[k]);
                recs_step_fast(r[k], c_ab0.x, c_ab0.y);
                vn = r[k].n1; vp = r[k].p1;
                const double un = -s0 * vn, up = -s0 * vp;
                xn_r[k] = fma(vn, a1.x, xn_r[k]); xn_i[k] = fma(vn, a1.y, xn_i[k]);
                yn_r[k] = fma(vp, a1.z, yn_r[k]); yn_i[k] = fma(vp, a1.w, yn_i[k]);
                xs_r[k] = fma(up, a1.x, xs_r[k]); xs_i[k] = fma(up, a1.y, xs_i[k]);
                ys_r[k] = fma(un, a1.z, ys_r[k]); ys_i[k] = fma(un, a1.w, ys_i[k]);
                recs_step_fast(r[k], c_ab1.x, c_ab1.y);
            }
        }
        if (i < nl) {
            const double4 a = aa[i];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double vn = r[k].n1, vp = r[k].p1;
                const double wn = sig * vn, wp = sig * vp;
                xn_r[k] = fma(vn, a.x, xn_r[k]); xn_i[k] = fma(vn, a.y, xn_i[k]);
                yn_r[k] = fma(vp, a.z, yn_r[k]); yn_i[k] = fma(vp, a.w, yn_i[k]);
                xs_r[k] = fma(wp, a.x, xs_r[k]); xs_i[k] = fma(wp, a.y, xs_i[k]);
                ys_r[k] = fma(wn, a.z, ys_r[k]); ys_i[k] = fma(wn, a.w, ys_i[k]);
            }
        }
    }
    // Q = X + Y, U = i (Y - X)
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int rl = k * 64 + lane;
        double *t = tile + rl * 32 + wave * 8;
        t[0] = xn_r[k] + yn_r[k]; t[1] = xn_i[k] + yn_i[k];
        t[2] = xs_r[k] + ys_r[k]; t[3] = xs_i[k] + ys_i[k];
        t[4] = -(yn_i[k] - xn_i[k]); t[5] = yn_r[k] - xn_r[k];
        t[6] = -(ys_i[k] - xs_i[k]); t[7] = ys_r[k] - xs_r[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < RG * 16; c += 256) {
        const int rl = c >> 4, part = c & 15;
        const int ip = g * RG + rl;
        if (ip < P.npairs) {
            double2 v = *reinterpret_cast<const double2 *>(tile + rl * 32 + part * 2);
            *reinterpret_cast<double2 *>(phase + ((int64_t)ip * P.mstride + 4 * mg) * 8 + part * 2) = v;
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// cross-lane transpose-reduce: every lane holds v[0..63]; on return lane L holds sum over lanes of v[L]
// -----------------------------------------------------------------------------------------------------
template <int HALF>
__device__ __forceinline__ void reduce_step(double *v, int lane)
{
    const bool up = (lane & HALF) != 0;  // the lane-id bit handled by this step equals the half size
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const double keep = up ? v[i + HALF] : v[i];
        const double send = up ? v[i] : v[i + HALF];
        const double recv = __shfl_xor(send, HALF, 64);
        v[i] = keep + recv;
    }
}

__device__ __forceinline__ double reduce64_transpose(double *v, int lane)
{
    reduce_step<32>(v, lane);
    reduce_step<16>(v, lane);
    reduce_step<8>(v, lane);
    reduce_step<4>(v, lane);
    reduce_step<2>(v, lane);
    reduce_step<1>(v, lane);
    return v[0];
}

// -----------------------------------------------------------------------------------------------------
// analysis, spin 0: partial[g][entry] = {C_re, C_im, D_re, D_im} summed over the ring pairs of group g
// -----------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_leg_anal0(DevPlan P, const double *__restrict__ phase, double *__restrict__ partial)
{
    constexpr int RG = 64 * R;
    constexpr int T = 16;
    __shared__ double tile[RG * 16];
    const int wave = wave_id();
    const int lane = threadIdx.x & 63;
    const int ngroups = (P.npairs + RG - 1) / RG;
    const int mg = blockIdx.x / ngroups, g = blockIdx.x % ngroups;
    const int last = min(P.npairs - 1, g * RG + RG - 1);
    if (P.mlim0[last] < 4 * mg) return;
    const int m = 4 * mg + wave;

    for (int c = threadIdx.x; c < RG * 8; c += 256) {
        const int rl = c >> 3, part = c & 7;
        const int ip = g * RG + rl;
        double2 v = make_double2(0., 0.);
        if (ip < P.npairs) v = *reinterpret_cast<const double2 *>(phase + ((int64_t)ip * P.mstride + 4 * mg) * 4 + part * 2);
        *reinterpret_cast<double2 *>(tile + rl * 16 + part * 2) = v;
    }
    __syncthreads();
    if (m > P.mmax) return;

    Rec0 r[R];
    double er[R], ei[R], orr[R], oi[R];
    const double seed = P.seed0[m];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int rl = k * 64 + lane;
        const int ip = g * RG + rl;
        const int ipc = min(ip, P.npairs - 1);
        const bool ok = ip < P.npairs && m <= P.mlim0[ipc];
        const double x = P.cth[ipc];
        rec0_init(r[k], seed, m, x, P.sth[ipc], ok);
        const double *t = tile + rl * 16 + wave * 4;
        const double nr = ok ? t[0] : 0., ni = ok ? t[1] : 0., sr = ok ? t[2] : 0., si = ok ? t[3] : 0.;
        er[k] = nr + sr; ei[k] = ni + si;
        orr[k] = (nr - sr) * x; oi[k] = (ni - si) * x;
    }
    const int nil = (P.lmax - m) / 2 + 1;
    const int64_t base = P.off0[m];
    const double2 *__restrict__ ab = reinterpret_cast<const double2 *>(P.ab0) + base;
    double *__restrict__ out = partial + ((int64_t)g * P.nent0 + base) * 4;
    bool all_active = false;
    for (int il0 = 0; il0 < nil; il0 += T) {
        double acc[64];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int il = il0 + t;
            double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
            if (il < nil) {
                const double2 c_ab = ab[il];
                if (!all_active) {
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const double v = rec0_value(r[k]);
                        a0 = fma(v, er[k], a0); a1 = fma(v, ei[k], a1); a2 = fma(v, orr[k], a2); a3 = fma(v, oi[k], a3);
                        rec0_step_careful(r[k], c_ab.x, c_ab.y);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const double v = r[k].p1;
                        a0 = fma(v, er[k], a0); a1 = fma(v, ei[k], a1); a2 = fma(v, orr[k], a2); a3 = fma(v, oi[k], a3);
                        rec0_step_fast(r[k], c_ab.x, c_ab.y);
                    }
                }
            }
            acc[4 * t] = a0; acc[4 * t + 1] = a1; acc[4 * t + 2] = a2; acc[4 * t + 3] = a3;
        }
        if (!all_active) {
            bool done = true;
#pragma unroll
            for (int k = 0; k < R; ++k) done = done && (r[k].sc == 0 || r[k].sc == kNeverActive);
            all_active = wave_all(done);
        }
        const double tot = reduce64_transpose(acc, lane);
        if (il0 + (lane >> 2) < nil) out[(int64_t)il0 * 4 + lane] = tot;
    }
}

// reduce partials over ring groups and convert (C, D) -> a_lm (fused hp.almxfl)
__global__ void k_post0(DevPlan P, int RG, const double4 *__restrict__ partial, const double *__restrict__ fl, double2 *__restrict__ alm)
{
    const int m = blockIdx.y;
    const int nil = (P.lmax - m) / 2 + 1;
    const int64_t base = P.off0[m];
    const int64_t abase = (int64_t)m * (2 * P.lmax + 1 - m) / 2;
    const int ngroups = (P.npairs + RG - 1) / RG;
    const int mg4 = 4 * (m / 4);
    for (int il = blockIdx.x * blockDim.x + threadIdx.x; il < nil; il += gridDim.x * blockDim.x) {
        const int64_t e = base + il;
        double c0r = 0., c0i = 0., c1r = 0., c1i = 0., dr = 0., di = 0.;
        for (int g = 0; g < ngroups; ++g) {
            const int last = min(P.npairs - 1, g * RG + RG - 1);
            if (P.mlim0[last] < mg4) continue;
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
        alm[abase + l] = a;
        if (l + 1 <= P.lmax) {
            double2 b;
            b.x = al * dr; b.y = al * di;
            if (fl) { b.x *= fl[l + 1]; b.y *= fl[l + 1]; }
            alm[abase + l + 1] = b;
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// analysis, spin s: partial[g][entry] = {G'_re, G'_im, C'_re, C'_im}
// -----------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_leg_anals(DevPlan P, DevSpinTab S, int spin, const double *__restrict__ phase,
                                                   double *__restrict__ partial, int64_t nent)
{
    constexpr int RG = 64 * R;
    constexpr int T = 16;
    __shared__ double tile[RG * 32];
    const int wave = wave_id();
    const int lane = threadIdx.x & 63;
    const int ngroups = (P.npairs + RG - 1) / RG;
    const int mg = blockIdx.x / ngroups, g = blockIdx.x % ngroups;
    const int last = min(P.npairs - 1, g * RG + RG - 1);
    if (S.mlim[last] < 4 * mg) return;
    const int m = 4 * mg + wave;
    const int l0 = m > spin ? m : spin;

    for (int c = threadIdx.x; c < RG * 16; c += 256) {
        const int rl = c >> 4, part = c & 15;
        const int ip = g * RG + rl;
        double2 v = make_double2(0., 0.);
        if (ip < P.npairs) v = *reinterpret_cast<const double2 *>(phase + ((int64_t)ip * P.mstride + 4 * mg) * 8 + part * 2);
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
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int rl = k * 64 + lane;
        const int ip = g * RG + rl;
        const int ipc = min(ip, P.npairs - 1);
        const bool ok = ip < P.npairs && m <= S.mlim[ipc];
        recs_init(r[k], fn, fp, psin, phalf, ucn, ucp, P.cth[ipc], P.sth[ipc], P.chalf[ipc], P.shalf[ipc], ok);
        const double *t = tile + rl * 32 + wave * 8;
        const double z = ok ? 1.0 : 0.0;
        const double qnr = z * t[0], qni = z * t[1], qsr = z * t[2], qsi = z * t[3];
        const double unr = z * t[4], uni = z * t[5], usr = z * t[6], usi = z * t[7];
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
    const int nl = P.lmax - l0 + 1;
    const int64_t base = S.off[m];
    const double2 *__restrict__ ab = reinterpret_cast<const double2 *>(S.ab) + base;
    double *__restrict__ out = partial + ((int64_t)g * nent + base) * 4;
    bool all_active = false;
    for (int i0 = 0; i0 < nl; i0 += T) {
        double acc[64];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int i = i0 + t;
            double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
            if (i < nl) {
                const double2 c_ab = ab[i];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    double vn, vp;
                    if (!all_active) { vn = recs_value_n(r[k]); vp = recs_value_p(r[k]); }
                    else { vn = r[k].n1; vp = r[k].p1; }
                    if ((t & 1) == 0) {
                        a0 = fma(vn, aer[k], a0); a1 = fma(vn, aei[k], a1); a2 = fma(vn, aor[k], a2); a3 = fma(vn, aoi[k], a3);
                        a0 = fma(vp, ber[k], a0); a1 = fma(vp, bei[k], a1); a2 = fma(-vp, bor[k], a2); a3 = fma(-vp, boi[k], a3);
                    } else {
                        a0 = fma(vn, aor[k], a0); a1 = fma(vn, aoi[k], a1); a2 = fma(vn, aer[k], a2); a3 = fma(vn, aei[k], a3);
                        a0 = fma(vp, bor[k], a0); a1 = fma(vp, boi[k], a1); a2 = fma(-vp, ber[k], a2); a3 = fma(-vp, bei[k], a3);
                    }
                    if (!all_active) recs_step_careful(r[k], c_ab.x, c_ab.y);
                    else recs_step_fast(r[k], c_ab.x, c_ab.y);
                }
            }
            acc[4 * t] = a0; acc[4 * t + 1] = a1; acc[4 * t + 2] = a2; acc[4 * t + 3] = a3;
        }
        if (!all_active) {
            bool done = true;
#pragma unroll
            for (int k = 0; k < R; ++k)
                done = done && (r[k].scn == 0 || r[k].scn == kNeverActive) && (r[k].scp == 0 || r[k].scp == kNeverActive);
            all_active = wave_all(done);
        }
        const double tot = reduce64_transpose(acc, lane);
        if (i0 + (lane >> 2) < nl) out[(int64_t)i0 * 4 + lane] = tot;
    }
}

// G_l = -1/2 beta_l G'_l,  C_l = i/2 beta_l C'_l
__global__ void k_posts(DevPlan P, DevSpinTab S, int spin, int RG, int64_t nent, const double4 *__restrict__ partial,
                        const double *__restrict__ fl, double2 *__restrict__ almG, double2 *__restrict__ almC)
{
    const int m = blockIdx.y;
    const int l0 = m > spin ? m : spin;
    const int64_t abase = (int64_t)m * (2 * P.lmax + 1 - m) / 2;
    // entries below the spin are zero
    for (int l = m + blockIdx.x * blockDim.x + threadIdx.x; l < l0 && l <= P.lmax; l += gridDim.x * blockDim.x) {
        almG[abase + l] = make_double2(0., 0.);
        almC[abase + l] = make_double2(0., 0.);
    }
    const int nl = P.lmax - l0 + 1;
    if (nl <= 0) return;
    const int64_t base = S.off[m];
    const int ngroups = (P.npairs + RG - 1) / RG;
    const int mg4 = 4 * (m / 4);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += gridDim.x * blockDim.x) {
        const int64_t e = base + i;
        double gr = 0., gi = 0., cr = 0., ci = 0.;
        for (int g = 0; g < ngroups; ++g) {
            const int last = min(P.npairs - 1, g * RG + RG - 1);
            if (S.mlim[last] < mg4) continue;
            const double4 v = partial[(int64_t)g * nent + e];
            gr += v.x; gi += v.y; cr += v.z; ci += v.w;
        }
        const int l = l0 + i;
        double f = 0.5 * S.beta[e];
        if (fl) f *= fl[l];
        almG[abase + l] = make_double2(-f * gr, -f * gi);   // G = -1/2 beta G'
        almC[abase + l] = make_double2(-f * ci, f * cr);    // C = i/2 beta C'
    }
}

// -----------------------------------------------------------------------------------------------------
// host launchers
// -----------------------------------------------------------------------------------------------------
constexpr int kR0 = 4;   // ring pairs per lane, spin 0
constexpr int kRS = 2;   // ring pairs per lane, spin s

int rings_per_group(int spin) { return 64 * (spin == 0 ? kR0 : kRS); }

void launch_prep0(const DevPlan &P, const double *alm, const double *fl, double *prep, hipStream_t st)
{
    dim3 grid(4, P.mmax + 1);
    hipLaunchKernelGGL(k_prep0, grid, dim3(256), 0, st, P, reinterpret_cast<const double2 *>(alm), fl,
                       reinterpret_cast<double4 *>(prep));
}

void launch_preps(const DevPlan &P, const DevSpinTab &S, int spin, const double *alm, const double *fl, double *prep, hipStream_t st)
{
    dim3 grid(4, P.mmax + 1);
    hipLaunchKernelGGL(k_preps, grid, dim3(256), 0, st, P, S, spin, reinterpret_cast<const double2 *>(alm),
                       reinterpret_cast<const double2 *>(alm) + P.nalm, fl, reinterpret_cast<double4 *>(prep));
}

void launch_synth0(const DevPlan &P, const double *prep, double *phase, hipStream_t st)
{
    constexpr int RG = 64 * kR0;
    const int ngroups = (P.npairs + RG - 1) / RG, nmg = (P.mmax + 4) / 4;
    hipLaunchKernelGGL(k_leg_synth0<kR0>, dim3(ngroups * nmg), dim3(256), 0, st, P, reinterpret_cast<const double4 *>(prep), phase);
}

void launch_synths(const DevPlan &P, const DevSpinTab &S, int spin, const double *prep, double *phase, hipStream_t st)
{
    constexpr int RG = 64 * kRS;
    const int ngroups = (P.npairs + RG - 1) / RG, nmg = (P.mmax + 4) / 4;
    hipLaunchKernelGGL(k_leg_synths<kRS>, dim3(ngroups * nmg), dim3(256), 0, st, P, S, spin,
                       reinterpret_cast<const double4 *>(prep), phase);
}

void launch_anal0(const DevPlan &P, const double *phase, double *partial, const double *fl, double *alm, hipStream_t st)
{
    constexpr int RG = 64 * kR0;
    const int ngroups = (P.npairs + RG - 1) / RG, nmg = (P.mmax + 4) / 4;
    hipLaunchKernelGGL(k_leg_anal0<kR0>, dim3(ngroups * nmg), dim3(256), 0, st, P, phase, partial);
    dim3 grid(4, P.mmax + 1);
    hipLaunchKernelGGL(k_post0, grid, dim3(256), 0, st, P, RG, reinterpret_cast<const double4 *>(partial), fl,
                       reinterpret_cast<double2 *>(alm));
}

void launch_anals(const DevPlan &P, const DevSpinTab &S, int spin, int64_t nent, const double *phase, double *partial,
                  const double *fl, double *alm, hipStream_t st)
{
    constexpr int RG = 64 * kRS;
    const int ngroups = (P.npairs + RG - 1) / RG, nmg = (P.mmax + 4) / 4;
    hipLaunchKernelGGL(k_leg_anals<kRS>, dim3(ngroups * nmg), dim3(256), 0, st, P, S, spin, phase, partial, nent);
    dim3 grid(4, P.mmax + 1);
    hipLaunchKernelGGL(k_posts, grid, dim3(256), 0, st, P, S, spin, RG, nent, reinterpret_cast<const double4 *>(partial), fl,
                       reinterpret_cast<double2 *>(alm), reinterpret_cast<double2 *>(alm) + P.nalm);
}

}  // namespace plshts
