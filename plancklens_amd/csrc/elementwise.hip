// HBM-bound helpers on alm / map arrays (hp.almxfl, hp.alm2cl, alm_copy, axpy, pixel products) and the
// FP64 FMA-rate microbenchmark.  All are streaming kernels: coalesced 16-byte accesses, grid-stride.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cstdint>

#include "device_plan.h"
#include "tproj_device.h"

namespace plshts {

// Batches (block vectors of the conjugate-gradient filter: several right-hand sides through every launch): an array argument of a
// batched launch holds the entries back to back; the kernels take the entry index from blockIdx.z (the (l, m)-grid kernels) or
// blockIdx.y (the others) and offset their pointers by whole arrays.  A batch of one launches exactly what it always did.
__device__ __forceinline__ int64_t alm_count(int lmax) { return (int64_t)(lmax + 1) * (lmax + 2) / 2; }

// alm index -> (l, m) without a table: thread per (m, l) on a 2-D grid
__global__ void k_almxfl(int lmax, const double2 *__restrict__ in_, const double *__restrict__ fl, int nfl, double2 *__restrict__ out_)
{
    const double2 *__restrict__ in = in_ + blockIdx.z * alm_count(lmax);
    double2 *__restrict__ out = out_ + blockIdx.z * alm_count(lmax);
    const int m = blockIdx.y;
    const int64_t base = (int64_t)m * (2 * lmax + 1 - m) / 2;
    for (int l = m + blockIdx.x * blockDim.x + threadIdx.x; l <= lmax; l += gridDim.x * blockDim.x) {
        const double f = l < nfl ? fl[l] : 0.0;
        const double2 a = in[base + l];
        out[base + l] = make_double2(a.x * f, a.y * f);
    }
}

__global__ void k_alm_copy(int lmax_in, const double2 *__restrict__ in_, int lmax_out, double2 *__restrict__ out_)
{
    const double2 *__restrict__ in = in_ + blockIdx.z * alm_count(lmax_in);
    double2 *__restrict__ out = out_ + blockIdx.z * alm_count(lmax_out);
    const int m = blockIdx.y;
    const int64_t bo = (int64_t)m * (2 * lmax_out + 1 - m) / 2;
    const int64_t bi = (int64_t)m * (2 * lmax_in + 1 - m) / 2;
    for (int l = m + blockIdx.x * blockDim.x + threadIdx.x; l <= lmax_out; l += gridDim.x * blockDim.x)
        out[bo + l] = (l <= lmax_in && m <= lmax_in) ? in[bi + l] : make_double2(0., 0.);
}

// cl[l] = 1/(2l+1) sum_m w_m Re(a b*); one workgroup per l, deterministic tree reduction
__global__ void k_alm2cl(int lmax, const double2 *__restrict__ a, const double2 *__restrict__ b, double *__restrict__ cl)
{
    __shared__ double red[256];
    const int l = blockIdx.x;
    double s = 0.0;
    for (int m = threadIdx.x; m <= l; m += blockDim.x) {
        const int64_t i = (int64_t)m * (2 * lmax + 1 - m) / 2 + l;
        const double2 x = a[i], y = b[i];
        const double p = x.x * y.x + x.y * y.y;
        s += (m == 0) ? p : 2.0 * p;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = blockDim.x >> 1; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) cl[l] = red[0] / (2.0 * l + 1.0);
}

__global__ void k_axpy(int64_t n, double a, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = fma(a, x[i], y[i]);
}

__global__ void k_map_mul(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] * b[i];
}

__global__ void k_map_cmul(int64_t n, const double *__restrict__ ar, const double *__restrict__ ai, double s1,
                           const double *__restrict__ br, const double *__restrict__ bi, double s2, double sign,
                           double *__restrict__ outr, double *__restrict__ outi, int accumulate)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double xr = ar[i], xi = s1 * ai[i], yr = br[i], yi = s2 * bi[i];
        double pr = sign * (xr * yr - xi * yi), pi = sign * (xr * yi + xi * yr);
        if (accumulate) { pr += outr[i]; pi += outi[i]; }
        outr[i] = pr; outi[i] = pi;
    }
}

// Polarization inverse-noise weighting with a QU cross term, in place (alm_filter_ninv.apply_map, opfilt_pp.py:295-300 and
// opfilt_tp.py:321-326):  (Q, U) <- (nqq Q + nqu U, nqu Q + nuu U).  One pass: 5 reads + 2 writes per pixel.
__global__ void k_map_qu_weight(int64_t n, double *__restrict__ q_, double *__restrict__ u_, const double *__restrict__ nqq,
                                const double *__restrict__ nqu, const double *__restrict__ nuu, int64_t bstride)
{
    double *__restrict__ q = q_ + blockIdx.y * bstride, *__restrict__ u = u_ + blockIdx.y * bstride;  // batch entry blockIdx.y
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double a = q[i], b = u[i], x = nqu[i];
        q[i] = a * nqq[i] + x * b;
        u[i] = b * nuu[i] + x * a;
    }
}

// Real-space product of the lensing estimators in one pass (qest.py:254-257 T part, :273-278 P part, :318-322 MV sum):
//   d = (rep - i imp)(g3 + i c3) - (rep + i imp)(g1 - i c1) + tmap (gt + i ct);  either part may be absent (null).
__global__ void k_qe_lens_product(int64_t n, const double *__restrict__ tmap, const double *__restrict__ gt, const double *__restrict__ ct,
                                  const double *__restrict__ rep, const double *__restrict__ imp, const double *__restrict__ g3,
                                  const double *__restrict__ c3, const double *__restrict__ g1, const double *__restrict__ c1,
                                  double *__restrict__ outr, double *__restrict__ outi)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double dr = 0., di = 0.;
        if (rep) {
            const double pr = rep[i], pi = imp[i], a3 = g3[i], b3 = c3[i], a1 = g1[i], b1 = c1[i];
            dr = (pr * a3 + pi * b3) - (pr * a1 + pi * b1);
            di = (pr * b3 - pi * a3) - (pi * a1 - pr * b1);
        }
        if (tmap) {
            const double t = tmap[i];
            dr = fma(t, gt[i], dr); di = fma(t, ct[i], di);
        }
        outr[i] = dr; outi[i] = di;
    }
}

// 16 independent FMA chains per lane: the FP64 vector-FMA issue ceiling of the chip.  MODE selects where the two
// non-accumulator operands live: 0 one VGPR + one SGPR, 1 both SGPR (wave-uniform), 2 both VGPR (three vector sources)
template <int MODE>
__global__ __launch_bounds__(256) void k_fma_peak(int iters, double xs, double ys, double *out)
{
    double a[16];
    const double xv = 1.0 + 1e-9 * threadIdx.x, yv = 1e-12 * threadIdx.x;
    const double x = MODE == 1 ? xs : xv, y = MODE == 2 ? yv : ys;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = i * 0.125;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = fma(a[i], x, y);
    }
    double s = 0.;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678) out[0] = s;
}

// Device -> pinned-host copy on a handful of workgroups (results crossing PCIe while the next reconstruction computes).
// The runtime's own blit kernel for hipMemcpyAsync fills the GPU with workgroups that sit on PCIe latency and slowed the
// concurrently running ring-FFT kernels by 40 %; posted PCIe writes need only a few waves in flight to saturate the link.
__global__ __launch_bounds__(256) void k_copy_slim(const double2 *__restrict__ src, double2 *__restrict__ dst, int64_t n2, const double *__restrict__ src_tail,
                                                   double *__restrict__ dst_tail, int ntail)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

// ---- CG vector primitives (cd_solve.py:53-107): one launch each, scalars stay on the device -----------------------
// dot = sum_{l >= lmin} sum_m w_m Re(a_lm conj(b_lm)), w_0 = 1, w_{m>0} = 2  (= sum_l (2l + 1) C_l^{ab}; opfilt_tt.py:43-51).
// One workgroup walks the m-major array with a fixed thread stride and reduces in a fixed tree: deterministic.
constexpr int kDotThreads = 1024;
__device__ __forceinline__ double block_sum_1024(double s, double *red)
{
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = kDotThreads / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    return red[0];
}
// weight of entry i of an m-major alm array: 1 on the m = 0 row (i <= lmax), 2 elsewhere, 0 for l < lmin.  Entries with
// l < lmin have m < lmin, i.e. they sit in the first lmin rows: only there is the row searched (lmin is 0 or 2 in practice).
__device__ __forceinline__ double alm_dot_weight(int lmax, int lmin, int64_t i)
{
    double w = i <= lmax ? 1.0 : 2.0;
    if (lmin > 0 && i < (int64_t)lmin * (2 * lmax + 1 - lmin) / 2 + lmin) {
        int m = 0;
        while (m + 1 < lmin && (int64_t)(m + 1) * (2 * lmax + 1 - (m + 1)) / 2 + (m + 1) <= i) ++m;
        if (i - (int64_t)m * (2 * lmax + 1 - m) / 2 < lmin) w = 0.0;
    }
    return w;
}
__device__ __forceinline__ double alm_dot_partial(int lmax, int lmin, const double2 *__restrict__ a, const double2 *__restrict__ b,
                                                  int64_t first, int64_t stride, int64_t nalm)
{
    double s = 0.0;
    for (int64_t i = first; i < nalm; i += 4 * stride) {
        double2 x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four independent loads in flight per thread
            const int64_t j = i + u * stride;
            const bool ok = j < nalm;
            x[u] = ok ? a[j] : make_double2(0., 0.);
            y[u] = ok ? b[j] : make_double2(0., 0.);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s = fma(alm_dot_weight(lmax, lmin, i + u * stride), x[u].x * y[u].x + x[u].y * y[u].y, s);
    }
    return s;
}
// The scalar product is left as kDotParts per-workgroup partial sums; whoever consumes it (k_axpy_dev, the host) adds
// them in index order.  One launch, no atomics, no cross-XCD fence, bit-reproducible.
constexpr int kDotParts = 64;
__global__ __launch_bounds__(kDotThreads) void k_alm_dot_parts(int lmax, int lmin, const double2 *__restrict__ a_, const double2 *__restrict__ b_,
                                                               int accumulate, double *__restrict__ parts_)
{
    __shared__ double red[kDotThreads];
    const int64_t nalm = (int64_t)(lmax + 1) * (lmax + 2) / 2;
    const double2 *__restrict__ a = a_ + blockIdx.y * nalm, *__restrict__ b = b_ + blockIdx.y * nalm;
    double *__restrict__ parts = parts_ + blockIdx.y * kDotParts;
    const double tot = block_sum_1024(alm_dot_partial(lmax, lmin, a, b, (int64_t)blockIdx.x * kDotThreads + threadIdx.x,
                                                      (int64_t)gridDim.x * kDotThreads, nalm), red);
    if (threadIdx.x == 0) parts[blockIdx.x] = accumulate ? parts[blockIdx.x] + tot : tot;
}

__device__ __forceinline__ double dot_parts_sum(const double *__restrict__ parts)
{
    double s = 0.0;
    for (int i = 0; i < kDotParts; ++i) s += parts[i];
    return s;
}
// y += sign * num / den * x, num and den given as partial sums in device memory (den may be null: 1)
__global__ void k_axpy_dev(int64_t n, const double *__restrict__ num_, const double *__restrict__ den_, double sign,
                           const double *__restrict__ x_, double *__restrict__ y_)
{
    const double *__restrict__ num = num_ + blockIdx.y * kDotParts, *__restrict__ den = den_ ? den_ + blockIdx.y * kDotParts : nullptr;
    const double *__restrict__ x = x_ + blockIdx.y * n;
    double *__restrict__ y = y_ + blockIdx.y * n;
    __shared__ double cs;
    if (threadIdx.x == 0) cs = den ? sign * dot_parts_sum(num) * (1.0 / dot_parts_sum(den)) : sign * dot_parts_sum(num);
    __syncthreads();
    const double c = cs;
    const int64_t n2 = n >> 1;
    const double2 *__restrict__ x2 = reinterpret_cast<const double2 *>(x);
    double2 *__restrict__ y2 = reinterpret_cast<double2 *>(y);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 u = x2[i];
        double2 v = y2[i];
        v.x = fma(c, u.x, v.x); v.y = fma(c, u.y, v.y);
        y2[i] = v;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = fma(c, x[n - 1], y[n - 1]);
}

// ---- scalar products and the vector updates that consume them in ONE launch ---------------------------------------------------
// cd_solve.py:66-84 is  dTAd = <d, q>, delta = <d, r>, x += (delta / dTAd) d, r -= (delta / dTAd) q  and :96-103 is
// s -= (<s, q'> / dTAd') d'.  Each is "scalar products over all fields, then updates scaled by their ratio": the first kDotParts
// workgroups form the partial sums exactly as k_alm_dot_parts does, a grid-wide barrier makes them visible, and every workgroup
// applies the updates as k_axpy_dev does (same arithmetic, bit-identical results).  The barrier is an arrival counter + generation
// word in device memory (agent-scope atomics; the release / acquire fences write back and invalidate the XCD-private L2s); all
// workgroups are co-resident (at most one per CU).  A waiter gives up after ~0.5 s and raises bar[2] instead of hanging the GPU.
struct CgFused {
    int nf, lmin;
    int lmax[3];
    const double2 *a[3], *b1[3], *b2[3];
    double2 *y1[3], *y2[3];
    const double2 *x1[3], *x2[3];
};
constexpr int kCgBlocks = 256;
__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned nblocks)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        // one release (L2 write-back) on arrival, relaxed polling, one acquire (invalidate) on the way out
        const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned prev = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nblocks - 1) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int spins = 0;
            while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 24)) { __hip_atomic_store(&bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
__device__ __forceinline__ double dot_parts_sum_coherent(const double *parts)
{
    double s = 0.0;
    for (int i = 0; i < kDotParts; ++i) s += __hip_atomic_load(&parts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return s;
}
// MODE 0: products, barrier, updates; 1: products only; 2: updates only (launched after a MODE 1 launch: no barrier, and the
// updates of all fields still share one launch)
// Batch entry blockIdx.y: arrays offset by whole alm arrays, its own partial sums, its own step length (`active`, optional: one
// double per entry, 1 or 0 -- an entry whose solve has converged keeps its vectors, cd_solve with per-entry stopping).
template <int MODE>
__global__ __launch_bounds__(kDotThreads) void k_cg_fused(CgFused f, double *__restrict__ parts1_, double *__restrict__ parts2_,
                                                          const double *__restrict__ den_, double sign1, double sign2, unsigned *bar,
                                                          const double *__restrict__ active)
{
    __shared__ double red[kDotThreads];
    __shared__ double cs;
    const int bq = blockIdx.y;
    double *__restrict__ parts1 = parts1_ + bq * kDotParts, *__restrict__ parts2 = parts2_ ? parts2_ + bq * kDotParts : nullptr;
    const double *__restrict__ den = den_ ? den_ + bq * kDotParts : nullptr;
    if (MODE != 2 && blockIdx.x < kDotParts) {
        double t1 = 0.0, t2 = 0.0;
        for (int k = 0; k < f.nf; ++k) {
            const int64_t nalm = (int64_t)(f.lmax[k] + 1) * (f.lmax[k] + 2) / 2;
            const int64_t first = (int64_t)blockIdx.x * kDotThreads + threadIdx.x, stride = (int64_t)kDotParts * kDotThreads;
            const int64_t o = bq * nalm;  // this batch entry's arrays
            const double u1 = block_sum_1024(alm_dot_partial(f.lmax[k], f.lmin, f.a[k] + o, f.b1[k] + o, first, stride, nalm), red);
            t1 = k == 0 ? u1 : t1 + u1;
            __syncthreads();
            if (f.b2[0]) {
                const double u2 = block_sum_1024(alm_dot_partial(f.lmax[k], f.lmin, f.a[k] + o, f.b2[k] + o, first, stride, nalm), red);
                t2 = k == 0 ? u2 : t2 + u2;
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) {
            parts1[blockIdx.x] = t1;
            if (f.b2[0]) parts2[blockIdx.x] = t2;
        }
    }
    if (MODE == 1) return;
    if (MODE == 0) grid_barrier(bar, gridDim.x);
    if (threadIdx.x == 0) {
        double num, dn;
        if (MODE == 0) {
            num = den ? dot_parts_sum_coherent(parts1) : dot_parts_sum_coherent(parts2);
            dn = den ? dot_parts_sum_coherent(den) : dot_parts_sum_coherent(parts1);
        } else {
            num = den ? dot_parts_sum(parts1) : dot_parts_sum(parts2);
            dn = den ? dot_parts_sum(den) : dot_parts_sum(parts1);
        }
        cs = num * (1.0 / dn);
    }
    // a frozen entry stands still whatever its products and directions are (0 / 0 and NaN directions for an all-zero right-hand
    // side: 0 x NaN would still be NaN, so its vectors are not touched at all)
    if (active && active[bq] == 0.0) return;
    __syncthreads();
    const double c1 = sign1 * cs, c2 = sign2 * cs;  // sign = +-1: same value as k_axpy_dev's sign * num * (1 / den)
    for (int k = 0; k < f.nf; ++k) {
        const int64_t nalm = (int64_t)(f.lmax[k] + 1) * (f.lmax[k] + 2) / 2;
        const int64_t o = bq * nalm;
        for (int64_t i = o + (int64_t)blockIdx.x * kDotThreads + threadIdx.x; i < o + nalm; i += (int64_t)gridDim.x * kDotThreads) {
            const double2 u = f.x1[k][i];
            double2 v = f.y1[k][i];
            v.x = fma(c1, u.x, v.x); v.y = fma(c1, u.y, v.y);
            f.y1[k][i] = v;
            if (f.y2[0]) {
                const double2 u2 = f.x2[k][i];
                double2 w = f.y2[k][i];
                w.x = fma(c2, u2.x, w.x); w.y = fma(c2, u2.y, w.y);
                f.y2[k][i] = w;
            }
        }
    }
}

// The updates of k_cg_fused<2> from scalar products that arrive as `npre` per-workgroup partial sums of the kernel that produced the vector (PostDots of
// k_post0 / k_posts: <d, q> and <d, r> formed while q is written; k_gemv_split / k_alm_splice: <s, q'> formed while s is written) instead of the
// kDotParts sums of a k_cg_fused<1> launch: every workgroup adds them in a fixed order (thread t takes entries t, t + 1024, ..., then the tree of
// block_sum_1024).  Same roles as in k_cg_fused: den given: step length = sum(pre1) / sum(den); else sum(pre2) / sum(pre1).  Workgroup 0 leaves the
// totals behind as canonical kDotParts-entry partial sums {total, 0, ...} (parts1 <- pre1, parts2 <- pre2) for the cache of cd_solve and the monitors.
__global__ __launch_bounds__(kDotThreads) void k_cg_axpy_pre(CgFused f, int npre, const double *__restrict__ pre1_, const double *__restrict__ pre2_,
                                                             const double *__restrict__ den_, double *__restrict__ parts1_, double *__restrict__ parts2_,
                                                             double sign1, double sign2, const double *__restrict__ active, int y1_assign)
{
    __shared__ double red[2][kDotThreads / 64];
    const int bq = blockIdx.y;
    const double *__restrict__ pre1 = pre1_ + (int64_t)bq * npre, *__restrict__ pre2 = pre2_ ? pre2_ + (int64_t)bq * npre : nullptr;
    const double *__restrict__ den = den_ ? den_ + bq * kDotParts : nullptr;
    // thread t adds entries t, t + 1024, ...; lanes by shuffles, the 16 wave sums in index order by every thread: one barrier, fixed order
    double u1 = 0.0, u2 = 0.0;
    for (int i = threadIdx.x; i < npre; i += kDotThreads) { u1 += pre1[i]; if (pre2) u2 += pre2[i]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { u1 += __shfl_down(u1, off, 64); u2 += __shfl_down(u2, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = u1; red[1][threadIdx.x >> 6] = u2; }
    __syncthreads();
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int w = 0; w < kDotThreads / 64; ++w) { t1 += red[0][w]; t2 += red[1][w]; }
    const double cs = den ? t1 * (1.0 / dot_parts_sum(den)) : t2 * (1.0 / t1);
    if (blockIdx.x == 0 && threadIdx.x < kDotParts) {
        if (parts1_) parts1_[bq * kDotParts + threadIdx.x] = threadIdx.x == 0 ? t1 : 0.0;
        if (parts2_ && pre2) parts2_[bq * kDotParts + threadIdx.x] = threadIdx.x == 0 ? t2 : 0.0;
    }
    // y1_assign: y1 = c x1 instead of y1 += c x1 -- the first step of a solve that starts from x = 0 writes its solution vector instead of
    // reading zeros somebody had to store first (an entry that stands still gets the zeros)
    const bool still = active && active[bq] == 0.0;
    if (still && !y1_assign) return;
    const double c1 = still ? 0.0 : sign1 * cs, c2 = sign2 * cs;
    for (int k = 0; k < f.nf; ++k) {
        const int64_t nalm = (int64_t)(f.lmax[k] + 1) * (f.lmax[k] + 2) / 2;
        const int64_t o = bq * nalm;
        for (int64_t i = o + (int64_t)blockIdx.x * kDotThreads + threadIdx.x; i < o + nalm; i += (int64_t)gridDim.x * kDotThreads) {
            const double2 u = f.x1[k][i];
            double2 v = y1_assign ? make_double2(0., 0.) : f.y1[k][i];
            if (still) { f.y1[k][i] = v; continue; }
            v.x = fma(c1, u.x, v.x); v.y = fma(c1, u.y, v.y);
            f.y1[k][i] = v;
            if (f.y2[0]) {
                const double2 w2 = f.x2[k][i];
                double2 w = f.y2[k][i];
                w.x = fma(c2, w2.x, w.x); w.y = fma(c2, w2.y, w.y);
                f.y2[k][i] = w;
            }
        }
    }
}

// ---- N^-1 with template marginalisation in two launches (opfilt_tt.py:196-205) ----------------------------------------
// t <- N^-1 t - N^-1 P (P^t N^-1 P)^-1 P^t N^-1 t with P (nmodes x n) and R = (P^t N^-1 P)^-1 (P . N^-1) (nmodes x n) given:
//   pass 1: t <- n_inv t and the per-workgroup partial sums of c_k = sum_i P_ki t_i;  pass 2: t_i -= sum_k R_ki c_k.
// The partial sums are added in index order by every workgroup of pass 2: bit-reproducible, no atomics.
// (kProjParts, kProjMaxModes, kFuseModesB, kProjChunk and the workgroup bodies of the coefficient pass: tproj_device.h)
// NT threads per workgroup, gridDim.x = nparts <= kProjParts workgroups: one partial sum per workgroup and mode.  Small maps (the
// coarse multigrid levels, where this pair of kernels runs dozens of times per CG iteration) take fewer, smaller workgroups:
// the work is a few microseconds and the cost is the launch and the reduction tail.
// Block vectors: up to kProjChunk maps per workgroup pass -- the template rows (shared by the batch) are read once for all of them.
// Every map's sums are formed exactly as the one-map kernel forms them (same pixels per thread, same order): bit-identical.
template <int NT>
__global__ __launch_bounds__(NT) void k_tproj_coeffs_b(int64_t n, int nmodes, int nb, double *__restrict__ t_, const double *__restrict__ n_inv,
                                                       const double *__restrict__ pm, double *__restrict__ parts_)
{
    tproj_coeffs_b_wg<NT>(n, nmodes, nb, t_, n_inv, pm, parts_, blockIdx.x, gridDim.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void k_tproj_apply_b(int64_t n, int nmodes, int nparts, int nb, double *__restrict__ t_, const double *__restrict__ rm,
                                                       const double *__restrict__ parts_)
{
    __shared__ double c[kProjChunk][kFuseModesB];
    const int b0 = blockIdx.y * kProjChunk, nbc = min(kProjChunk, nb - b0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = wave; q < nmodes * nbc; q += 4) {
        const int b = q / nmodes, k = q % nmodes;
        const double *parts = parts_ + (int64_t)(b0 + b) * (kProjMaxModes * kProjParts);
        double v = 0.0;
        for (int j = lane; j < nparts; j += 64) v += parts[k * kProjParts + j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) c[b][k] = v;
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double r[kFuseModesB];
#pragma unroll
        for (int k = 0; k < kFuseModesB; ++k) r[k] = k < nmodes ? rm[(int64_t)k * n + i] : 0.0;
#pragma unroll
        for (int b = 0; b < kProjChunk; ++b) {
            if (b < nbc) {
                double *tb = t_ + (int64_t)(b0 + b) * n;
                double v = tb[i];
#pragma unroll
                for (int k = 0; k < kFuseModesB; ++k)
                    if (k < nmodes) v = fma(-r[k], c[b][k], v);
                tb[i] = v;
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_tproj_coeffs(int64_t n, int nmodes, double *__restrict__ t_, const double *__restrict__ n_inv,
                                                     const double *__restrict__ pm, double *__restrict__ parts_)
{
    // batch entry blockIdx.y: its own map and partial sums, shared n_inv and modes
    tproj_coeffs_wg<NT>(n, nmodes, t_ + blockIdx.y * n, n_inv, pm, parts_ + blockIdx.y * (kProjMaxModes * kProjParts), blockIdx.x, gridDim.x);
}
// c_k = sum of the nparts partial sums of mode k, by one wavefront per mode in a fixed order (lane j takes parts j, j + 64, ...,
// then a fixed shuffle tree): one barrier instead of a shared-memory tree per mode
__global__ __launch_bounds__(256) void k_tproj_apply(int64_t n, int nmodes, int nparts, double *__restrict__ t_, const double *__restrict__ rm,
                                                     const double *__restrict__ parts_)
{
    double *__restrict__ t = t_ + blockIdx.y * n;
    const double *__restrict__ parts = parts_ + blockIdx.y * (kProjMaxModes * kProjParts);
    __shared__ double c[kProjMaxModes];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < nmodes; k += 4) {
        double v = 0.0;
        for (int j = lane; j < nparts; j += 64) v += parts[k * kProjParts + j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) c[k] = v;
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v = t[i];
        for (int k = 0; k < nmodes; ++k) v = fma(-rm[(int64_t)k * n + i], c[k], v);
        t[i] = v;
    }
}

// ---- monopole + dipole marginalisation with the templates evaluated from the ring geometry ---------------------------------------
// The templates of template_removal.py:116-150 are (1, x, y, z) of the pixel centres.  Read as stored maps (k_tproj_*) they cost 4
// template rows in the coefficient pass and 4 rows of R in the projection pass: 11 passes over a map for the two launches.  Here a
// workgroup owns a ring pair (z = +- cos theta, sin theta, phi_j = phi0 + 2 pi j / n known from the plan) and evaluates x, y per pixel
// with one sincospi: 4 passes (map in; map in, n_inv in, map out), at the price of a third tiny launch that adds the ring sums.
//   coeffs:  parts[b][k][pair] = sum over the pair's pixels of (1, x, y, z)_k u_i,  u = n_inv t (stored back) or t itself (weighted = 1)
//   reduce:  c_k = sum_pairs parts (fixed order), d = Pinv c  (Pinv = (P^t N^-1 P)^-1, symmetric 4 x 4)
//   apply:   t_i -= n_inv_i (d_0 + d_1 x_i + d_2 y_i + d_3 z_i)
// Up to kProjChunk maps of a block share the geometry per pass.  Deterministic (fixed trees), batch entries bit-identical to single calls.
__global__ __launch_bounds__(256) void k_tproj_md_coeffs(DevPlan P, int nb, double *__restrict__ t_, const double *__restrict__ n_inv, int weighted,
                                                         double *__restrict__ parts)
{
    __shared__ double red[kProjChunk][4][4];
    const int ip = blockIdx.x, b0 = blockIdx.y * kProjChunk, nbc = min(kProjChunk, nb - b0);
    const int n = P.nphi[ip];
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    const double z = P.cth[ip], s = P.sth[ip], ph0 = P.phi0[ip] * 0.31830988618379067154, inv_n2 = 2.0 / n;
    double acc[kProjChunk][4];
#pragma unroll
    for (int b = 0; b < kProjChunk; ++b) acc[b][0] = acc[b][1] = acc[b][2] = acc[b][3] = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        double sn, cs;
        sincospi(ph0 + j * inv_n2, &sn, &cs);
        const double x = s * cs, y = s * sn;
        const double wn = (weighted || !n_inv) ? 1.0 : n_inv[on + j], ws = (weighted || !n_inv || os < 0) ? 1.0 : n_inv[os + j];
#pragma unroll
        for (int b = 0; b < kProjChunk; ++b) {
            if (b < nbc) {
                double *tb = t_ + (int64_t)(b0 + b) * P.npix;
                double un = tb[on + j], us = os >= 0 ? tb[os + j] : 0.0;
                if (!weighted && n_inv) {
                    un *= wn; tb[on + j] = un;
                    if (os >= 0) { us *= ws; tb[os + j] = us; }
                }
                const double sum = un + us;
                acc[b][0] += sum;
                acc[b][1] = fma(x, sum, acc[b][1]);
                acc[b][2] = fma(y, sum, acc[b][2]);
                acc[b][3] = fma(z, un - us, acc[b][3]);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int b = 0; b < kProjChunk; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double v = acc[b][k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) red[b][k][wave] = v;
        }
    __syncthreads();
    if ((int)threadIdx.x < 4 * nbc) {
        const int b = threadIdx.x >> 2, k = threadIdx.x & 3;
        parts[((int64_t)(b0 + b) * 4 + k) * P.npairs + ip] = (red[b][k][0] + red[b][k][1]) + (red[b][k][2] + red[b][k][3]);
    }
}
__global__ __launch_bounds__(256) void k_tproj_md_reduce(int npairs, const double *__restrict__ parts, const double *__restrict__ pinv,
                                                         double *__restrict__ d)
{
    __shared__ double c[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, k = threadIdx.x >> 6;  // one wavefront per mode
    double v = 0.0;
    for (int j = lane; j < npairs; j += 64) v += parts[((int64_t)b * 4 + k) * npairs + j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) c[k] = v;
    __syncthreads();
    if (threadIdx.x < 4) {
        const int r = threadIdx.x;
        d[b * 4 + r] = ((pinv[4 * r] * c[0] + pinv[4 * r + 1] * c[1]) + pinv[4 * r + 2] * c[2]) + pinv[4 * r + 3] * c[3];
    }
}
__global__ __launch_bounds__(256) void k_tproj_md_apply(DevPlan P, int nb, double *__restrict__ t_, const double *__restrict__ n_inv,
                                                        const double *__restrict__ d_)
{
    const int ip = blockIdx.x, b0 = blockIdx.y * kProjChunk, nbc = min(kProjChunk, nb - b0);
    const int n = P.nphi[ip];
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    const double z = P.cth[ip], s = P.sth[ip], ph0 = P.phi0[ip] * 0.31830988618379067154, inv_n2 = 2.0 / n;
    double d[kProjChunk][4];
#pragma unroll
    for (int b = 0; b < kProjChunk; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) d[b][k] = b < nbc ? d_[(b0 + b) * 4 + k] : 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        double sn, cs;
        sincospi(ph0 + j * inv_n2, &sn, &cs);
        const double x = s * cs, y = s * sn;
        const double wn = n_inv[on + j], ws = os >= 0 ? n_inv[os + j] : 0.0;
#pragma unroll
        for (int b = 0; b < kProjChunk; ++b) {
            if (b < nbc) {
                double *tb = t_ + (int64_t)(b0 + b) * P.npix;
                const double h = fma(d[b][2], y, fma(d[b][1], x, d[b][0]));  // d0 + d1 x + d2 y
                tb[on + j] = fma(-wn, fma(d[b][3], z, h), tb[on + j]);
                if (os >= 0) tb[os + j] = fma(-ws, fma(-d[b][3], z, h), tb[os + j]);
            }
        }
    }
}

// One partial sum per workgroup of 256 threads, fixed tree (lanes by shuffles, then the four wave sums): the share of this workgroup in a scalar
// product that the kernel forms about the vector it writes (k_gemv_split, k_alm_splice: <s, q'> of cd_solve.py:96-103 without a launch of its own)
__device__ __forceinline__ void wg256_sum_store(double t, double *red4, double *dst)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) *dst = (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

// ---- dense mat-vec y = A x (the dense coarse preconditioner, dense.py:118-119,201-202,284-285) -------------------------
// A row-major (nrows x ncols, leading dimension lda).  One wavefront per row: every trip the wave reads 1 KiB of the row
// with 16-byte loads per lane (VEC = 2), four trips in flight; x comes from L2 / L1 (34 KiB at the 4290-column T block).
// The 147 MB T matrix lives in the 256 MB Infinity Cache between applications: the floor is its streaming rate, not HBM.
// Lane partial sums are added in a fixed tree: bit-reproducible.
template <int VEC>
__global__ __launch_bounds__(256) void k_gemv(int nrows, int ncols, int64_t lda, const double *__restrict__ A, const double *__restrict__ x,
                                              double *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const double *__restrict__ a = A + (int64_t)row * lda;
    double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
    if constexpr (VEC == 2) {
        const double2 *__restrict__ a2 = reinterpret_cast<const double2 *>(a);
        const double2 *__restrict__ x2 = reinterpret_cast<const double2 *>(x);
        const int n2 = ncols >> 1;
        int c = lane;
        for (; c + 448 < n2; c += 512) {  // eight 16-byte loads of the row in flight per lane: the matrix streams from the Infinity Cache
            const double2 u0 = a2[c], u1 = a2[c + 64], u2 = a2[c + 128], u3 = a2[c + 192];
            const double2 u4 = a2[c + 256], u5 = a2[c + 320], u6 = a2[c + 384], u7 = a2[c + 448];
            const double2 v0 = x2[c], v1 = x2[c + 64], v2 = x2[c + 128], v3 = x2[c + 192];
            const double2 v4 = x2[c + 256], v5 = x2[c + 320], v6 = x2[c + 384], v7 = x2[c + 448];
            s0 = fma(u0.x, v0.x, s0); s0 = fma(u0.y, v0.y, s0);
            s1 = fma(u1.x, v1.x, s1); s1 = fma(u1.y, v1.y, s1);
            s2 = fma(u2.x, v2.x, s2); s2 = fma(u2.y, v2.y, s2);
            s3 = fma(u3.x, v3.x, s3); s3 = fma(u3.y, v3.y, s3);
            s0 = fma(u4.x, v4.x, s0); s0 = fma(u4.y, v4.y, s0);
            s1 = fma(u5.x, v5.x, s1); s1 = fma(u5.y, v5.y, s1);
            s2 = fma(u6.x, v6.x, s2); s2 = fma(u6.y, v6.y, s2);
            s3 = fma(u7.x, v7.x, s3); s3 = fma(u7.y, v7.y, s3);
        }
        for (; c + 192 < n2; c += 256) {
            const double2 u0 = a2[c], u1 = a2[c + 64], u2 = a2[c + 128], u3 = a2[c + 192];
            const double2 v0 = x2[c], v1 = x2[c + 64], v2 = x2[c + 128], v3 = x2[c + 192];
            s0 = fma(u0.x, v0.x, s0); s0 = fma(u0.y, v0.y, s0);
            s1 = fma(u1.x, v1.x, s1); s1 = fma(u1.y, v1.y, s1);
            s2 = fma(u2.x, v2.x, s2); s2 = fma(u2.y, v2.y, s2);
            s3 = fma(u3.x, v3.x, s3); s3 = fma(u3.y, v3.y, s3);
        }
        for (; c < n2; c += 64) { const double2 u = a2[c], v = x2[c]; s0 = fma(u.x, v.x, s0); s0 = fma(u.y, v.y, s0); }
        if ((ncols & 1) && lane == 0) s1 = fma(a[ncols - 1], x[ncols - 1], s1);
    } else {
        for (int c = lane; c < ncols; c += 64) s0 = fma(a[c], x[c], s0);
    }
    double v = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) y[row] = v;
}

// pre_op_split around the dense block in ONE launch (multigrid.py:163-182 with dense.py:118-119): out (band-limit lmax_hi) =
//   [l <= lmax_lo:  A x_lo,  x_lo = the entries l <= lmax_lo of hi (the truncating alm_copy, read in place through `map`)]
//   [l  > lmax_lo:  fl_hi[l] hi]                                                  (the diagonal preconditioner, the splice)
// A acts on the interleaved (re, im) view of the lmax_lo alm layout (pre_op_dense's flat matrix); map[e] = index in the lmax_hi layout
// of entry e of the lmax_lo layout.  Workgroups [0, ngemv) are those of k_gemv<2> -- same lane partition, chains and tree, so the low
// part is bit-identical to alm_copy + k_gemv<2> + k_alm_splice -- the others write the high part.
struct SplitFields { const double2 *hi[2]; double2 *out[2]; const double *fl[2]; };
// optional: pre[workgroup] = this workgroup's share of sum over the fields of <out, q> (weights of k_alm_dot_parts, l < lmin excluded)
struct SplitDot { const double2 *q[2] = {nullptr, nullptr}; double *pre = nullptr; int lmin = 0; };
template <int NF>
__global__ __launch_bounds__(256) void k_gemv_split(int nrows, int64_t lda, const double *__restrict__ A, SplitFields F, const int *__restrict__ map,
                                                    int lmax_lo, int lmax_hi, int ngemv, SplitDot D)
{
    __shared__ double red4[4];
    const int lane = threadIdx.x & 63;
    const int nlo = (lmax_lo + 1) * (lmax_lo + 2) / 2;  // complex entries of one field at lmax_lo
    if ((int)blockIdx.x >= ngemv) {  // high multipoles: 4 workgroups per (field, m), as k_alm_splice
        const int b = blockIdx.x - ngemv, f = NF == 1 ? 0 : b / (4 * (lmax_hi + 1)), bb = b - f * 4 * (lmax_hi + 1), m = bb >> 2;
        const int64_t bh = (int64_t)m * (2 * lmax_hi + 1 - m) / 2;
        const int lstart = m > lmax_lo + 1 ? m : lmax_lo + 1;
        const double2 *__restrict__ hi = F.hi[f];
        const double *__restrict__ fl = F.fl[f];
        double2 *__restrict__ out = F.out[f];
        double t = 0.0;
        for (int l = lstart + (bb & 3) * 256 + threadIdx.x; l <= lmax_hi; l += 4 * 256) {
            double2 v = hi[bh + l];
            v.x *= fl[l]; v.y *= fl[l];
            out[bh + l] = v;
            if (D.pre && l >= D.lmin) { const double2 qv = D.q[f][bh + l]; t = fma(m == 0 ? 1.0 : 2.0, v.x * qv.x + v.y * qv.y, t); }
        }
        if (D.pre) wg256_sum_store(t, red4, D.pre + blockIdx.x);
        return;
    }
    // (the last workgroup of the mat-vec part may hold fewer than four rows: its idle waves skip the row and fall through with t = 0 to
    // the ONE wg256_sum_store every wave of the workgroup executes -- a barrier is never reached from two program points)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool valid = row < nrows;
    double t = 0.0;
    if (valid) {
        const double2 *__restrict__ a2 = reinterpret_cast<const double2 *>(A + (int64_t)row * lda);
        // entry c of the concatenated low-band-limit vector [field 0 | field 1]: read in place from the field's full array
        auto x = [&](int c) -> double2 {
            if constexpr (NF == 1) return F.hi[0][map[c]];
            else return c < nlo ? F.hi[0][map[c]] : F.hi[1][map[c - nlo]];
        };
        double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
        const int n2 = NF * nlo;
        int c = lane;
        for (; c + 448 < n2; c += 512) {
            const double2 u0 = a2[c], u1 = a2[c + 64], u2 = a2[c + 128], u3 = a2[c + 192];
            const double2 u4 = a2[c + 256], u5 = a2[c + 320], u6 = a2[c + 384], u7 = a2[c + 448];
            const double2 v0 = x(c), v1 = x(c + 64), v2 = x(c + 128), v3 = x(c + 192);
            const double2 v4 = x(c + 256), v5 = x(c + 320), v6 = x(c + 384), v7 = x(c + 448);
            s0 = fma(u0.x, v0.x, s0); s0 = fma(u0.y, v0.y, s0);
            s1 = fma(u1.x, v1.x, s1); s1 = fma(u1.y, v1.y, s1);
            s2 = fma(u2.x, v2.x, s2); s2 = fma(u2.y, v2.y, s2);
            s3 = fma(u3.x, v3.x, s3); s3 = fma(u3.y, v3.y, s3);
            s0 = fma(u4.x, v4.x, s0); s0 = fma(u4.y, v4.y, s0);
            s1 = fma(u5.x, v5.x, s1); s1 = fma(u5.y, v5.y, s1);
            s2 = fma(u6.x, v6.x, s2); s2 = fma(u6.y, v6.y, s2);
            s3 = fma(u7.x, v7.x, s3); s3 = fma(u7.y, v7.y, s3);
        }
        for (; c + 192 < n2; c += 256) {
            const double2 u0 = a2[c], u1 = a2[c + 64], u2 = a2[c + 128], u3 = a2[c + 192];
            const double2 v0 = x(c), v1 = x(c + 64), v2 = x(c + 128), v3 = x(c + 192);
            s0 = fma(u0.x, v0.x, s0); s0 = fma(u0.y, v0.y, s0);
            s1 = fma(u1.x, v1.x, s1); s1 = fma(u1.y, v1.y, s1);
            s2 = fma(u2.x, v2.x, s2); s2 = fma(u2.y, v2.y, s2);
            s3 = fma(u3.x, v3.x, s3); s3 = fma(u3.y, v3.y, s3);
        }
        for (; c < n2; c += 64) { const double2 u = a2[c], v = x(c); s0 = fma(u.x, v.x, s0); s0 = fma(u.y, v.y, s0); }
        double v = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) {
            const int f = NF == 1 ? 0 : row / (2 * nlo), r = row - f * 2 * nlo;
            const int64_t o = 2 * (int64_t)map[r >> 1] + (r & 1);
            reinterpret_cast<double *>(F.out[f])[o] = v;
            if (D.pre) t = alm_dot_weight(lmax_lo, D.lmin, r >> 1) * (v * reinterpret_cast<const double *>(D.q[f])[o]);
        }
    }
    if (D.pre) wg256_sum_store(t, red4, D.pre + blockIdx.x);
}

// The same for NB right-hand sides at once (Y[b] = A X[b], X: [nb][ncols], Y: [nb][nrows]): the block vectors of a batched
// conjugate-gradient solve.  The matrix is read once for all of them (the 147 MB T block streams from the Infinity Cache at the
// same rate as for one vector; 2 nb flops per matrix entry is far below the FMA rate: no MFMA needed for nb <= 8).  Every entry's sum
// is formed exactly as k_gemv<2> forms it (same lane partition, same four chains, same tree): bit-identical to nb separate calls.
template <int NB>
__global__ __launch_bounds__(256) void k_gemv_nb(int nrows, int ncols, int64_t lda, const double *__restrict__ A, int nb, const double *__restrict__ X,
                                                 double *__restrict__ Y)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const double2 *__restrict__ a2 = reinterpret_cast<const double2 *>(A + (int64_t)row * lda);
    const double2 *__restrict__ x2[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) x2[b] = reinterpret_cast<const double2 *>(X + (int64_t)(b < nb ? b : nb - 1) * ncols);
    double s[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b][0] = s[b][1] = s[b][2] = s[b][3] = 0.;
    const int n2 = ncols >> 1;
    int c = lane;
    for (; c + 448 < n2; c += 512) {
        double2 u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = a2[c + 64 * j];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double2 v = x2[b][c + 64 * j];
                s[b][j & 3] = fma(u[j].x, v.x, s[b][j & 3]); s[b][j & 3] = fma(u[j].y, v.y, s[b][j & 3]);
            }
        }
    }
    for (; c + 192 < n2; c += 256) {
        double2 u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = a2[c + 64 * j];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double2 v = x2[b][c + 64 * j];
                s[b][j] = fma(u[j].x, v.x, s[b][j]); s[b][j] = fma(u[j].y, v.y, s[b][j]);
            }
        }
    }
    for (; c < n2; c += 64) {
        const double2 u = a2[c];
#pragma unroll
        for (int b = 0; b < NB; ++b) { const double2 v = x2[b][c]; s[b][0] = fma(u.x, v.x, s[b][0]); s[b][0] = fma(u.y, v.y, s[b][0]); }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        double v = (s[b][0] + s[b][1]) + (s[b][2] + s[b][3]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && b < nb) Y[(int64_t)b * nrows + row] = v;
    }
}

// out (band-limit lmax_hi) = alm_lo for l <= lsplit, alm_hi above (util_alm.py:8-24)
// fl_hi (optional, lmax_hi + 1 entries): the high part is fl_hi[l] * alm_hi -- the diagonal preconditioner of pre_op_split's high
// multipoles (multigrid.py:163-182 with opfilt_tt.py:76-93) applied on the way
// dot_q / dot_pre (optional): dot_pre[(batch entry, m, workgroup of the m)] = this workgroup's share of <out, dot_q> (weights of k_alm_dot_parts)
__global__ void k_alm_splice(int lmax_lo, const double2 *__restrict__ lo_, int lmax_hi, const double2 *__restrict__ hi_, int lsplit,
                             double2 *__restrict__ out_, const double *__restrict__ fl_hi, const double2 *__restrict__ dot_q, int dot_lmin,
                             double *__restrict__ dot_pre)
{
    __shared__ double red4[4];
    double t = 0.0;
    const double2 *__restrict__ lo = lo_ + blockIdx.z * alm_count(lmax_lo), *__restrict__ hi = hi_ + blockIdx.z * alm_count(lmax_hi);
    double2 *__restrict__ out = out_ + blockIdx.z * alm_count(lmax_hi);
    const int m = blockIdx.y;
    const int64_t bh = (int64_t)m * (2 * lmax_hi + 1 - m) / 2;
    const int64_t bl = (int64_t)m * (2 * lmax_lo + 1 - m) / 2;
    for (int l = m + blockIdx.x * blockDim.x + threadIdx.x; l <= lmax_hi; l += gridDim.x * blockDim.x) {
        double2 v;
        if (l <= lsplit) v = lo[bl + l];
        else {
            v = hi[bh + l];
            if (fl_hi) { v.x *= fl_hi[l]; v.y *= fl_hi[l]; }
        }
        out[bh + l] = v;
        if (dot_pre && l >= dot_lmin) {
            const double2 qv = dot_q[blockIdx.z * alm_count(lmax_hi) + bh + l];
            t = fma(m == 0 ? 1.0 : 2.0, v.x * qv.x + v.y * qv.y, t);
        }
    }
    if (dot_pre) wg256_sum_store(t, red4, dot_pre + ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// out = a + f_l b (fwd_op: N-part + S^-1 x, opfilt_tt.py:67-73); out may alias a
__global__ void k_almxfl_add(int lmax, const double2 *a_, const double2 *__restrict__ b_, const double *__restrict__ fl, int nfl, double2 *out_)
{
    const double2 *a = a_ + blockIdx.z * alm_count(lmax), *__restrict__ b = b_ + blockIdx.z * alm_count(lmax);
    double2 *out = out_ + blockIdx.z * alm_count(lmax);
    const int m = blockIdx.y;
    const int64_t base = (int64_t)m * (2 * lmax + 1 - m) / 2;
    for (int l = m + blockIdx.x * blockDim.x + threadIdx.x; l <= lmax; l += gridDim.x * blockDim.x) {
        const double f = l < nfl ? fl[l] : 0.0;
        const double2 x = a[base + l], y = b[base + l];
        out[base + l] = make_double2(fma(f, y.x, x.x), fma(f, y.y, x.y));
    }
}

// ---- m-block sharding of one transform: phase slices between the Legendre stage (sharded by m-group) and the ring FFTs (by ring pair) ----
// buf [j][component][k][16 doubles] <-> phase [pair0 + j pstride][component][4 (mg0 + k mgstride) ... + 3][4]; one 128-byte piece per
// (pair, component, m-group), 8 lanes of 16 bytes each
__global__ void k_phase_pack(DevPlan P, int ncomp, double *__restrict__ phase, double *__restrict__ buf, int pair0, int pstride, int mg0, int mgstride,
                             int nsel, int nmsel, int unpack)
{
    const int64_t npieces = (int64_t)nsel * ncomp * nmsel;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < npieces * 8; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t piece = t >> 3;
        const int part = (int)(t & 7);
        const int k = (int)(piece % nmsel);
        const int c = (int)((piece / nmsel) % ncomp);
        const int j = (int)(piece / ((int64_t)nmsel * ncomp));
        const int ip = pair0 + j * pstride, mg = mg0 + k * mgstride;
        double2 *g = reinterpret_cast<double2 *>(phase + (((int64_t)ip * ncomp + c) * P.mstride + 4 * mg) * 4) + part;
        double2 *b = reinterpret_cast<double2 *>(buf + piece * 16) + part;
        if (unpack) *g = *b; else *b = *g;
    }
}

// The pixels of the ring pairs pair0, pair0 + pstride, ... of a map (a rank's rings of a sharded transform) as one packed buffer
// [component][pair j: north ring, south ring] and back.  Doubles in front of selected pair j: ring lengths are 4 (ip + 1) in the cap
// (ip <= nside - 2) and 4 nside in the belt; the equator (last pair) has no partner.
__host__ __device__ inline int64_t ring_pack_offset(int nside, int pair0, int pstride, int j)
{
    int64_t k = 0;
    if (pair0 <= nside - 2) { k = (nside - 2 - pair0) / pstride + 1; if (k > j) k = j; }
    return 2 * (4 * (k * (int64_t)(pair0 + 1) + (int64_t)pstride * k * (k - 1) / 2) + 4 * (int64_t)nside * (j - k));
}
__global__ __launch_bounds__(256) void k_map_pack_rings(DevPlan P, double *__restrict__ map, double *__restrict__ buf, int pair0, int pstride,
                                                        int64_t per_comp, int unpack)
{
    const int j = blockIdx.x, c = blockIdx.y;
    const int ip = pair0 + j * pstride;
    const int n = P.nphi[ip];
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    double2 *b = reinterpret_cast<double2 *>(buf + (int64_t)c * per_comp + ring_pack_offset(P.nside, pair0, pstride, j));
    double2 *mn = reinterpret_cast<double2 *>(map + (int64_t)c * P.npix + on);
    double2 *ms = os >= 0 ? reinterpret_cast<double2 *>(map + (int64_t)c * P.npix + os) : nullptr;
    for (int t = threadIdx.x; t < n / 2; t += 256) {  // (ring lengths and offsets are multiples of 4 doubles)
        if (unpack) { mn[t] = b[t]; if (ms) ms[t] = b[n / 2 + t]; }
        else { b[t] = mn[t]; if (ms) b[n / 2 + t] = ms[t]; }
    }
}

__global__ void k_alm_keep_mgroups(int lmax, double2 *__restrict__ alm_, int mg0, int mgstride)
{
    double2 *__restrict__ alm = alm_ + blockIdx.z * alm_count(lmax);
    const int m = blockIdx.y;
    const int mg = m >> 2;
    if (mg >= mg0 && (mg - mg0) % mgstride == 0) return;
    const int64_t base = (int64_t)m * (2 * lmax + 1 - m) / 2;
    for (int l = m + blockIdx.x * blockDim.x + threadIdx.x; l <= lmax; l += gridDim.x * blockDim.x) alm[base + l] = make_double2(0., 0.);
}

// ---- simulation inputs on the device (plancklens/sims/phas.py:125-195, sims/maps.py:46-77,136-173) -------------------
// Standard normal deviates as a pure function of (key, position): Philox4x32-10 (Salmon et al. 2011, the public counter-based
// generator; known-answer vectors in tests/test_sims.py) on the counter (pair index, tag), its 128 bits -> two 53-bit uniforms ->
// Box-Muller -> the deviates of positions 2 p and 2 p + 1.  No state, no tensor of deviates: a map receives sigma n(0, 1) in the one
// pass that reads and writes it, and any (seed, field, simulation, position) can be regenerated anywhere (on any rank) at will.
__device__ __forceinline__ void philox4x32_10(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0, hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += W0; k1 += W1;
    }
}

__device__ __forceinline__ double2 normal_pair(uint64_t key, uint64_t p, uint32_t tag)
{
    uint32_t c0 = (uint32_t)p, c1 = (uint32_t)(p >> 32), c2 = tag, c3 = 0u;
    philox4x32_10(c0, c1, c2, c3, (uint32_t)key, (uint32_t)(key >> 32));
    const uint64_t a = ((uint64_t)c0 | ((uint64_t)c1 << 32)) >> 11, b = ((uint64_t)c2 | ((uint64_t)c3 << 32)) >> 11;
    const double u1 = (double)(a + 1) * 0x1.0p-53;  // (0, 1]
    const double u2 = (double)b * 0x1.0p-53;        // [0, 1)
    const double rad = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    return make_double2(rad * c, rad * s);
}

// out[i] = (in ? in[i] : 0) + sigma n_i, n_i the deviate of position i under `key` (tag 0); two positions per thread, 16-byte accesses
__global__ __launch_bounds__(256) void k_map_add_normal(int64_t n, const double *in, double *out, double sigma, uint64_t key)
{
    const int64_t np = (n + 1) >> 1, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += stride) {
        const double2 g = normal_pair(key, (uint64_t)p, 0u);
        if (2 * p + 1 < n) {
            double2 v = in ? *reinterpret_cast<const double2 *>(in + 2 * p) : make_double2(0., 0.);
            v.x = fma(sigma, g.x, v.x); v.y = fma(sigma, g.y, v.y);
            *reinterpret_cast<double2 *>(out + 2 * p) = v;
        } else {
            out[2 * p] = fma(sigma, g.x, in ? in[2 * p] : 0.0);
        }
    }
}

// unit-variance harmonic coefficients (healpy layout, mmax = lmax): entry i > lmax gets (n_2i, n_2i+1) / sqrt 2, the real m = 0 column
// (i <= lmax) gets (n_2i, 0) -- the convention of phas.lib_phas.get_sim (phas.py:162-168); tag 1
__global__ __launch_bounds__(256) void k_alm_unit_phases(int lmax, int64_t nalm, double2 *out, uint64_t key)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nalm; i += stride) {
        const double2 g = normal_pair(key, (uint64_t)i, 1u);
        out[i] = i <= lmax ? make_double2(g.x, 0.0) : make_double2(g.x * 0.70710678118654752440, g.y * 0.70710678118654752440);
    }
}

// out[k] = f[k][0] a[k][0] (+ f[k][1] a[k][1]), k < nout <= 2: the Wiener-filtered legs of the estimators (X^WF = C^XX Xb + C^TE Yb,
// qest.py:566-638) for both components of a spin transform in ONE launch.  Rounded exactly as k_almxfl followed by k_almxfl_add.
struct LinComb { const double2 *a[2][2]; const double *f[2][2]; double2 *out[2]; int nterm[2]; };
__global__ void k_alm_lincomb(int lmax, LinComb C)
{
    const int m = blockIdx.y, k = blockIdx.z;
    const int64_t base = (int64_t)m * (2 * lmax + 1 - m) / 2;
    const double2 *__restrict__ a0 = C.a[k][0], *__restrict__ a1 = C.a[k][1];
    const double *__restrict__ f0 = C.f[k][0], *__restrict__ f1 = C.f[k][1];
    double2 *__restrict__ out = C.out[k];
    const bool two = C.nterm[k] == 2;
    for (int l = m + blockIdx.x * blockDim.x + threadIdx.x; l <= lmax; l += gridDim.x * blockDim.x) {
        const double2 x = a0[base + l];
        const double g = f0[l];
        double2 v = make_double2(x.x * g, x.y * g);
        if (two) {
            const double2 y = a1[base + l];
            const double h = f1[l];
            v = make_double2(fma(h, y.x, v.x), fma(h, y.y, v.y));
        }
        out[base + l] = v;
    }
}

static inline int nblocks(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

void launch_phase_pack(const DevPlan &P, int ncomp, double *phase, double *buf, int pair0, int pstride, int mg0, int mgstride, bool unpack, hipStream_t st)
{
    const int nsel = P.npairs > pair0 ? (P.npairs - pair0 + pstride - 1) / pstride : 0;
    const int nmg_all = (P.mmax + 4) / 4;
    const int nmsel = nmg_all > mg0 ? (nmg_all - mg0 + mgstride - 1) / mgstride : 0;
    const int64_t n = (int64_t)nsel * ncomp * nmsel * 8;
    if (n == 0) return;
    hipLaunchKernelGGL(k_phase_pack, dim3(nblocks(n)), dim3(256), 0, st, P, ncomp, phase, buf, pair0, pstride, mg0, mgstride, nsel, nmsel, unpack ? 1 : 0);
}
int64_t map_pack_doubles(const DevPlan &P, int pair0, int pstride)
{
    const int nsel = P.npairs > pair0 ? (P.npairs - pair0 + pstride - 1) / pstride : 0;
    if (nsel == 0) return 0;
    int64_t n = ring_pack_offset(P.nside, pair0, pstride, nsel);
    if (pair0 + (nsel - 1) * pstride == P.npairs - 1) n -= 4 * (int64_t)P.nside;  // the equator has no mirror ring
    return n;
}
void launch_map_pack_rings(const DevPlan &P, int ncomp, double *map, double *buf, int pair0, int pstride, bool unpack, hipStream_t st)
{
    const int nsel = P.npairs > pair0 ? (P.npairs - pair0 + pstride - 1) / pstride : 0;
    if (nsel == 0) return;
    hipLaunchKernelGGL(k_map_pack_rings, dim3(nsel, ncomp), dim3(256), 0, st, P, map, buf, pair0, pstride, map_pack_doubles(P, pair0, pstride), unpack ? 1 : 0);
}
void launch_alm_keep_mgroups(int lmax, double *alm, int mg0, int mgstride, int nb, hipStream_t st)
{
    hipLaunchKernelGGL(k_alm_keep_mgroups, dim3(4, lmax + 1, nb), dim3(256), 0, st, lmax, reinterpret_cast<double2 *>(alm), mg0, mgstride);
}
void launch_almxfl(int lmax, const double *in, const double *fl, int nfl, double *out, hipStream_t st, int nb)
{
    hipLaunchKernelGGL(k_almxfl, dim3(4, lmax + 1, nb), dim3(256), 0, st, lmax, reinterpret_cast<const double2 *>(in), fl, nfl,
                       reinterpret_cast<double2 *>(out));
}
void launch_alm_copy(int lmax_in, const double *in, int lmax_out, double *out, hipStream_t st, int nb)
{
    hipLaunchKernelGGL(k_alm_copy, dim3(4, lmax_out + 1, nb), dim3(256), 0, st, lmax_in, reinterpret_cast<const double2 *>(in),
                       lmax_out, reinterpret_cast<double2 *>(out));
}
void launch_alm_dot(int lmax, int lmin, const double *a, const double *b, int accumulate, double *parts, hipStream_t st, int nb)
{
    hipLaunchKernelGGL(k_alm_dot_parts, dim3(kDotParts, nb), dim3(kDotThreads), 0, st, lmax, lmin, reinterpret_cast<const double2 *>(a),
                       reinterpret_cast<const double2 *>(b), accumulate, parts);
}
void launch_axpy_dev(int64_t n, const double *num, const double *den, double sign, const double *x, double *y, hipStream_t st, int nb)
{
    hipLaunchKernelGGL(k_axpy_dev, dim3(nblocks((n + 1) / 2), nb), dim3(256), 0, st, n, num, den, sign, x, y);
}
void launch_cg_fused(int nf, const int *lmax, int lmin, const double *const *a, const double *const *b1, const double *const *b2, double *parts1,
                     double *parts2, const double *den, double *const *y1, const double *const *x1, double sign1, double *const *y2,
                     const double *const *x2, double sign2, unsigned *bar, hipStream_t st, int nbatch, const double *active)
{
    // bar null: two launches (products, then updates); else one launch with a grid barrier in between
    CgFused f = {};
    f.nf = nf; f.lmin = lmin;
    int64_t nmax = 0;
    for (int k = 0; k < nf; ++k) {
        f.lmax[k] = lmax[k];
        f.a[k] = reinterpret_cast<const double2 *>(a[k]);
        f.b1[k] = reinterpret_cast<const double2 *>(b1[k]);
        f.b2[k] = b2 ? reinterpret_cast<const double2 *>(b2[k]) : nullptr;
        f.y1[k] = reinterpret_cast<double2 *>(y1[k]);
        f.x1[k] = reinterpret_cast<const double2 *>(x1[k]);
        f.y2[k] = y2 ? reinterpret_cast<double2 *>(y2[k]) : nullptr;
        f.x2[k] = x2 ? reinterpret_cast<const double2 *>(x2[k]) : nullptr;
        const int64_t nalm = (int64_t)(lmax[k] + 1) * (lmax[k] + 2) / 2;
        if (nalm > nmax) nmax = nalm;
    }
    // the scalar products always use kDotParts workgroups; long vectors get more for the updates (never more than one per CU)
    int nb = (int)((nmax + 4 * kDotThreads - 1) / (4 * kDotThreads));
    if (nb < kDotParts) nb = kDotParts;
    if (nb > kCgBlocks) nb = kCgBlocks;
    if (bar) {
        hipLaunchKernelGGL(k_cg_fused<0>, dim3(nb), dim3(kDotThreads), 0, st, f, parts1, parts2, den, sign1, sign2, bar, active);  // (one entry only)
    } else {
        hipLaunchKernelGGL(k_cg_fused<1>, dim3(kDotParts, nbatch), dim3(kDotThreads), 0, st, f, parts1, parts2, den, sign1, sign2, bar, active);
        const int nb2 = (int)((nmax + kDotThreads - 1) / kDotThreads);  // updates: one entry per thread up to 1024 workgroups
        hipLaunchKernelGGL(k_cg_fused<2>, dim3(nb2 < 1 ? 1 : (nb2 > 1024 ? 1024 : nb2), nbatch), dim3(kDotThreads), 0, st, f, parts1, parts2, den, sign1,
                           sign2, bar, active);
    }
}
void launch_cg_axpy_pre(int nf, const int *lmax, int npre, const double *pre1, const double *pre2, const double *den, double *parts1, double *parts2,
                        double *const *y1, const double *const *x1, double sign1, double *const *y2, const double *const *x2, double sign2, hipStream_t st,
                        int nbatch, const double *active, int y1_assign)
{
    CgFused f = {};
    f.nf = nf;
    int64_t nmax = 0;
    for (int k = 0; k < nf; ++k) {
        f.lmax[k] = lmax[k];
        f.y1[k] = reinterpret_cast<double2 *>(y1[k]);
        f.x1[k] = reinterpret_cast<const double2 *>(x1[k]);
        f.y2[k] = y2 ? reinterpret_cast<double2 *>(y2[k]) : nullptr;
        f.x2[k] = x2 ? reinterpret_cast<const double2 *>(x2[k]) : nullptr;
        const int64_t nalm = (int64_t)(lmax[k] + 1) * (lmax[k] + 2) / 2;
        if (nalm > nmax) nmax = nalm;
    }
    const int nb2 = (int)((nmax + kDotThreads - 1) / kDotThreads);  // as the update launch of launch_cg_fused
    const int cap = npre > 1024 ? 256 : 1024;  // every workgroup adds all npre partial sums: long lists (the fine grids) go to one workgroup per CU
    hipLaunchKernelGGL(k_cg_axpy_pre, dim3(nb2 < 1 ? 1 : (nb2 > cap ? cap : nb2), nbatch), dim3(kDotThreads), 0, st, f, npre, pre1, pre2, den, parts1,
                       parts2, sign1, sign2, active, y1_assign);
}
// t_apply (optional): the projection is subtracted from this vector instead of t, which is then only read (n_inv null) -- the
// rank-nmodes update y -= rm^t (pm x) of launch_lowrank_update
// phase: 0 both launches on st; 1 the coefficient pass only; 2 the subtraction only (the two halves of a low-rank update whose
// coefficient pass runs beside the transforms on another stream, pl_cg_fwd_tt_lr_b)
void launch_template_project(int64_t n, int nmodes, double *t, const double *n_inv, const double *pm, const double *rm, double *parts, hipStream_t st,
                             int nb, double *t_apply, int phase)
{
    double *ta = t_apply ? t_apply : t;
    const bool do_c = phase != 2, do_a = phase != 1;
    if (nb > 1 && nmodes <= kFuseModesB) {  // block vectors: the template rows read once per chunk of kProjChunk maps
        const int nchunk = (nb + kProjChunk - 1) / kProjChunk;
        if (n >= (int64_t)kProjParts * 4096) {
            if (do_c) hipLaunchKernelGGL(k_tproj_coeffs_b<1024>, dim3(kProjParts, nchunk), dim3(1024), 0, st, n, nmodes, nb, t, n_inv, pm, parts);
            if (do_a) hipLaunchKernelGGL(k_tproj_apply_b, dim3(nblocks(n), nchunk), dim3(256), 0, st, n, nmodes, kProjParts, nb, ta, rm, parts);
        } else {
            int nparts = (int)((n + 511) / 512);  // two entries per thread: the coefficient pass of a coarse grid is a latency chain of its loads
            if (nparts < 1) nparts = 1;
            if (nparts > kProjParts) nparts = kProjParts;
            if (do_c) hipLaunchKernelGGL(k_tproj_coeffs_b<256>, dim3(nparts, nchunk), dim3(256), 0, st, n, nmodes, nb, t, n_inv, pm, parts);
            if (do_a) hipLaunchKernelGGL(k_tproj_apply_b, dim3(nblocks(n), nchunk), dim3(256), 0, st, n, nmodes, nparts, nb, ta, rm, parts);
        }
        return;
    }
    if (n >= (int64_t)kProjParts * 4096) {  // fine grids: 256 workgroups of 1024 threads
        if (do_c) hipLaunchKernelGGL(k_tproj_coeffs<1024>, dim3(kProjParts, nb), dim3(1024), 0, st, n, nmodes, t, n_inv, pm, parts);
        if (do_a) hipLaunchKernelGGL(k_tproj_apply, dim3(nblocks(n), nb), dim3(256), 0, st, n, nmodes, kProjParts, ta, rm, parts);
    } else {  // coarse grids: workgroups of 256 threads
        int nparts = (int)((n + 511) / 512);  // two entries per thread: the coefficient pass of a coarse grid is a latency chain of its loads
        if (nparts < 1) nparts = 1;
        if (nparts > kProjParts) nparts = kProjParts;
        if (do_c) hipLaunchKernelGGL(k_tproj_coeffs<256>, dim3(nparts, nb), dim3(256), 0, st, n, nmodes, t, n_inv, pm, parts);
        if (do_a) hipLaunchKernelGGL(k_tproj_apply, dim3(nblocks(n), nb), dim3(256), 0, st, n, nmodes, nparts, ta, rm, parts);
    }
}
// layout of the partial sums k_tproj_coeffs leaves for a single vector of n doubles: (number of partial sums per mode, stride between modes)
void tproj_parts_layout(int64_t n, int *nparts, int *pstride)
{
    int np = kProjParts;
    if (n < (int64_t)kProjParts * 4096) { np = (int)((n + 511) / 512); if (np < 1) np = 1; if (np > kProjParts) np = kProjParts; }
    *nparts = np; *pstride = kProjParts;
}
int64_t tproj_parts_bstride() { return (int64_t)kProjMaxModes * kProjParts; }  // between the partial sums of consecutive batch entries
// scratch: nb x 4 x npairs partial sums followed by nb x 4 coefficients
void launch_template_project_md(const DevPlan &P, int nb, double *t, const double *n_inv, int weighted, const double *pinv, double *scratch,
                                hipStream_t st)
{
    const int nchunk = (nb + kProjChunk - 1) / kProjChunk;
    double *d = scratch + (int64_t)nb * 4 * P.npairs;
    hipLaunchKernelGGL(k_tproj_md_coeffs, dim3(P.npairs, nchunk), dim3(256), 0, st, P, nb, t, n_inv, weighted, scratch);
    hipLaunchKernelGGL(k_tproj_md_reduce, dim3(nb), dim3(256), 0, st, P.npairs, scratch, pinv, d);
    hipLaunchKernelGGL(k_tproj_md_apply, dim3(P.npairs, nchunk), dim3(256), 0, st, P, nb, t, n_inv, d);
}
// up to 8 device addresses, passed BY VALUE in the kernel arguments, written to a table in device memory (pl_store_addresses): stream-ordered like any
// kernel, no host buffer whose lifetime the caller would have to manage
struct AddrList { unsigned long long a[8]; };
__global__ void k_store_addresses(AddrList v, int n, unsigned long long *__restrict__ dst)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = v.a[threadIdx.x];
}
void launch_store_addresses(int n, const unsigned long long *vals, unsigned long long *dst, hipStream_t st)
{
    AddrList v = {};
    for (int i = 0; i < n && i < 8; ++i) v.a[i] = vals[i];
    hipLaunchKernelGGL(k_store_addresses, dim3(1), dim3(64), 0, st, v, n, dst);
}
void launch_copy_slim(const double *src, double *dst, int64_t ndoubles, int nblocks, hipStream_t st)
{
    const int64_t n2 = ndoubles / 2;
    hipLaunchKernelGGL(k_copy_slim, dim3(nblocks < 1 ? 1 : nblocks), dim3(256), 0, st, reinterpret_cast<const double2 *>(src),
                       reinterpret_cast<double2 *>(dst), n2, src + 2 * n2, dst + 2 * n2, (int)(ndoubles - 2 * n2));
}
void launch_gemv(int nrows, int ncols, int64_t lda, const double *A, const double *x, double *y, hipStream_t st)
{
    const bool vec = (lda & 1) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec) hipLaunchKernelGGL(k_gemv<2>, dim3((nrows + 3) / 4), dim3(256), 0, st, nrows, ncols, lda, A, x, y);
    else hipLaunchKernelGGL(k_gemv<1>, dim3((nrows + 3) / 4), dim3(256), 0, st, nrows, ncols, lda, A, x, y);
}
int gemv_split_dot_count(int nf, int lmax_lo, int lmax_hi) { return (nf * (lmax_lo + 1) * (lmax_lo + 2) + 3) / 4 + nf * 4 * (lmax_hi + 1); }
void launch_gemv_split(int nf, int64_t lda, const double *A, const double *const *hi, const int *map, int lmax_lo, int lmax_hi, const double *const *fl_hi,
                       double *const *out, hipStream_t st, const double *const *dot_q, int dot_lmin, double *dot_pre)
{
    SplitDot D;
    if (dot_pre) { for (int f = 0; f < nf; ++f) D.q[f] = reinterpret_cast<const double2 *>(dot_q[f]); D.pre = dot_pre; D.lmin = dot_lmin; }
    const int nrows = nf * (lmax_lo + 1) * (lmax_lo + 2);
    const int ngemv = (nrows + 3) / 4;
    SplitFields F = {};
    for (int f = 0; f < nf; ++f) { F.hi[f] = reinterpret_cast<const double2 *>(hi[f]); F.out[f] = reinterpret_cast<double2 *>(out[f]); F.fl[f] = fl_hi[f]; }
    const dim3 grid(ngemv + nf * 4 * (lmax_hi + 1));
    if (nf == 1) hipLaunchKernelGGL(k_gemv_split<1>, grid, dim3(256), 0, st, nrows, lda, A, F, map, lmax_lo, lmax_hi, ngemv, D);
    else hipLaunchKernelGGL(k_gemv_split<2>, grid, dim3(256), 0, st, nrows, lda, A, F, map, lmax_lo, lmax_hi, ngemv, D);
}
// nb right-hand sides: x [nb][ncols] -> y [nb][nrows]
void launch_gemv_nb(int nrows, int ncols, int64_t lda, const double *A, int nb, const double *x, double *y, hipStream_t st)
{
    const bool vec = (lda & 1) == 0 && (ncols & 1) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (nb == 1 || !vec) {  // (odd sizes / unaligned: one plain mat-vec per right-hand side)
        for (int b = 0; b < nb; ++b) launch_gemv(nrows, ncols, lda, A, x + (int64_t)b * ncols, y + (int64_t)b * nrows, st);
        return;
    }
    for (int b0 = 0; b0 < nb; b0 += 8) {
        const int n = nb - b0 < 8 ? nb - b0 : 8;
        const double *xb = x + (int64_t)b0 * ncols;
        double *yb = y + (int64_t)b0 * nrows;
        if (n <= 2) hipLaunchKernelGGL(k_gemv_nb<2>, dim3((nrows + 3) / 4), dim3(256), 0, st, nrows, ncols, lda, A, n, xb, yb);
        else if (n <= 4) hipLaunchKernelGGL(k_gemv_nb<4>, dim3((nrows + 3) / 4), dim3(256), 0, st, nrows, ncols, lda, A, n, xb, yb);
        else hipLaunchKernelGGL(k_gemv_nb<8>, dim3((nrows + 3) / 4), dim3(256), 0, st, nrows, ncols, lda, A, n, xb, yb);
    }
}
int alm_splice_dot_count(int lmax_hi) { return 4 * (lmax_hi + 1); }
void launch_alm_splice(int lmax_lo, const double *lo, int lmax_hi, const double *hi, int lsplit, double *out, hipStream_t st,
                       const double *fl_hi, int nb, const double *dot_q, int dot_lmin, double *dot_pre)
{
    hipLaunchKernelGGL(k_alm_splice, dim3(4, lmax_hi + 1, nb), dim3(256), 0, st, lmax_lo, reinterpret_cast<const double2 *>(lo), lmax_hi,
                       reinterpret_cast<const double2 *>(hi), lsplit, reinterpret_cast<double2 *>(out), fl_hi,
                       reinterpret_cast<const double2 *>(dot_q), dot_lmin, dot_pre);
}
void launch_almxfl_add(int lmax, const double *a, const double *b, const double *fl, int nfl, double *out, hipStream_t st, int nb)
{
    hipLaunchKernelGGL(k_almxfl_add, dim3(4, lmax + 1, nb), dim3(256), 0, st, lmax, reinterpret_cast<const double2 *>(a),
                       reinterpret_cast<const double2 *>(b), fl, nfl, reinterpret_cast<double2 *>(out));
}
void launch_alm2cl(int lmax, const double *a, const double *b, double *cl, hipStream_t st)
{
    hipLaunchKernelGGL(k_alm2cl, dim3(lmax + 1), dim3(256), 0, st, lmax, reinterpret_cast<const double2 *>(a),
                       reinterpret_cast<const double2 *>(b), cl);
}
void launch_axpy(int64_t n, double a, const double *x, const double *y, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_axpy, dim3(nblocks(n)), dim3(256), 0, st, n, a, x, y, out);
}
void launch_map_mul(int64_t n, const double *a, const double *b, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_map_mul, dim3(nblocks(n)), dim3(256), 0, st, n, a, b, out);
}
void launch_map_qu_weight(int64_t n, double *q, double *u, const double *nqq, const double *nqu, const double *nuu, hipStream_t st, int nb,
                          int64_t bstride)
{
    hipLaunchKernelGGL(k_map_qu_weight, dim3(nblocks(n), nb), dim3(256), 0, st, n, q, u, nqq, nqu, nuu, bstride);
}
void launch_map_cmul(int64_t n, const double *ar, const double *ai, double s1, const double *br, const double *bi, double s2,
                     double sign, double *outr, double *outi, int accumulate, hipStream_t st)
{
    hipLaunchKernelGGL(k_map_cmul, dim3(nblocks(n)), dim3(256), 0, st, n, ar, ai, s1, br, bi, s2, sign, outr, outi, accumulate);
}
void launch_qe_lens_product(int64_t n, const double *tmap, const double *gt, const double *ct, const double *rep, const double *imp,
                            const double *g3, const double *c3, const double *g1, const double *c1, double *outr, double *outi, hipStream_t st)
{
    hipLaunchKernelGGL(k_qe_lens_product, dim3(nblocks(n)), dim3(256), 0, st, n, tmap, gt, ct, rep, imp, g3, c3, g1, c1, outr, outi);
}
void launch_fma_peak(int mode, int iters, double *out, int nblk, hipStream_t st)
{
    const double xs = 1.0 + 1e-9, ys = 1e-12;
    if (mode == 1) hipLaunchKernelGGL(k_fma_peak<1>, dim3(nblk), dim3(256), 0, st, iters, xs, ys, out);
    else if (mode == 2) hipLaunchKernelGGL(k_fma_peak<2>, dim3(nblk), dim3(256), 0, st, iters, xs, ys, out);
    else hipLaunchKernelGGL(k_fma_peak<0>, dim3(nblk), dim3(256), 0, st, iters, xs, ys, out);
}

void launch_map_add_normal(int64_t n, const double *in, double *out, double sigma, uint64_t key, hipStream_t st)
{
    hipLaunchKernelGGL(k_map_add_normal, dim3(nblocks((n + 1) / 2)), dim3(256), 0, st, n, in, out, sigma, key);
}
void launch_alm_unit_phases(int lmax, double *out, uint64_t key, hipStream_t st)
{
    const int64_t nalm = (int64_t)(lmax + 1) * (lmax + 2) / 2;
    hipLaunchKernelGGL(k_alm_unit_phases, dim3(nblocks(nalm)), dim3(256), 0, st, lmax, nalm, reinterpret_cast<double2 *>(out), key);
}

void launch_alm_lincomb(int lmax, int nout, const int *nterm, const double *const *alm, const double *const *fl, double *const *out, hipStream_t st)
{
    LinComb C{};
    for (int k = 0; k < nout; ++k) {
        C.nterm[k] = nterm[k];
        C.out[k] = reinterpret_cast<double2 *>(out[k]);
        for (int t = 0; t < nterm[k]; ++t) { C.a[k][t] = reinterpret_cast<const double2 *>(alm[2 * k + t]); C.f[k][t] = fl[2 * k + t]; }
    }
    hipLaunchKernelGGL(k_alm_lincomb, dim3(4, lmax + 1, nout), dim3(256), 0, st, lmax, C);
}

}  // namespace plshts
