// Fourier stage of the HEALPix SHTs for gfx950: per-ring Fourier coefficients F_m(ring) <-> RING pixels.
//
// One workgroup per ring pair (north ring + mirror south ring, identical nphi and phi0) and component.  The two real
// rings are transformed by ONE complex DFT of length n = nphi = 4 q (z = north + i south), split 4 x q: radix-4 on the
// pixel side (pixel j = j1 + q j2, bin k = 4 k1 + k2) and four sub-DFTs of length q.
//   * Register-resident kernels (k_phase2map_fast / k_map2phase_fast, second half of this file) serve every ring without
//     aliasing whose transform size N is 256 ... 4096: Stockham autosort radix-8 passes in registers, one swizzled N-point
//     LDS exchange buffer; q a power of two runs directly (N = q), any other q as a band-limited Bluestein convolution of
//     size N >= q + 2 K + 1 (K = in-band sub-DFT bins) with per-q filter spectra precomputed at plan creation; a ring that
//     would need twice the class size runs as two half-size convolutions sharing one transform (SPLIT: 3 transforms of size
//     N for 2 of size 2 N).  The synthesis kernels can multiply the pixels by a weight map on the way out (WGT: the
//     inverse-noise weighting of the CG operators).
//   * The generic kernel (k_phase2map / k_map2phase, first half) keeps a whole sub-DFT in LDS (radix-8/4/2 DIF forward /
//     DIT inverse, Bluestein size M >= 2 q - 1 with digit-reversed filter spectrum): short polar rings, aliased rings
//     (lmax >= 2 nside, coarse multigrid levels), and the reference route the register kernels are tested against.  Its
//     workgroups are sized by the longest ring of its list; when the LDS allows, the four sub-DFTs of a ring are transformed
//     side by side (B4).  It also carries the CG's pixel-space operator with template marginalisation (NinvProj).
// Synthesis gathers the spectrum bins straight from the phase array and finishes with the radix-4 butterfly while
// writing pixels; analysis is the exact transpose.
//
// Bound: HBM (8 npix + 32 (mmax + 1) npairs bytes per component) with an FP64 add/mul + LDS-exchange floor close behind.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "device_plan.h"
#include "plshts_internal.h"
#include "ringfft.h"

namespace plshts {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a conj(b)
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
// a * i^r
__device__ __forceinline__ double2 crot(double2 a, int r)
{
    r &= 3;
    double2 o;
    o.x = (r == 0) ? a.x : (r == 1) ? -a.y : (r == 2) ? -a.x : a.y;
    o.y = (r == 0) ? a.y : (r == 1) ? a.x : (r == 2) ? -a.y : -a.x;
    return o;
}
__device__ __forceinline__ double2 cispi(double t)  // e^{i pi t}
{
    double s, c;
    sincospi(t, &s, &c);
    return make_double2(c, s);
}

// ---- in-LDS power-of-two FFT, radix 8 / 4 / 2 passes done in registers ------------------------------------
// Forward = decimation in frequency (natural order in, digit-reversed order out); inverse = the exact reverse
// (decimation in time, digit-reversed in, natural out, unnormalised).  One __syncthreads per pass; a size-4096
// transform is 4 passes of radix 8 instead of 12 radix-2 sweeps.
constexpr double kSqrtHalf = 0.70710678118654752440;

template <bool FWD>
__device__ __forceinline__ double2 mul_i(double2 a)  // forward: a * (-i); inverse: a * (+i)
{
    return FWD ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x);
}

template <int R, bool FWD>
__device__ __forceinline__ void dft_small(double2 (&x)[R])
{
    if constexpr (R == 2) {
        const double2 a = x[0], b = x[1];
        x[0] = cadd(a, b); x[1] = csub(a, b);
    } else if constexpr (R == 4) {
        const double2 s02 = cadd(x[0], x[2]), d02 = csub(x[0], x[2]);
        const double2 s13 = cadd(x[1], x[3]), d13 = mul_i<FWD>(csub(x[1], x[3]));
        x[0] = cadd(s02, s13); x[2] = csub(s02, s13);
        x[1] = cadd(d02, d13); x[3] = csub(d02, d13);
    } else {  // R == 8
        double2 e[4] = {x[0], x[2], x[4], x[6]}, o[4] = {x[1], x[3], x[5], x[7]};
        dft_small<4, FWD>(e);
        dft_small<4, FWD>(o);
        // o_k * w8^k, w8 = e^{-+ 2 pi i / 8}
        const double2 o1 = FWD ? make_double2((o[1].x + o[1].y) * kSqrtHalf, (o[1].y - o[1].x) * kSqrtHalf)
                               : make_double2((o[1].x - o[1].y) * kSqrtHalf, (o[1].y + o[1].x) * kSqrtHalf);
        const double2 o2 = mul_i<FWD>(o[2]);
        const double2 o3 = FWD ? make_double2((o[3].y - o[3].x) * kSqrtHalf, -(o[3].x + o[3].y) * kSqrtHalf)
                               : make_double2(-(o[3].x + o[3].y) * kSqrtHalf, (o[3].x - o[3].y) * kSqrtHalf);
        x[0] = cadd(e[0], o[0]); x[4] = csub(e[0], o[0]);
        x[1] = cadd(e[1], o1);   x[5] = csub(e[1], o1);
        x[2] = cadd(e[2], o2);   x[6] = csub(e[2], o2);
        x[3] = cadd(e[3], o3);   x[7] = csub(e[3], o3);
    }
}

// Pass structure for size M = 2^k: radix 8 while the remaining block size allows, then one radix-4 or radix-2
// pass (block sizes L: M, M/8, M/64, ..., tail).  Twiddles W_L^{pos}, W_L^{2 pos}, W_L^{4 pos} of every pass are
// staged once per workgroup in LDS (`twl`, pass with the smallest L first): a global-memory twiddle fetch
// (~500 cycles) inside every butterfly was the long pole of the passes.  twl == nullptr uses the global table.
__device__ __forceinline__ int fft_tail_radix(int M)  // radix of the smallest-L pass
{
    const int k = 31 - __clz(M);
    return (k % 3 == 0) ? 8 : (k % 3 == 2 ? 4 : 2);
}
__device__ __forceinline__ int fft_nk(int r) { return r == 8 ? 3 : r == 4 ? 2 : 1; }

template <int NT>
__device__ void fft_build_twl(double2 *twl, int M, const double2 *__restrict__ tw, int Mtw)
{
    int off = 0;
    int r = fft_tail_radix(M);
    for (int L = r; L <= M; L *= 8) {  // after the tail pass every pass is radix 8
        const int s = L / r, nk = fft_nk(r), tstep = Mtw / L;
        for (int i = threadIdx.x; i < nk * s; i += NT) {
            const int kk = i / s, pos = i - kk * s;
            twl[off + i] = tw[(pos << kk) * tstep];
        }
        off += nk * s;
        r = 8;
    }
    __syncthreads();
}

// one pass over blocks of size L (stride s = L / R); twp: this pass's LDS twiddles or nullptr
// `batch` transforms of size M at a, a + bstride, ... are swept together (the four sub-DFTs of a short ring: a single
// size-256 transform has only 32 radix-8 butterflies per pass for 256 threads)
template <int NT, int R, bool FWD>
__device__ __forceinline__ void fft_pass(double2 *a, int M, int L, const double2 *twp, const double2 *__restrict__ tw, int Mtw,
                                         int batch = 1, int bstride = 0)
{
    const int s = L / R;
    const int ls = 31 - __clz(s);
    const int lmr = 31 - __clz(M / R);
    const int tstep = Mtw / L;
    for (int tb = threadIdx.x; tb < batch * (M / R); tb += NT) {
        const int t = tb & (M / R - 1);
        const int blk = t >> ls, pos = t & (s - 1);
        double2 *p = a + (tb >> lmr) * bstride + blk * L + pos;
        double2 x[R];
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = p[j * s];
        double2 w[R];
        w[0] = make_double2(1., 0.);
        if (twp) {
            w[1] = twp[pos];
            if constexpr (R >= 4) { w[2] = twp[s + pos]; w[3] = cmul(w[1], w[2]); }
            if constexpr (R == 8) { w[4] = twp[2 * s + pos]; w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]); }
        } else {
            w[1] = tw[pos * tstep];
            if constexpr (R >= 4) { w[2] = tw[2 * pos * tstep]; w[3] = cmul(w[1], w[2]); }
            if constexpr (R == 8) { w[4] = tw[4 * pos * tstep]; w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]); }
        }
        if constexpr (FWD) {
            dft_small<R, true>(x);
#pragma unroll
            for (int k = 1; k < R; ++k) x[k] = cmul(x[k], w[k]);
        } else {
#pragma unroll
            for (int k = 1; k < R; ++k) x[k] = cmulc(x[k], w[k]);
            dft_small<R, false>(x);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) p[j * s] = x[j];
    }
}

// position of natural index f in the digit-reversed layout produced by the forward transform
__device__ __forceinline__ int digit_reverse(int f, int M)
{
    const int rt = fft_tail_radix(M);
    int p = 0, s = M;
    while (s > rt) { s >>= 3; p += (f & 7) * s; f >>= 3; }  // radix-8 passes, largest L first
    return p + f;                                             // tail digit has stride 1
}

template <int NT>
__device__ void fft_dif_fwd(double2 *a, int M, const double2 *twl, const double2 *__restrict__ tw, int Mtw, int batch = 1, int bstride = 0)
{
    if (M < 2) { __syncthreads(); return; }
    const int rt = fft_tail_radix(M);
    // LDS twiddle offset of the largest pass = total - its own size; walk downwards
    int off = 0;
    { int r = rt; for (int L = rt; L <= M; L *= 8) { off += fft_nk(r) * (L / r); r = 8; } }
    for (int L = M; L > rt; L >>= 3) {
        off -= 3 * (L / 8);
        __syncthreads();
        fft_pass<NT, 8, true>(a, M, L, twl ? twl + off : nullptr, tw, Mtw, batch, bstride);
    }
    __syncthreads();
    if (rt == 8) fft_pass<NT, 8, true>(a, M, 8, twl, tw, Mtw, batch, bstride);
    else if (rt == 4) fft_pass<NT, 4, true>(a, M, 4, twl, tw, Mtw, batch, bstride);
    else fft_pass<NT, 2, true>(a, M, 2, twl, tw, Mtw, batch, bstride);
    __syncthreads();
}

template <int NT>
__device__ void fft_dit_inv(double2 *a, int M, const double2 *twl, const double2 *__restrict__ tw, int Mtw, int batch = 1, int bstride = 0)
{
    if (M < 2) { __syncthreads(); return; }
    const int rt = fft_tail_radix(M);
    __syncthreads();
    if (rt == 8) fft_pass<NT, 8, false>(a, M, 8, twl, tw, Mtw, batch, bstride);
    else if (rt == 4) fft_pass<NT, 4, false>(a, M, 4, twl, tw, Mtw, batch, bstride);
    else fft_pass<NT, 2, false>(a, M, 2, twl, tw, Mtw, batch, bstride);
    int off = fft_nk(rt) * 1;  // tail pass table: nk * (rt / rt) entries
    for (int L = rt * 8; L <= M; L *= 8) {
        __syncthreads();
        fft_pass<NT, 8, false>(a, M, L, twl ? twl + off : nullptr, tw, Mtw, batch, bstride);
        off += 3 * (L / 8);
    }
    __syncthreads();
}

// In: ws[0..q) filled (Bluestein: x_k w_k, zero padded to M by the caller; direct: x at bit-reversed positions).
// Out: ws[j] (times chirp[j] for Bluestein, applied by the caller through sub_value) = sum_k x_k e^{+2 pi i jk/q}.
template <int NT>
__device__ __forceinline__ void sub_dft_inverse(double2 *ws, int q, int M, const double2 *__restrict__ filt, const DevFFT &F,
                                                const double2 *twl, int batch = 1, int bstride = 0)
{
    if (M) {
        fft_dif_fwd<NT>(ws, M, twl, F.tw, F.Mtw, batch, bstride);
        const int lm = 31 - __clz(M);
        for (int t = threadIdx.x; t < batch * M; t += NT) {
            double2 *w = ws + (t >> lm) * bstride + (t & (M - 1));
            *w = cmul(*w, filt[t & (M - 1)]);
        }
        fft_dit_inv<NT>(ws, M, twl, F.tw, F.Mtw, batch, bstride);
    } else {
        fft_dit_inv<NT>(ws, q, twl, F.tw, F.Mtw, batch, bstride);
    }
}

// -----------------------------------------------------------------------------------------------------
// plan-time setup of the Bluestein tables for one q per workgroup
// -----------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void k_bluestein_setup(DevFFT F, const int *__restrict__ qlist, double2 *__restrict__ chirp_out,
                                                        double2 *__restrict__ filt_out)
{
    extern __shared__ double2 ws[];
    const int q = qlist[blockIdx.x];
    const int M = F.Mof[q];  // 0: this q is served by a register-resident class, only its chirp is needed
    double2 *chirp = chirp_out + F.woff[q];
    double2 *filt = filt_out + F.coff[q];
    for (int t = threadIdx.x; t < M; t += NT) ws[t] = make_double2(0., 0.);
    __syncthreads();
    for (int t = threadIdx.x; t < q; t += NT) {
        const long long t2 = ((long long)t * t) % (2LL * q);
        const double2 w = cispi((double)t2 / (double)q);  // e^{i pi t^2 / q}
        chirp[t] = w;
        if (M == 0) continue;
        const double2 c = cconj(w);
        ws[t] = c;
        if (t > 0) ws[M - t] = c;
    }
    if (M == 0) return;
    fft_dif_fwd<NT>(ws, M, nullptr, F.tw, F.Mtw);
    const double inv = 1.0 / M;
    for (int t = threadIdx.x; t < M; t += NT) filt[t] = make_double2(ws[t].x * inv, ws[t].y * inv);
}

// twiddle table e^{-2 pi i t / Mtw}, t < Mtw / 2
__global__ void k_twiddles(double2 *tw, int Mtw)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Mtw / 2) tw[t] = cispi(-2.0 * t / Mtw);
}

// -----------------------------------------------------------------------------------------------------
// generic kernels: one workgroup per ring pair and component, transforms in LDS
// -----------------------------------------------------------------------------------------------------
// Thread t of NT owns the pixels j = j1 + q j2 with j1 = t + NT qq (qq < QMAX, j2 < 4) of both rings of the pair -- in the synthesis
// as well as in the analysis, which is what lets k_ring_roundtrip hand them over in registers.
struct RingCtx {
    int n, q, M, ml;
    bool shifted;
    double inv_n;
    const double2 *chirp, *filt;
    double2 *twl;
};

template <int NT, bool B4>
__device__ __forceinline__ RingCtx ring_ctx(const DevPlan &P, const DevFFT &F, int ip, const int *__restrict__ mlim, double2 *ws)
{
    RingCtx c;
    c.n = P.nphi[ip]; c.q = c.n >> 2;
    c.M = F.Mof[c.q];
    c.chirp = F.chirp + F.woff[c.q];
    c.filt = F.filt + F.coff[c.q];
    c.ml = min(mlim[ip], P.mmax);
    c.shifted = P.phi0[ip] != 0.0;
    c.inv_n = 1.0 / c.n;
    c.twl = F.twl_cap ? ws + (B4 ? 4 : 1) * F.Lmax : nullptr;
    if (c.twl) fft_build_twl<NT>(c.twl, c.M ? c.M : c.q, F.tw, F.Mtw);
    return c;
}

// synthesis: phase rows ph[m][N re, N im, S re, S im] -> acc[j2][qq] = (north, south) values of the thread's pixels
// B4: the four sub-DFTs of the ring are transformed side by side (LDS for 4 F.Lmax points) -- the coarse grids of the CG
// multigrid, where one short transform leaves most of the workgroup idle and the chain of barriers is the cost.
template <int NT, int QMAX, bool B4>
__device__ __forceinline__ void ring_synth(const RingCtx &c, const DevFFT &F, const double *__restrict__ ph, double2 *ws, double2 (&acc)[4][QMAX],
                                           const double2 (&e1)[QMAX])
{
    constexpr int estride = 4;
    const int n = c.n, q = c.q, M = c.M, ml = c.ml;
    const bool shifted = c.shifted;
    const double inv_n = c.inv_n;
    const double2 *__restrict__ chirp = c.chirp;
#pragma unroll
    for (int qq = 0; qq < QMAX; ++qq) acc[0][qq] = acc[1][qq] = acc[2][qq] = acc[3][qq] = make_double2(0., 0.);
    // input bin k = 4 k1 + k2 of the ring transform: all orders aliased onto it
    auto fold = [&](int k1, int k2) -> double2 {
        const int k = 4 * k1 + k2;
        double zr = 0., zi = 0.;
        for (int m = k; m <= ml; m += n) {  // positive frequencies aliased onto bin k
            const double4 f = *reinterpret_cast<const double4 *>(ph + (int64_t)m * estride);
            double2 p = shifted ? cispi(m * inv_n) : make_double2(1., 0.);
            const double2 fn = cmul(make_double2(f.x, f.y), p), fs = cmul(make_double2(f.z, f.w), p);
            zr += fn.x - fs.y; zi += fn.y + fs.x;  // f_N + i f_S
        }
        for (int m = n - k; m <= ml; m += n) {  // negative frequencies -m = k (mod n)
            const double4 f = *reinterpret_cast<const double4 *>(ph + (int64_t)m * estride);
            double2 p = shifted ? cispi(m * inv_n) : make_double2(1., 0.);
            const double2 fn = cmul(make_double2(f.x, f.y), p), fs = cmul(make_double2(f.z, f.w), p);
            zr += fn.x + fs.y; zi += -fn.y + fs.x;  // conj(f_N) + i conj(f_S)
        }
        return make_double2(zr, zi);
    };
    // sub-DFT k2 is in w[0 .. q): twiddle and radix-4 butterfly into the four quarter rings
    auto scatter = [&](int k2, const double2 *w) {
#pragma unroll
        for (int qq = 0; qq < QMAX; ++qq) {
            const int j1 = threadIdx.x + NT * qq;
            if (j1 < q) {
                double2 y = w[j1];
                if (M) y = cmul(y, chirp[j1]);
                // twiddle e^{2 pi i j1 k2 / n}
                double2 tw = make_double2(1., 0.);
                if (k2 >= 1) tw = e1[qq];
                if (k2 >= 2) tw = cmul(tw, e1[qq]);
                if (k2 >= 3) tw = cmul(tw, e1[qq]);
                y = cmul(y, tw);
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) acc[j2][qq] = cadd(acc[j2][qq], crot(y, j2 * k2));
            }
        }
    };

    if constexpr (B4) {
        const int S = M ? M : q;  // a power of two either way (Bluestein size, or q itself on the direct route)
        const int lS = 31 - __clz(S);
        for (int idx = threadIdx.x; idx < 4 * S; idx += NT) {
            const int k2 = idx >> lS, k1 = idx & (S - 1);
            if (k1 >= q) { ws[idx] = make_double2(0., 0.); continue; }
            const double2 z = fold(k1, k2);
            if (M) ws[idx] = cmul(z, chirp[k1]);
            else ws[k2 * S + digit_reverse(k1, q)] = z;
        }
        sub_dft_inverse<NT>(ws, q, M, c.filt, F, c.twl, 4, S);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) scatter(k2, ws + k2 * S);
    } else {
        for (int k2 = 0; k2 < 4; ++k2) {
            if (M) for (int t = q + threadIdx.x; t < M; t += NT) ws[t] = make_double2(0., 0.);
            for (int k1 = threadIdx.x; k1 < q; k1 += NT) {
                const double2 z = fold(k1, k2);
                if (M) ws[k1] = cmul(z, chirp[k1]);
                else ws[digit_reverse(k1, q)] = z;
            }
            sub_dft_inverse<NT>(ws, q, M, c.filt, F, c.twl);
            scatter(k2, ws);
            __syncthreads();
        }
    }
}

// analysis: zc[j2][qq] = (north, -south) values of the thread's pixels (conj(z_j), j = j1 + q j2) -> phase rows, uniform quadrature
// weights 4 pi / npix (wgt includes the 1/2 of the N/S split)
template <int NT, int QMAX, bool B4>
__device__ __forceinline__ void ring_anal(const RingCtx &c, const DevFFT &F, double *__restrict__ ph, double2 *ws, const double2 (&zc)[4][QMAX],
                                          const double2 (&e1)[QMAX], bool has_s, double wgt)
{
    constexpr int estride = 4;
    const int n = c.n, q = c.q, M = c.M, ml = c.ml;
    const bool shifted = c.shifted;
    const double inv_n = c.inv_n;
    const double2 *__restrict__ chirp = c.chirp;
    double2 hold[QMAX];
#pragma unroll
    for (int qq = 0; qq < QMAX; ++qq) hold[qq] = make_double2(0., 0.);

    // store F_N, F_S of order m given V_k = conj(Z_k) and V_{n-k}
    auto emit = [&](int m, double2 vk, double2 vm) {
        // F_N = (conj(V_k) + V_{n-k}) / 2,  F_S = (conj(V_k) - V_{n-k}) / (2i)
        const double2 a = cconj(vk);
        double2 fn = cadd(a, vm);
        const double2 d = csub(a, vm);
        double2 fs = make_double2(d.y, -d.x);  // d / i
        const double2 p = shifted ? cispi(-m * inv_n) : make_double2(1., 0.);
        fn = cmul(fn, p); fs = cmul(fs, p);
        double4 o;
        o.x = fn.x * wgt; o.y = fn.y * wgt;
        o.z = has_s ? fs.x * wgt : 0.0; o.w = has_s ? fs.y * wgt : 0.0;
        *reinterpret_cast<double4 *>(ph + (int64_t)m * estride) = o;
    };

    // input of sub-DFT k2 into w[0 .. S): radix-4 butterfly over the quarter rings, twiddle, (chirp), zero padding
    auto gather = [&](int k2, double2 *w) {
        if (M) for (int t = q + threadIdx.x; t < M; t += NT) w[t] = make_double2(0., 0.);
#pragma unroll
        for (int qq = 0; qq < QMAX; ++qq) {
            const int j1 = threadIdx.x + NT * qq;
            if (j1 < q) {
                double2 x = make_double2(0., 0.);
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) x = cadd(x, crot(zc[j2][qq], j2 * k2));
                double2 tw = make_double2(1., 0.);
                if (k2 >= 1) tw = e1[qq];
                if (k2 >= 2) tw = cmul(tw, e1[qq]);
                if (k2 >= 3) tw = cmul(tw, e1[qq]);
                x = cmul(x, tw);
                if (M) w[j1] = cmul(x, chirp[j1]);
                else w[digit_reverse(j1, q)] = x;
            }
        }
    };
    if constexpr (B4) {  // the four sub-DFTs side by side: V_{4 k1 + k2} = ws[k2 S + k1] (* chirp[k1])
        const int S = M ? M : q;
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) gather(k2, ws + k2 * S);
        sub_dft_inverse<NT>(ws, q, M, c.filt, F, c.twl, 4, S);
        auto val = [&](int k) {  // V_k, 0 <= k < n
            double2 v = ws[(k & 3) * S + (k >> 2)];
            if (M) v = cmul(v, chirp[k >> 2]);
            return v;
        };
        for (int m = threadIdx.x; m <= ml; m += NT) {
            const int k = m % n;
            emit(m, val(k), val((n - k) % n));
        }
        return;
    }
    for (int kk = 0; kk < 4; ++kk) {
        const int k2 = (kk == 0) ? 0 : (kk == 1) ? 2 : (kk == 2) ? 1 : 3;
        gather(k2, ws);
        sub_dft_inverse<NT>(ws, q, M, c.filt, F, c.twl);
        // now V_{4 k1 + k2} = ws[k1] (* chirp[k1])
        if (k2 == 0 || k2 == 2) {
            for (int m = k2 + 4 * threadIdx.x; m <= ml; m += 4 * NT) {
                const int k = m % n;
                const int km = (n - k) % n;
                double2 vk = ws[k >> 2], vm = ws[km >> 2];
                if (M) { vk = cmul(vk, chirp[k >> 2]); vm = cmul(vm, chirp[km >> 2]); }
                emit(m, vk, vm);
            }
        } else if (k2 == 1) {
#pragma unroll
            for (int qq = 0; qq < QMAX; ++qq) {
                const int k1 = threadIdx.x + NT * qq;
                if (k1 < q) {
                    double2 v = ws[k1];
                    if (M) v = cmul(v, chirp[k1]);
                    hold[qq] = v;
                }
            }
        } else {
#pragma unroll
            for (int qq = 0; qq < QMAX; ++qq) {
                const int k1 = threadIdx.x + NT * qq;
                if (k1 < q) {
                    const int k1p = q - 1 - k1;
                    const double2 a = hold[qq];      // V_{4 k1 + 1}
                    double2 b = ws[k1p];             // V_{4 k1p + 3} = V_{n - (4 k1 + 1)}
                    if (M) b = cmul(b, chirp[k1p]);
                    for (int m = 4 * k1 + 1; m <= ml; m += n) emit(m, a, b);
                    for (int m = 4 * k1p + 3; m <= ml; m += n) emit(m, b, a);
                }
            }
        }
        __syncthreads();
    }
}

// synthesis: phase -> pixels
template <int NT, int QMAX, bool B4>
__global__ __launch_bounds__(NT) void k_phase2map(DevPlan P, DevFFT F, const int *__restrict__ pairs, const int *__restrict__ mlim, int ncomp,
                                                  const double *__restrict__ phase, double *__restrict__ map, NinvProj W)
{
    extern __shared__ double2 ws[];
    const int ip = pairs[blockIdx.x];  // largest rings first
    const int comp = blockIdx.y;
    const RingCtx c = ring_ctx<NT, B4>(P, F, ip, mlim, ws);
    const int q = c.q;
    // phase array [ring pair][component][m][N re, N im, S re, S im]: a component's orders are contiguous
    const double *__restrict__ ph = phase + ((int64_t)ip * ncomp + comp) * P.mstride * 4;
    double2 acc[4][QMAX], e1[QMAX];
#pragma unroll
    for (int qq = 0; qq < QMAX; ++qq) e1[qq] = cispi(2.0 * (threadIdx.x + NT * qq) * c.inv_n);  // e^{2 pi i j1 / n}
    ring_synth<NT, QMAX, B4>(c, F, ph, ws, acc, e1);
    double *__restrict__ mp = map + (int64_t)comp * P.npix;
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    double cw[kFuseModes] = {0., 0., 0., 0.};  // NinvProj: this thread's share of the template coefficients
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
#pragma unroll
        for (int qq = 0; qq < QMAX; ++qq) {
            const int j1 = threadIdx.x + NT * qq;
            if (j1 < q) {
                const int j = j1 + q * j2;
                double xn = acc[j2][qq].x, xs = acc[j2][qq].y;
                if (W.n_inv) {
                    xn *= W.n_inv[on + j];
                    if (os >= 0) xs *= W.n_inv[os + j];
#pragma unroll
                    for (int k = 0; k < kFuseModes; ++k) {
                        if (k < W.nmodes) {
                            cw[k] = fma(W.pm[(int64_t)k * P.npix + on + j], xn, cw[k]);
                            if (os >= 0) cw[k] = fma(W.pm[(int64_t)k * P.npix + os + j], xs, cw[k]);
                        }
                    }
                }
                mp[on + j] = xn;
                if (os >= 0) mp[os + j] = xs;
            }
        }
    }
    if (W.n_inv && W.nmodes > 0) {  // wave-uniform: the ring pair's partial sums, reduced in a fixed order
        __syncthreads();  // the FFT workspace is free now
        double *red = reinterpret_cast<double *>(ws);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < kFuseModes; ++k) {
            if (k < W.nmodes) {
                double v = cw[k];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) red[k * (NT / 64) + wave] = v;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < W.nmodes) {
            double v = 0.0;
            for (int w = 0; w < NT / 64; ++w) v += red[threadIdx.x * (NT / 64) + w];
            W.parts[((int64_t)comp * W.nmodes + threadIdx.x) * W.nparts + blockIdx.x] = v;  // every component (batch entry) has its own sums
        }
    }
}

// analysis: pixels -> phase
template <int NT, int QMAX, bool B4>
__global__ __launch_bounds__(NT) void k_map2phase(DevPlan P, DevFFT F, const int *__restrict__ pairs, const int *__restrict__ mlim, int ncomp,
                                                  const double *__restrict__ map, double *__restrict__ phase, NinvProj W)
{
    extern __shared__ double2 ws[];
    __shared__ double cproj[kFuseModes];
    if (W.rm) {  // NinvProj: template coefficients = the partial sums of the synthesis side in ring-pair order
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int k = wave; k < W.nmodes; k += NT / 64) {
            double v = 0.0;
            for (int j = lane; j < W.nparts; j += 64) v += W.parts[((int64_t)blockIdx.y * W.nmodes + k) * W.nparts + j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) cproj[k] = v;
        }
        __syncthreads();
    }
    const int ip = pairs[blockIdx.x];
    const int comp = blockIdx.y;
    const RingCtx c = ring_ctx<NT, B4>(P, F, ip, mlim, ws);
    const int q = c.q;
    double *__restrict__ ph = phase + ((int64_t)ip * ncomp + comp) * P.mstride * 4;
    const double wgt = 0.5 * 4.0 * 3.14159265358979323846 / (double)P.npix;  // includes the 1/2 of the N/S split
    const double *__restrict__ mp = fft_input_map(F, P, map, comp);
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    const bool has_s = os >= 0;

    // conj(z_j) = north - i south, j = j1 + q j2
    double2 zc[4][QMAX], e1[QMAX];
#pragma unroll
    for (int qq = 0; qq < QMAX; ++qq) {
        const int j1 = threadIdx.x + NT * qq;
        e1[qq] = cispi(2.0 * j1 * c.inv_n);
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
            double2 v = make_double2(0., 0.);
            if (j1 < q) {
                const int j = j1 + q * j2;
                double xn = mp[on + j], xs = has_s ? mp[os + j] : 0.0;
                if (W.rm) {
                    for (int k = 0; k < W.nmodes; ++k) {
                        xn = fma(-W.rm[(int64_t)k * P.npix + on + j], cproj[k], xn);
                        if (has_s) xs = fma(-W.rm[(int64_t)k * P.npix + os + j], cproj[k], xs);
                    }
                }
                v.x = xn;
                v.y = -xs;
            }
            zc[j2][qq] = v;
        }
    }
    ring_anal<NT, QMAX, B4>(c, F, ph, ws, zc, e1, has_s, wgt);
}

// The pixel-space part of a CG operator whose inverse noise is diagonal, phase -> pixels -> n_inv x pixels -> phase, for the rings
// of the generic kernel in ONE launch (the coarse levels of the multigrid chains, where every ring is one): the pixel values
// stay in the registers of the threads that own them, no map is written.  In place on the phase array (a workgroup reads its
// rows before it writes them).  Same arithmetic as k_phase2map (weighted) followed by k_map2phase: results identical.
template <int NT, int QMAX, bool B4>
__global__ __launch_bounds__(NT) void k_ring_roundtrip(DevPlan P, DevFFT F, const int *__restrict__ pairs, const int *__restrict__ mlim, int ncomp,
                                                       double *phase, const double *__restrict__ n_inv)
{
    extern __shared__ double2 ws[];
    const int ip = pairs[blockIdx.x];
    const int comp = blockIdx.y;
    const RingCtx c = ring_ctx<NT, B4>(P, F, ip, mlim, ws);
    const int q = c.q;
    double *ph = phase + ((int64_t)ip * ncomp + comp) * P.mstride * 4;
    const double wgt = 0.5 * 4.0 * 3.14159265358979323846 / (double)P.npix;
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    const bool has_s = os >= 0;
    double2 px[4][QMAX], e1[QMAX];
#pragma unroll
    for (int qq = 0; qq < QMAX; ++qq) e1[qq] = cispi(2.0 * (threadIdx.x + NT * qq) * c.inv_n);
    ring_synth<NT, QMAX, B4>(c, F, ph, ws, px, e1);
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
#pragma unroll
        for (int qq = 0; qq < QMAX; ++qq) {
            const int j1 = threadIdx.x + NT * qq;
            double2 v = make_double2(0., 0.);
            if (j1 < q) {
                const int j = j1 + q * j2;
                v.x = px[j2][qq].x * n_inv[on + j];
                v.y = -(has_s ? px[j2][qq].y * n_inv[os + j] : 0.0);  // (the sign of the zero as k_map2phase forms it)
            }
            px[j2][qq] = v;
        }
    }
    __syncthreads();  // the transforms of the synthesis have been read out: the workspace takes the analysis side
    ring_anal<NT, QMAX, B4>(c, F, ph, ws, px, e1, has_s, wgt);
}

// =====================================================================================================
// Register-resident path for the long rings (sub-DFT or Bluestein convolution size N = 512 ... 4096)
// =====================================================================================================
// One workgroup of G = N / 8 threads per ring pair and component; every thread keeps 8 points of each of the four
// sub-DFTs in registers (4 x 8 complex).  A size-N transform is a Stockham autosort FFT: radix-8 butterflies in
// registers, data exchanged between passes through one N-point LDS buffer (natural order in, natural order out, so
// neither the direct nor the Bluestein route needs a reordering pass).  Thread tl always owns points tl + G j.
// LDS slots are XOR-swizzled so that both the stride-8^p stores (ds_write_b128: 8-lane groups, 32 banks) and the
// unit-stride loads (ds_read_b128: 16-lane groups, 64 banks) are conflict-free.
// Bluestein here uses the band limit of the ring: only the bins |m| <= mlim are non-zero, i.e. sub-DFT inputs
// c in [-K, K] (K = F.K2of[q]), so a convolution of size N >= q + 2 K + 1 is enough (instead of 2 q - 1).
__device__ __forceinline__ int swz(int i) { return i ^ ((i >> 3) & 7); }
// Launders a table index: the kernels below are fully unrolled and hipcc would otherwise load every table entry once
// at the top -- or hoist the loads of a later phase above the barriers of the current one -- and park the values in
// registers for the whole kernel (> 400 VGPRs, occupancy 1).  The memory clobber pins the load behind the preceding barrier.
__device__ __forceinline__ int fresh(int i) { asm volatile("" : "+v"(i)::"memory"); return i; }
__device__ __forceinline__ void phase_fence() { asm volatile("" ::: "memory"); }

template <int N>
struct Tw8 {
    static constexpr int P8 = (N >= 4096) ? 4 : (N >= 512) ? 3 : 2;   // radix-8 passes
    static constexpr int T = N >> (3 * P8);           // tail radix: 1, 2 or 4
    double2 w[P8 - 1];                                // pass p = 1 .. P8-1: W_L^k, L = 8^(p+1), k = tl mod 8^p
    double2 wt[T == 2 ? 4 : 2];                       // tail: W_N^(tl + G u) of its 8 / T butterflies
};

template <int N>
__device__ __forceinline__ void tw8_load(Tw8<N> &t, int tl, const double2 *__restrict__ tw, int Mtw)
{
    constexpr int G = N / 8;
#pragma unroll
    for (int p = 1; p < Tw8<N>::P8; ++p) {
        const int L = 1 << (3 * (p + 1));
        t.w[p - 1] = tw[(tl & ((1 << (3 * p)) - 1)) * (Mtw / L)];
    }
    if constexpr (Tw8<N>::T > 1) {
#pragma unroll
        for (int u = 0; u < 8 / Tw8<N>::T; ++u) t.wt[u] = tw[(tl + G * u) * (Mtw / N)];
    }
}

template <bool FWD>
__device__ __forceinline__ double2 twmul(double2 a, double2 w) { return FWD ? cmul(a, w) : cmulc(a, w); }

// x[j] = point tl + G j on entry and on return; FWD: e^{-2 pi i jk/N}, else e^{+2 pi i jk/N} (unnormalised).
// Powers of a twiddle are formed by multiplication (at most 3 products deep: ~4 ulp), which keeps one complex
// number per pass in registers instead of seven.
__device__ __forceinline__ void launder(double2 &w) { asm volatile("" : "+v"(w.x), "+v"(w.y)); }

template <int N, bool FWD>
__device__ __forceinline__ void fft8(double2 (&x)[8], double2 *lds, int tl_in, const Tw8<N> &tw_in)
{
    constexpr int G = N / 8, P8 = Tw8<N>::P8, T = Tw8<N>::T;
    // A kernel calls this 4 to 8 times with the same twiddles and thread index.  Left alone, hipcc computes the twiddle
    // powers and LDS addresses of all passes once and keeps them live across every call (CSE): +130 VGPRs.  Laundering the
    // inputs makes each call recompute them (a few dozen multiplies) and keeps the kernel at 2-3 waves per SIMD.
    Tw8<N> tw = tw_in;
    int tl = tl_in;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int p = 0; p < P8 - 1; ++p) launder(tw.w[p]);
#pragma unroll
    for (int u = 0; u < (T == 2 ? 4 : 2); ++u) launder(tw.wt[u]);
    // sched_barrier: hipcc otherwise hoists the table loads of every later phase of the fully unrolled kernel to the
    // top (hundreds of VGPRs in flight, occupancy 1)
    __builtin_amdgcn_sched_barrier(0);
    dft_small<8, FWD>(x);  // pass 0: sub-transform length 1, no twiddles
#pragma unroll
    for (int p = 1; p <= P8; ++p) {
        if (p == P8 && T == 1) break;
        // results of pass p - 1 (sub-transform length Ns = 8^(p-1)) go to (tl / Ns) * 8 Ns + (tl mod Ns) + r Ns
        const int Ns = 1 << (3 * (p - 1));
        const int base = ((tl >> (3 * (p - 1))) << (3 * p)) | (tl & (Ns - 1));
        __syncthreads();  // every thread is done reading the previous contents
#pragma unroll
        for (int r = 0; r < 8; ++r) lds[swz(base + r * Ns)] = x[r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = lds[swz(tl + G * j)];
        if (p < P8) {
            const double2 w1 = tw.w[p - 1], w2 = cmul(w1, w1), w3 = cmul(w1, w2), w4 = cmul(w2, w2);
            const double2 w5 = cmul(w1, w4), w6 = cmul(w2, w4), w7 = cmul(w3, w4);
            x[1] = twmul<FWD>(x[1], w1); x[2] = twmul<FWD>(x[2], w2); x[3] = twmul<FWD>(x[3], w3); x[4] = twmul<FWD>(x[4], w4);
            x[5] = twmul<FWD>(x[5], w5); x[6] = twmul<FWD>(x[6], w6); x[7] = twmul<FWD>(x[7], w7);
            dft_small<8, FWD>(x);
        } else if constexpr (T == 2) {  // 4 radix-2 butterflies: points u and u + 4, twiddle W_N^(tl + G u)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double2 a = x[u], b = twmul<FWD>(x[u + 4], tw.wt[u]);
                x[u] = cadd(a, b); x[u + 4] = csub(a, b);
            }
        } else if constexpr (T == 4) {  // 2 radix-4 butterflies: points u, u + 2, u + 4, u + 6
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const double2 w1 = tw.wt[u], w2 = cmul(w1, w1), w3 = cmul(w1, w2);
                double2 y[4] = {x[u], twmul<FWD>(x[u + 2], w1), twmul<FWD>(x[u + 4], w2), twmul<FWD>(x[u + 6], w3)};
                dft_small<4, FWD>(y);
                x[u] = y[0]; x[u + 2] = y[1]; x[u + 4] = y[2]; x[u + 6] = y[3];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// The register classes are only used for rings without aliasing (Bluestein classes: mlim < n / 2 - 5; direct classes: mlim <= n / 2,
// see pl_plan_create), so every spectrum bin k = 4 k1 + k2 holds at most one term: frequency +k when k <= mlim, else -(n - k) when
// n - k <= mlim -- except the bin n / 2 of a direct ring with mlim = n / 2, which the kernels treat on its own.
// Slot idx = tl + G j of a thread is sub-DFT bin k1 = idx (direct, or Bluestein input c = idx >= 0) or, in the
// Bluestein classes, c = idx - N < 0, i.e. k1 = q + c.  sgn: +1 positive-frequency side, -1 negative side, 0 out of band.
struct FastBin { int k1, cabs, sgn; };
__device__ __forceinline__ FastBin fast_bin(bool blue, int idx, int N, int q, int K)
{
    FastBin b;
    if (!blue) { b.k1 = idx; b.cabs = 0; b.sgn = 2 * idx < q ? 1 : -1; return b; }  // direct: first half +, second half -
    if (idx <= K) { b.k1 = idx; b.cabs = idx; b.sgn = 1; return b; }
    if (idx >= N - K) { b.k1 = q - (N - idx); b.cabs = N - idx; b.sgn = -1; return b; }
    b.k1 = 0; b.cabs = 0; b.sgn = 0;
    return b;
}

// ---- synthesis: radix-4 on the pixel side ---------------------------------------------------------------------------
// pixel j = j1 + q j2, bin k = 4 k1 + k2: x_(j1 + q j2) = sum_k2 i^(j2 k2) e^{2 pi i j1 k2 / n} [sum_k1 e^{2 pi i j1 k1 / q} X_(4 k1 + k2)].
// All four sub-DFTs stay resident (d[4][8]) because every pixel needs all of them; the workgroup keeps to 256 registers
// per thread (two workgroups per CU).  The sub-DFT inputs are band-limited (|c| <= K), so the Bluestein classes use the
// same convolution sizes and filter tables as the analysis (side A lists and tables).
// WGT: every pixel is multiplied by wgt (one map for all components) on the way out -- the inverse-noise weighting of the CG operators.
// SPLIT (Bluestein only): the ring's convolution of size 2 N as two of size N that share the forward transform -- pixels
// j1 < N / 2 from the first filter spectrum, j1 >= N / 2 from the second (thread tl owns j1 = tl + G j either way: j < 4 and
// j >= 4).  The forward spectrum waits in a thread-private LDS slot while the first half is transformed back.
template <int N, bool BLUE, bool WGT = false, bool SPLIT = false>
__global__ __launch_bounds__(N / 8, 2) void k_phase2map_fast(DevPlan P, DevFFT F, const int *__restrict__ pairs, const int *__restrict__ mlim,
                                                             int ncomp, const double *__restrict__ phase, double *__restrict__ map,
                                                             const double *__restrict__ wgt)
{
    extern __shared__ double2 lds[];
    constexpr int G = N / 8;
    const int tl0 = threadIdx.x;
    const int ip = pairs[blockIdx.x], comp = blockIdx.y;
    const int n = P.nphi[ip], q = n >> 2;
    constexpr bool blue = BLUE;  // Bluestein rings and direct (q == N) rings run as separate launches: half the code each
    const int K = F.K2of[q];
    const double2 *__restrict__ chirp = F.chirp + F.woff[q];
    const double2 *__restrict__ filt = F.A.filt + F.A.coff[q];
    const int ml = min(mlim[ip], P.mmax);
    const bool shifted = P.phi0[ip] != 0.0;
    const double inv_n = 1.0 / n;
    constexpr int estride = 4;  // phase array [ring pair][component][m][N re, N im, S re, S im]: a component's orders are contiguous
    const double *__restrict__ ph = phase + ((int64_t)ip * ncomp + comp) * P.mstride * estride;
    Tw8<N> tw;
    tw8_load<N>(tw, tl0, F.tw, F.Mtw);

    // gather: d[k2][j] = z_(4 k1 + k2) (times the chirp), z = f_N + i f_S (+ side) or conj(f_N) + i conj(f_S) (- side).
    // Ring offset phase e^{i pi m / n} of the shifted rings by recurrence: with Pk = e^{i pi k / n}, k = 4 idx + k2,
    // the + side (m = k) needs Pk and the - side (m = 4 (N - idx) - k2) needs e^{i pi 4 N / n} conj(Pk).
    double2 d[4][8];
    {
        const int tl = fresh(tl0);
        double2 pj = make_double2(1., 0.), pstep = pj, s1 = pj, s2 = pj, s3 = pj, uneg = pj;
        if (shifted) {  // (the wave-uniform factors come from the plan's per-ring table F.ringc, the per-thread one from sincos)
            pj = cispi(4.0 * tl * inv_n); pstep = F.ringc[4 * q + 2];
            s1 = F.ringc[4 * q]; s2 = cmul(s1, s1); s3 = cmul(s2, s1);
            uneg = F.ringc[4 * q + 3];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // Round 5: a block of G slots that holds no in-band bin for any thread of the workgroup is skipped outright (a workgroup-uniform
            // branch per j: no loads, no chirp, no selects).  With a band of K ~ 300-410 bins (the cap rings of nside 2048 at lmax = nside)
            // only j = 0, 1 (orders +k) and j = 6, 7 (orders n - k) survive: 16 of the 32 loads of a thread.
            const bool band_j = blue ? (G * j <= K || G * (j + 1) - 1 >= N - K)
                                     : (4 * G * j <= ml || n - 4 * (G * (j + 1) - 1) - 3 <= ml);
            if (!band_j) {
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) d[k2][j] = make_double2(0., 0.);
                pj = cmul(pj, pstep);
                continue;
            }
            const FastBin b = fast_bin(blue, tl + G * j, N, q, K);
            double2 cw = make_double2(1., 0.);
            if (blue) cw = chirp[b.cabs];
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                // branch-free: out-of-band bins load entry m = 0 (always valid) and are zeroed by the select below
                const int k = 4 * b.k1 + k2;
                const int mm = b.sgn > 0 ? k : n - k;
                const bool have = b.sgn != 0 && mm <= ml;
                const int m = have ? mm : 0;
                const double4 f = *reinterpret_cast<const double4 *>(ph + (int64_t)m * estride);
                double2 fn = make_double2(f.x, f.y), fs = make_double2(f.z, f.w);
                if (shifted) {
                    const double2 pk = cmul(pj, k2 == 0 ? make_double2(1., 0.) : k2 == 1 ? s1 : k2 == 2 ? s2 : s3);
                    const double2 pm = b.sgn > 0 ? pk : cmulc(uneg, pk);
                    fn = cmul(fn, pm); fs = cmul(fs, pm);
                }
                double2 z;
                z.x = have ? (b.sgn > 0 ? fn.x - fs.y : fn.x + fs.y) : 0.0;
                z.y = have ? (b.sgn > 0 ? fn.y + fs.x : -fn.y + fs.x) : 0.0;
                if (!blue && k2 == 0 && j == 4) {
                    // direct rings with mlim = n / 2 (lmax = 2 nside on the belt): bin n / 2 = slot q / 2 of sub-DFT 0 (tl = 0, j = 4)
                    // holds the order n / 2 from both sides, (f_N + i f_S) + (conj f_N + i conj f_S) = 2 Re f_N + 2 i Re f_S
                    const bool nyq = have && tl == 0;
                    z.x = nyq ? 2.0 * fn.x : z.x;
                    z.y = nyq ? 2.0 * fs.x : z.y;
                }
                d[k2][j] = cmul(z, cw);
            }
            pj = cmul(pj, pstep);
        }
    }
    // the four sub-DFTs, in place
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        if constexpr (SPLIT) {
            fft8<N, true>(d[k2], lds, tl0, tw);
            double2 *stash = lds + N;  // thread-private slots (the forward spectrum, then the first half of the pixels): no barrier
            {
                const int tl = fresh(tl0);
#pragma unroll
                for (int j = 0; j < 8; ++j) { stash[tl + G * j] = d[k2][j]; d[k2][j] = cmul(d[k2][j], filt[tl + G * j]); }
            }
            fft8<N, false>(d[k2], lds, tl0, tw);
            {
                const int tl = fresh(tl0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double2 a = stash[tl + G * j];
                    if (j < 4) stash[tl + G * j] = d[k2][j];  // pixels j1 < N / 2 wait in the slots the spectrum has left
                    d[k2][j] = cmul(a, filt[N + tl + G * j]);
                    if ((j & 1) == 1) phase_fence();
                }
            }
            fft8<N, false>(d[k2], lds, tl0, tw);
            {
                const int tl = fresh(tl0);
#pragma unroll
                for (int j = 0; j < 4; ++j) { d[k2][j + 4] = d[k2][j]; d[k2][j] = stash[tl + G * j]; }
            }
        } else if (blue) {
            fft8<N, true>(d[k2], lds, tl0, tw);
            const int tl = fresh(tl0);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[k2][j] = cmul(d[k2][j], filt[tl + G * j]);
            fft8<N, false>(d[k2], lds, tl0, tw);
        } else {
            fft8<N, false>(d[k2], lds, tl0, tw);
        }
    }
    // twiddle e^{2 pi i j1 k2 / n} (recurrence over j), radix-4 butterfly over k2, pixels j = j1 + q j2
    double *__restrict__ mp = map + (int64_t)comp * P.npix;
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    const int tl = fresh(tl0);
    double2 e1 = cispi(2.0 * tl * inv_n);
    const double2 estep = F.ringc[4 * q + 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int j1 = tl + G * j;
        double2 cw = make_double2(1., 0.);
        if (blue) cw = chirp[min(j1, q - 1)];
        if (j1 < q) {
            const double2 e2 = cmul(e1, e1), e3 = cmul(e2, e1);
            double2 y[4] = {cmul(d[0][j], cw), cmul(d[1][j], cmul(cw, e1)), cmul(d[2][j], cmul(cw, e2)), cmul(d[3][j], cmul(cw, e3))};
            dft_small<4, false>(y);  // y[j2] = sum_k2 i^(j2 k2) y_k2
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                if constexpr (WGT) {
                    mp[on + j1 + q * j2] = y[j2].x * wgt[on + j1 + q * j2];
                    if (os >= 0) mp[os + j1 + q * j2] = y[j2].y * wgt[os + j1 + q * j2];
                } else {
                    mp[on + j1 + q * j2] = y[j2].x;
                    if (os >= 0) mp[os + j1 + q * j2] = y[j2].y;
                }
            }
        }
        e1 = cmul(e1, estep);
    }
}

// ---- analysis: radix-4 on the pixel side ----------------------------------------------------------------------------
// pixel j = j1 + q j2, bin k = 4 k1 + k2: V_(4 k1 + k2) = sum_j1 e^{2 pi i j1 k1 / q} [e^{2 pi i j1 k2 / n} sum_j2 i^(j2 k2) conj(z)_(j1 + q j2)].
// Sub-DFTs are processed as the pairs (k2 = 0, 2) and (1, 3) -- the mirror bin n - k of k2 lives in sub-DFT (4 - k2) mod 4,
// so each pair is self-contained -- with the ring pixels loaded again for the second pair (coalesced, from L2).
// SPLIT (Bluestein only): the convolution of size 2 N as two of size N that share the inverse transform -- the pixels j1 < N / 2
// (the thread's points j < 4) and j1 >= N / 2 (j >= 4) are transformed separately, multiplied by their own filter spectra, added.
template <int N, bool BLUE, bool SPLIT = false>
__global__ __launch_bounds__(N / 8, 2) void k_map2phase_fast(DevPlan P, DevFFT F, const int *__restrict__ pairs, const int *__restrict__ mlim,
                                                          int ncomp, const double *map, double *phase)
{
    extern __shared__ double2 lds[];
    constexpr int G = N / 8;
    const int tl0 = threadIdx.x;
    const int ip = pairs[blockIdx.x], comp = blockIdx.y;
    const int n = P.nphi[ip], q = n >> 2;
    constexpr bool blue = BLUE;  // Bluestein rings and direct (q == N) rings run as separate launches: half the code each
    const int K = F.K2of[q];
    const double2 *__restrict__ chirp = F.chirp + F.woff[q];
    const double2 *__restrict__ filt = F.A.filt + F.A.coff[q];
    const int ml = min(mlim[ip], P.mmax);
    const bool shifted = P.phi0[ip] != 0.0;
    const double inv_n = 1.0 / n;
    constexpr int estride = 4;  // phase array [ring pair][component][m][N re, N im, S re, S im]
    double *ph = phase + ((int64_t)ip * ncomp + comp) * P.mstride * estride;
    const double wgt = 0.5 * 4.0 * 3.14159265358979323846 / (double)P.npix;  // includes the 1/2 of the N/S split
    const double *mp = fft_input_map(F, P, map, comp);  // no __restrict__: see k_phase2map_fast
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    const bool has_s = os >= 0;
    Tw8<N> tw;
    tw8_load<N>(tw, tl0, F.tw, F.Mtw);
    const double2 e10 = cispi(2.0 * tl0 * inv_n), e1step = F.ringc[4 * q + 1];
    double2 pj0 = make_double2(1., 0.), pstep = pj0, s1 = pj0;
    if (shifted) { pj0 = cispi(4.0 * tl0 * inv_n); pstep = F.ringc[4 * q + 2]; s1 = F.ringc[4 * q]; }

    // F_N, F_S of order m = 4 k1 + k2 (+ side owner): a = V_m (own register), vm = V_(n - m) (from the - side owner via LDS)
    auto emit_side = [&](int k2, const double2 (&own)[8], const double2 (&other)[8]) {
        const int tl = fresh(tl0);  // per-call copy: keeps the bin bookkeeping of one call from living across the others
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const FastBin b = fast_bin(blue, tl + G * j, N, q, K);
            if (b.sgn < 0 || (b.sgn > 0 && b.k1 == 0)) lds[swz(b.k1)] = other[j];  // - side values of the mirror sub-DFT
        }
        __syncthreads();
        double2 pjj = pj0;
        launder(pjj);
        const double2 sk = k2 == 0 ? make_double2(1., 0.) : k2 == 1 ? s1 : k2 == 2 ? cmul(s1, s1) : cmul(s1, cmul(s1, s1));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const FastBin b = fast_bin(blue, tl + G * j, N, q, K);
            const int m = 4 * b.k1 + k2;
            // (direct rings with mlim = n / 2: the order n / 2 sits in the first - side slot, q / 2 of sub-DFT 0, and mirrors into itself)
            const bool nyq = !blue && k2 == 0 && j == 4 && tl == 0;
            if ((b.sgn > 0 || nyq) && m <= ml) {
                const int k1m = k2 == 0 ? (b.k1 == 0 ? 0 : q - b.k1) : q - 1 - b.k1;
                const double2 a = cconj(own[j]);
                const double2 vm = lds[swz(k1m)];
                double2 fn = cadd(a, vm);
                const double2 dd = csub(a, vm);
                double2 fs = make_double2(dd.y, -dd.x);   // (conj(V_m) - V_(n-m)) / i
                if (shifted) { const double2 pk = cmul(pjj, sk); fn = cmulc(fn, pk); fs = cmulc(fs, pk); }  // e^{-i pi m / n}
                double4 o;
                o.x = fn.x * wgt; o.y = fn.y * wgt;
                o.z = has_s ? fs.x * wgt : 0.0; o.w = has_s ? fs.y * wgt : 0.0;
                *reinterpret_cast<double4 *>(ph + (int64_t)m * estride) = o;
            }
            pjj = cmul(pjj, pstep);
            if ((j & 1) == 1) phase_fence();  // at most two j (8 loads) in flight
        }
    };

#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int k2a = pp, k2b = pp + 2;   // (0, 2) then (1, 3)
        phase_fence();  // the loads of this pass stay behind the barriers of the previous one
        const int tl = fresh(tl0);  // see k_phase2map_fast
        double2 d[2][8];
        double2 e1 = e10;
        launder(e1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int j1 = tl + G * j;
            double2 y[4];
            const int j1c = min(j1, q - 1);  // branch-free loads; slots beyond the ring are zeroed by the select
            const double *mps = has_s ? mp + os : mp + on;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const double vn = mp[on + j1c + q * j2], vs = mps[j1c + q * j2];
                y[j2].x = j1 < q ? vn : 0.0;
                y[j2].y = (j1 < q && has_s) ? -vs : 0.0;
            }
            dft_small<4, false>(y);  // y[k2] = sum_j2 i^(j2 k2) conj(z)_(j1 + q j2)
            double2 cw = make_double2(1., 0.);
            if (blue) cw = chirp[j1c];
            const double2 e2 = cmul(e1, e1);
            if (pp == 0) { d[0][j] = cmul(y[0], cw); d[1][j] = cmul(y[2], cmul(cw, e2)); }
            else { d[0][j] = cmul(y[1], cmul(cw, e1)); d[1][j] = cmul(y[3], cmul(cw, cmul(e2, e1))); }
            e1 = cmul(e1, e1step);
            if ((j & 1) == 1) phase_fence();  // at most two j (8 loads) in flight
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (SPLIT) {
                double2 hi[8] = {d[h][4], d[h][5], d[h][6], d[h][7], make_double2(0., 0.), make_double2(0., 0.), make_double2(0., 0.),
                                 make_double2(0., 0.)};
#pragma unroll
                for (int j = 4; j < 8; ++j) d[h][j] = make_double2(0., 0.);
                fft8<N, true>(d[h], lds, tl, tw);
                phase_fence();
#pragma unroll
                for (int j = 0; j < 8; ++j) d[h][j] = cmul(d[h][j], filt[(N - (tl + G * j)) & (N - 1)]);
                fft8<N, true>(hi, lds, tl, tw);
                phase_fence();
#pragma unroll
                for (int j = 0; j < 8; ++j) d[h][j] = cadd(d[h][j], cmul(hi[j], filt[N + ((N - (tl + G * j)) & (N - 1))]));
                fft8<N, false>(d[h], lds, tl, tw);
                phase_fence();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const FastBin b = fast_bin(blue, tl + G * j, N, q, K);
                    d[h][j] = cmul(d[h][j], chirp[b.cabs]);
                }
            } else if (blue) {
                fft8<N, true>(d[h], lds, tl, tw);
                phase_fence();
#pragma unroll
                for (int j = 0; j < 8; ++j) d[h][j] = cmul(d[h][j], filt[(N - (tl + G * j)) & (N - 1)]);  // spectrum of the mirrored filter
                fft8<N, false>(d[h], lds, tl, tw);
                phase_fence();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const FastBin b = fast_bin(blue, tl + G * j, N, q, K);
                    d[h][j] = cmul(d[h][j], chirp[b.cabs]);  // out-of-band slots: chirp[0], value never used
                }
            } else {
                fft8<N, false>(d[h], lds, tl, tw);
            }
        }
        if (pp == 0) { emit_side(k2a, d[0], d[0]); emit_side(k2b, d[1], d[1]); }   // k2 = 0 and 2 mirror into themselves
        else { emit_side(k2a, d[0], d[1]); emit_side(k2b, d[1], d[0]); }           // k2 = 1 <-> 3
    }
}

// =====================================================================================================
// Wavefront-private transforms for the direct rings of sub-DFT length q = N = 2048 (the belt of an nside-2048 grid: half its pixels)
// =====================================================================================================
// The register kernels above run a size-N transform as three or four radix-8 passes with the whole workgroup exchanging its points
// through LDS between passes: 24 barrier pairs per ring pair, and a wave that spends half its cycles waiting at them.  Here ONE
// wavefront owns a whole sub-DFT: 32 points per lane, N = 32 x 64 --
//   (1) DFT-32 over the lane's own points (registers only),  (2) twiddle W_N^(b L),  (3) a 64 x 32 transpose through a strip of LDS
//   that belongs to the wave (no barrier: a wave's LDS operations execute in order),  (4) DFT-32 again,  (5) one radix-2 step across
//   lane pairs (DPP) --
// so the four sub-DFTs of a ring pair proceed on the four waves of a workgroup without ever synchronising; only the radix-4 step on
// the pixel side, which needs all four, goes through LDS between waves (four barriers per ring pair).  LDS operations per point
// fall from 3 writes + 3 reads to 2 + 2.  Index algebra (inverse transform, W = e^{+2 pi i / N}): input k1 = L + 64 j (lane L, register
// j), output j1 = 32 a + b:  W^(j1 k1) = W_64^(a L) W_N^(b L) W_32^(b j);  a = a' + 32 s:  y[32 a + b] = E_0[a'] + (-1)^s W_64^(a') E_1[a'],
// E_h[a'] = sum_i v[2 i + h][b] W_32^(a' i).
__device__ constexpr double kC64[33] = {1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867, 0.881921264348355, 0.8314696123025452, 0.773010453362737, 0.7071067811865476, 0.6343932841636455, 0.5555702330196022, 0.47139673682599764, 0.3826834323650898, 0.2902846772544624, 0.19509032201612828, 0.0980171403295606, 0.0, -0.0980171403295606, -0.19509032201612828, -0.2902846772544624, -0.3826834323650898, -0.47139673682599764, -0.5555702330196022, -0.6343932841636455, -0.7071067811865476, -0.773010453362737, -0.8314696123025452, -0.881921264348355, -0.9238795325112867, -0.9569403357322088, -0.9807852804032304, -0.9951847266721969, -1.0};
__device__ constexpr double kS64[33] = {0.0, 0.0980171403295606, 0.19509032201612828, 0.2902846772544624, 0.3826834323650898, 0.47139673682599764, 0.5555702330196022, 0.6343932841636455, 0.7071067811865476, 0.773010453362737, 0.8314696123025452, 0.881921264348355, 0.9238795325112867, 0.9569403357322088, 0.9807852804032304, 0.9951847266721969, 1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867, 0.881921264348355, 0.8314696123025452, 0.773010453362737, 0.7071067811865476, 0.6343932841636455, 0.5555702330196022, 0.47139673682599764, 0.3826834323650898, 0.2902846772544624, 0.19509032201612828, 0.0980171403295606, 0.0};

__device__ constexpr int kBrev5[32] = {0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30, 1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31};

// d * e^{-+ 2 pi i t / 64} (FWD: -), t a compile-time index in [0, 32]
template <bool FWD>
__device__ __forceinline__ double2 rot64(double2 d, int t)
{
    if (t == 0) return d;
    if (t == 16) return mul_i<FWD>(d);
    if (t == 32) return make_double2(-d.x, -d.y);
    const double c = kC64[t], s = FWD ? -kS64[t] : kS64[t];
    return make_double2(d.x * c - d.y * s, d.x * s + d.y * c);
}

// In-register DFT-32, decimation in frequency: natural order in, x[i] = X[kBrev5[i]] out (the register index is a compile-time
// constant everywhere, so the permutation costs nothing).  FWD: e^{-2 pi i jk/32}, else e^{+2 pi i jk/32}; unnormalised.
template <bool FWD>
__device__ __forceinline__ void dft32(double2 (&x)[32])
{
#pragma unroll
    for (int len = 32; len >= 2; len >>= 1) {
        const int half = len >> 1, step = 64 / len;  // twiddle W_len^k = W_64^(k step)
#pragma unroll
        for (int blk = 0; blk < 32; blk += len) {
#pragma unroll
            for (int k = 0; k < half; ++k) {
                const double2 a = x[blk + k], b = x[blk + k + half];
                x[blk + k] = cadd(a, b);
                x[blk + k + half] = rot64<FWD>(csub(a, b), k * step);
            }
        }
    }
}

__device__ __forceinline__ double dpp_xor1(double v)  // the value of lane ^ 1 (quad_perm [1, 0, 3, 2])
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// orders a wave's own LDS traffic for the compiler (the hardware executes one wave's DS instructions in order)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kWaveN = 2048, kWaveRow = 66, kWaveReg = 16 * kWaveRow;  // LDS strip of a wave in double2 units: 32 rows of 64 + 2 (pad) doubles = 16.5 KB
constexpr int kWaveLds = 4 * kWaveReg * (int)sizeof(double2);

// synthesis of the direct rings, q = N = 2048: workgroup = 4 waves = the four sub-DFTs k2 of one ring pair and component
template <bool WGT>
__global__ __launch_bounds__(256, 2) void k_phase2map_wave(DevPlan P, DevFFT F, const int *__restrict__ pairs, const int *__restrict__ mlim,
                                                           int ncomp, const double *__restrict__ phase, double *__restrict__ map,
                                                           const double *__restrict__ wgt)
{
    extern __shared__ double2 lds[];
    constexpr int N = kWaveN;
    const int tl = threadIdx.x, L = tl & 63, k2 = tl >> 6, h = L & 1, bl = L >> 1;
    const int ip = pairs[blockIdx.x], comp = blockIdx.y;
    const int n = P.nphi[ip], q = n >> 2;  // q == N (the launcher's list holds those rings only)
    const int ml = min(mlim[ip], P.mmax);
    const bool shifted = P.phi0[ip] != 0.0;
    const double inv_n = 1.0 / n;
    constexpr int estride = 4;
    const double *__restrict__ ph = phase + ((int64_t)ip * ncomp + comp) * P.mstride * estride;
    double2 *reg = lds + k2 * kWaveReg;

    // ---- gather: x[j] = z of bin k = 4 k1 + k2, k1 = L + 64 j (first half of the sub-DFT: order +k; second half: order n - k) --------
    // Two regimes, chosen per ring (wave-uniform): with mlim < 2304 (every ring of a grid with lmax <= nside + 255) only the register slots
    // j <= 8 (orders +k) and j >= 23 (orders n - k) can hold in-band bins -- 18 loads, all in flight at once, the other slots are zero;
    // otherwise all 32 slots, 16 loads in flight at a time.
    double2 x[32];
    auto gather = [&](auto shifted_c, auto narrow_c) {
        constexpr bool SH = decltype(shifted_c)::value, NARROW = decltype(narrow_c)::value;
        double2 p0 = make_double2(1., 0.);
        if constexpr (SH) p0 = cispi((4.0 * L + k2) * inv_n);  // e^{i pi (4 L + k2) / n}; the step to the next j is e^{i pi 256 / n} = e^{2 pi i / 64}
#pragma unroll
        for (int g = 0; g < 2; ++g) {  // batch g: slots j = 16 g .. 16 g + 15 (narrow: the in-band ones of both halves in batch 0)
            double4 f[18];
            bool hv[18];
#pragma unroll
            for (int t = 0; t < (NARROW ? 18 : 16); ++t) {
                const int j = NARROW ? (t < 9 ? t : 14 + t) : 16 * g + t;
                if (NARROW && g == 1) continue;
                const int k = 4 * (L + 64 * j) + k2;
                const int mm = j < 16 ? k : n - k;
                hv[t] = mm <= ml;
                f[t] = *reinterpret_cast<const double4 *>(ph + (int64_t)(hv[t] ? mm : 0) * estride);  // branch-free: out-of-band bins read order 0, zeroed below
            }
            phase_fence();
#pragma unroll
            for (int t = 0; t < (NARROW ? 18 : 16); ++t) {
                const int j = NARROW ? (t < 9 ? t : 14 + t) : 16 * g + t;
                if (NARROW && g == 1) continue;
                const bool plus = j < 16, have = hv[t];
                double2 fn = make_double2(f[t].x, f[t].y), fs = make_double2(f[t].z, f[t].w);
                if constexpr (SH) {
                    const double2 pk = rot64<false>(p0, j);                            // e^{i pi k / n}
                    const double2 pm = plus ? pk : make_double2(-pk.x, pk.y);          // order n - k: e^{i pi} conj
                    fn = cmul(fn, pm); fs = cmul(fs, pm);
                }
                double2 z;
                z.x = have ? (plus ? fn.x - fs.y : fn.x + fs.y) : 0.0;
                z.y = have ? (plus ? fn.y + fs.x : -fn.y + fs.x) : 0.0;
                if (j == 16) {  // bin n / 2 (k2 = 0, k1 = q / 2) of a ring with mlim = n / 2 holds that order from both sides
                    const bool nyq = have && k2 == 0 && L == 0;
                    z.x = nyq ? 2.0 * fn.x : z.x;
                    z.y = nyq ? 2.0 * fs.x : z.y;
                }
                x[j] = z;
            }
            phase_fence();
        }
        if constexpr (NARROW) {
#pragma unroll
            for (int j = 9; j < 23; ++j) x[j] = make_double2(0., 0.);
        }
    };
    if (ml < 2304) { if (shifted) gather(std::true_type{}, std::true_type{}); else gather(std::false_type{}, std::true_type{}); }
    else { if (shifted) gather(std::true_type{}, std::false_type{}); else gather(std::false_type{}, std::false_type{}); }
    // ---- (1) DFT-32 over j, (2) twiddle W_N^(b L) ----------------------------------------------------------------------------------
    __builtin_amdgcn_sched_barrier(0);  // (phases stay apart: hipcc otherwise hoists the table / LDS accesses of later phases into earlier ones)
    dft32<false>(x);
    __builtin_amdgcn_sched_barrier(0);
    {
        const int ts = F.Mtw / N;  // F.tw[t] = e^{-2 pi i t / Mtw}, t < Mtw / 2: every index below is < Mtw / 2 (16 L <= 1008 < N / 2)
        double2 lo[8], hi[4];
#pragma unroll
        for (int s = 0; s < 3; ++s) lo[1 << s] = cconj(F.tw[(L << s) * ts]);
        hi[1] = cconj(F.tw[(L << 3) * ts]); hi[2] = cconj(F.tw[(L << 4) * ts]);
        lo[3] = cmul(lo[1], lo[2]); lo[5] = cmul(lo[1], lo[4]); lo[6] = cmul(lo[2], lo[4]); lo[7] = cmul(lo[3], lo[4]);
        hi[3] = cmul(hi[1], hi[2]);
#pragma unroll
        for (int b = 1; b < 32; ++b) {
            const double2 w = (b & 7) == 0 ? hi[b >> 3] : ((b >> 3) == 0 ? lo[b & 7] : cmul(lo[b & 7], hi[b >> 3]));
            x[kBrev5[b]] = cmul(x[kBrev5[b]], w);
        }
    }
    // ---- (3) transpose through the wave's own strip, real parts then imaginary parts (a strip holds 32 rows of 64 doubles): lane
    // (b = L / 2, h = L % 2) receives v[2 i + h][b], i < 32.  Every lane takes part in both rounds; after a round's stores the x of
    // that part are dead, so the live set stays near 128 registers.  Row stride 66 doubles: the 16 rows met by a group of 32 lanes of a
    // ds_read_b64 start 4 banks apart (conflict-free), the stores are contiguous.
    __builtin_amdgcn_sched_barrier(0);
    double2 u[32];
    {
        double *regd = reinterpret_cast<double *>(reg);
        const double *row = regd + bl * kWaveRow + h;
#pragma unroll
        for (int b = 0; b < 32; ++b) regd[b * kWaveRow + L] = x[kBrev5[b]].x;
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 32; ++i) u[i].x = row[2 * i];
        wave_lds_fence();
#pragma unroll
        for (int b = 0; b < 32; ++b) regd[b * kWaveRow + L] = x[kBrev5[b]].y;
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 32; ++i) u[i].y = row[2 * i];
        wave_lds_fence();
    }
    // ---- (4) DFT-32 over i, (5) radix-2 across the lane pair: lane h keeps a = a' + 32 h -------------------------------------------------
    __builtin_amdgcn_sched_barrier(0);
    dft32<false>(u);
    __builtin_amdgcn_sched_barrier(0);
    {
        const double hd = (double)h, sg = 1.0 - 2.0 * hd;
#pragma unroll
        for (int a = 0; a < 32; ++a) {
            const double2 own = u[kBrev5[a]];
            // lane 0 sends E_0, lane 1 sends W_64^(a') E_1: the factor is 1 or W by lane, formed without a select
            const double2 f = make_double2(fma(hd, kC64[a] - 1.0, 1.0), hd * kS64[a]);
            const double2 t = (a == 0) ? own : cmul(own, f);
            const double2 r = make_double2(dpp_xor1(t.x), dpp_xor1(t.y));
            u[kBrev5[a]] = make_double2(fma(sg, t.x, r.x), fma(sg, t.y, r.y));  // lane 0: E_0 + W E_1; lane 1: E_0 - W E_1
        }
    }
    // ---- radix-4 over k2 on the pixel side: the sub-DFT values of 512 pixels at a time meet in LDS (double-buffered, own strips) --------
    // lane (b, h), register a' holds pixel j1 = 1024 h + 32 a' + b of sub-DFT k2; thread tl combines pixels j1 = 1024 uu + 256 c + tl
    __builtin_amdgcn_sched_barrier(0);
    double *__restrict__ mp = map + (int64_t)comp * P.npix;
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    const double2 e0 = cispi(2.0 * tl * inv_n);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double2 *wb = reg + (c & 1) * 520 + h * 260 + bl;
#pragma unroll
        for (int a = 0; a < 8; ++a) wb[32 * a] = u[kBrev5[8 * c + a]];
        __syncthreads();
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const int j1 = 1024 * uu + 256 * c + tl;
            const double2 *rb = lds + (c & 1) * 520 + uu * 260 + tl;
            const double2 d0 = rb[0], d1 = rb[kWaveReg], d2 = rb[2 * kWaveReg], d3 = rb[3 * kWaveReg];
            const double2 e1 = rot64<false>(e0, 2 * (4 * uu + c));  // e^{2 pi i j1 / n}
            const double2 e2 = cmul(e1, e1), e3 = cmul(e2, e1);
            double2 y[4] = {d0, cmul(d1, e1), cmul(d2, e2), cmul(d3, e3)};
            dft_small<4, false>(y);  // y[j2] = sum_k2 i^(j2 k2) y_k2
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                if constexpr (WGT) {
                    mp[on + j1 + q * j2] = y[j2].x * wgt[on + j1 + q * j2];
                    if (os >= 0) mp[os + j1 + q * j2] = y[j2].y * wgt[os + j1 + q * j2];
                } else {
                    mp[on + j1 + q * j2] = y[j2].x;
                    if (os >= 0) mp[os + j1 + q * j2] = y[j2].y;
                }
            }
        }
    }
}

// analysis of the direct rings, q = N = 2048: the mirror image of k_phase2map_wave.  All 256 threads first work as pixel columns (load the
// 4 + 4 pixels j1 + q j2 of both rings, radix-4 over j2, twiddle e^{2 pi i j1 k2 / n}) and hand sub-DFT k2's input to wave k2 through that
// wave's LDS strip (512 columns at a time, double-buffered: four barriers); every wave then transforms its sub-DFT on its own
// (wave_fft: no barrier) and ends with bin k1 = 32 (a' + 32 h) + b in lane (b, h), register a'.  The lanes h = 1 hold the upper half of the
// bins -- the negative-frequency side -- and publish it in their strip (one barrier); the lanes h = 0 own the orders m = 4 k1 + k2 <= mlim
// and combine V_m with V_(n - m), which lives in sub-DFT (4 - k2) mod 4.
__global__ __launch_bounds__(256, 2) void k_map2phase_wave(DevPlan P, DevFFT F, const int *__restrict__ pairs, const int *__restrict__ mlim,
                                                           int ncomp, const double *map, double *phase)
{
    extern __shared__ double2 lds[];
    constexpr int N = kWaveN;
    const int tl = threadIdx.x, L = tl & 63, k2 = tl >> 6, h = L & 1, bl = L >> 1;
    const int ip = pairs[blockIdx.x], comp = blockIdx.y;
    const int n = P.nphi[ip], q = n >> 2;  // q == N
    const int ml = min(mlim[ip], P.mmax);
    const bool shifted = P.phi0[ip] != 0.0;
    const double inv_n = 1.0 / n;
    constexpr int estride = 4;
    double *ph = phase + ((int64_t)ip * ncomp + comp) * P.mstride * estride;
    const double wgt = 0.5 * 4.0 * 3.14159265358979323846 / (double)P.npix;  // includes the 1/2 of the N/S split
    const double *mp = fft_input_map(F, P, map, comp);
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    const bool has_s = os >= 0;
    const double *mps = has_s ? mp + os : mp + on;
    double2 *reg = lds + k2 * kWaveReg;

    // ---- pixel columns -> sub-DFT inputs: thread tl makes columns j1 = 1024 uu + 256 c + tl; wave k2, lane L takes j1 = L + 64 j -------------
    double2 x[32];
    {
        const double2 e0 = cispi(2.0 * tl * inv_n);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double2 yk[2][4];
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int j1 = 1024 * uu + 256 * c + tl;
                double2 y[4];
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const double vn = mp[on + j1 + q * j2], vs = mps[j1 + q * j2];
                    y[j2] = make_double2(vn, has_s ? -vs : 0.0);
                }
                dft_small<4, false>(y);  // y[k2] = sum_j2 i^(j2 k2) conj(z)_(j1 + q j2)
                const double2 e1 = rot64<false>(e0, 2 * (4 * uu + c));  // e^{2 pi i j1 / n}
                const double2 e2 = cmul(e1, e1), e3 = cmul(e2, e1);
                yk[uu][0] = y[0]; yk[uu][1] = cmul(y[1], e1); yk[uu][2] = cmul(y[2], e2); yk[uu][3] = cmul(y[3], e3);
            }
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                double2 *wb = lds + (c & 1) * 520 + uu * 260 + tl;
#pragma unroll
                for (int k = 0; k < 4; ++k) wb[k * kWaveReg] = yk[uu][k];
            }
            __syncthreads();
            const double2 *rb = reg + (c & 1) * 520 + L;
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
#pragma unroll
                for (int w = 0; w < 4; ++w) x[16 * uu + 4 * c + w] = rb[uu * 260 + 64 * w];
            }
        }
    }
    // ---- the sub-DFT (inverse sign, as in the synthesis): DFT-32, twiddle, transpose, DFT-32, radix-2 across the lane pair ----------------------
    __builtin_amdgcn_sched_barrier(0);
    dft32<false>(x);
    __builtin_amdgcn_sched_barrier(0);
    {
        const int ts = F.Mtw / N;
        double2 lo[8], hi[4];
#pragma unroll
        for (int s = 0; s < 3; ++s) lo[1 << s] = cconj(F.tw[(L << s) * ts]);
        hi[1] = cconj(F.tw[(L << 3) * ts]); hi[2] = cconj(F.tw[(L << 4) * ts]);
        lo[3] = cmul(lo[1], lo[2]); lo[5] = cmul(lo[1], lo[4]); lo[6] = cmul(lo[2], lo[4]); lo[7] = cmul(lo[3], lo[4]);
        hi[3] = cmul(hi[1], hi[2]);
#pragma unroll
        for (int b = 1; b < 32; ++b) {
            const double2 w = (b & 7) == 0 ? hi[b >> 3] : ((b >> 3) == 0 ? lo[b & 7] : cmul(lo[b & 7], hi[b >> 3]));
            x[kBrev5[b]] = cmul(x[kBrev5[b]], w);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    double2 u[32];
    {
        double *regd = reinterpret_cast<double *>(reg);
        const double *row = regd + bl * kWaveRow + h;
#pragma unroll
        for (int b = 0; b < 32; ++b) regd[b * kWaveRow + L] = x[kBrev5[b]].x;
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 32; ++i) u[i].x = row[2 * i];
        wave_lds_fence();
#pragma unroll
        for (int b = 0; b < 32; ++b) regd[b * kWaveRow + L] = x[kBrev5[b]].y;
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 32; ++i) u[i].y = row[2 * i];
        wave_lds_fence();
    }
    __builtin_amdgcn_sched_barrier(0);
    dft32<false>(u);
    __builtin_amdgcn_sched_barrier(0);
    {
        const double hd = (double)h, sg = 1.0 - 2.0 * hd;
#pragma unroll
        for (int a = 0; a < 32; ++a) {
            const double2 own = u[kBrev5[a]];
            const double2 f = make_double2(fma(hd, kC64[a] - 1.0, 1.0), hd * kS64[a]);
            const double2 t = (a == 0) ? own : cmul(own, f);
            const double2 r = make_double2(dpp_xor1(t.x), dpp_xor1(t.y));
            u[kBrev5[a]] = make_double2(fma(sg, t.x, r.x), fma(sg, t.y, r.y));
        }
    }
    // ---- orders: lane (b, h = 0), register a' owns V_m, m = 4 k1 + k2, k1 = 32 a' + b < q / 2; V_(n - m) = bin k1m of sub-DFT (4 - k2) mod 4,
    // k1m = q - 1 - k1 (k2 = 0: q - k1), in the upper half: published by the lanes h = 1 of that wave at entry k1m - q / 2 of its strip
    __builtin_amdgcn_sched_barrier(0);
    if (h == 1) {
#pragma unroll
        for (int a = 0; a < 32; ++a) reg[32 * a + bl] = u[kBrev5[a]];
    }
    __syncthreads();
    {
        const double2 *mir = lds + ((4 - k2) & 3) * kWaveReg;
        double2 p0 = make_double2(1., 0.), p1 = p0;
        if (shifted) {
            p0 = cispi((4.0 * bl + k2) * inv_n);                                         // e^{i pi (4 b + k2) / n}; step per a': e^{i pi 128 / n} = e^{2 pi i / 128}
            p1 = cmul(p0, make_double2(0.9987954562051724, 0.049067674327418015));   // x e^{2 pi i / 128}
        }
#pragma unroll
        for (int a = 0; a < 32; ++a) {
            if (128 * a + k2 > ml) break;  // (wave-uniform: the smallest order of this register, at b = 0, is already out of band)
            const int k1 = 32 * a + bl, m = 4 * k1 + k2;
            if (h == 0 && m <= ml) {
                const double2 own = u[kBrev5[a]];
                const int k1m = k2 == 0 ? 1024 - k1 : 1023 - k1;          // k1m - q / 2 (k2 = 0, k1 = 0: the order 0 mirrors into itself)
                const double2 vm = (k2 == 0 && k1 == 0) ? own : mir[k1m];
                const double2 av = cconj(own);
                double2 fn = cadd(av, vm);
                const double2 dd = csub(av, vm);
                double2 fs = make_double2(dd.y, -dd.x);   // (conj(V_m) - V_(n-m)) / i
                if (shifted) { const double2 pk = rot64<false>((a & 1) ? p1 : p0, a >> 1); fn = cmulc(fn, pk); fs = cmulc(fs, pk); }  // e^{-i pi m / n}
                double4 o;
                o.x = fn.x * wgt; o.y = fn.y * wgt;
                o.z = has_s ? fs.x * wgt : 0.0; o.w = has_s ? fs.y * wgt : 0.0;
                *reinterpret_cast<double4 *>(ph + (int64_t)m * estride) = o;
            }
        }
        // the order n / 2 of a ring with mlim = n / 2: bin q / 2 of sub-DFT 0 (lane b = 0, h = 1, register a' = 0) mirrors into itself
        if (k2 == 0 && L == 1 && 2 * q <= ml) {
            const double2 own = u[kBrev5[0]];
            const double2 av = cconj(own);
            double2 fn = cadd(av, own);
            const double2 dd = csub(av, own);
            double2 fs = make_double2(dd.y, -dd.x);
            if (shifted) { const double2 pk = cispi(2.0 * q * inv_n); fn = cmulc(fn, pk); fs = cmulc(fs, pk); }
            double4 o;
            o.x = fn.x * wgt; o.y = fn.y * wgt;
            o.z = has_s ? fs.x * wgt : 0.0; o.w = has_s ? fs.y * wgt : 0.0;
            *reinterpret_cast<double4 *>(ph + (int64_t)(2 * q) * estride) = o;
        }
    }
}

// =====================================================================================================
// "Quad" variants of the register-resident kernels: one workgroup of 4 G = N / 2 threads per ring pair and component, thread group
// k2 = threadIdx / G owning sub-DFT k2 -- 8 points per thread instead of 4 x 8.  The four sub-DFTs of a ring then go through their
// LDS exchanges at the same time (4 exchange buffers, 6 instead of 24 barrier pairs per ring), every gather or store round is
// one set of 8 accesses per thread with four times as many threads in flight, and at ~100 registers a CU holds 16 waves instead of 8.
// The radix-4 step on the pixel side (which needs all four sub-DFTs of a pixel) goes through the exchange buffers once more.
// Same arithmetic per sub-DFT as k_phase2map_fast / k_map2phase_fast (fft8, fast_bin, the chirp / filter tables): N <= 2048
// (N = 4096 would need 2048 threads and 256 KB of LDS).
template <int N, bool BLUE, bool WGT = false, bool SPLIT = false>
__global__ __launch_bounds__(N / 2) void k_phase2map_quad(DevPlan P, DevFFT F, const int *__restrict__ pairs, const int *__restrict__ mlim,
                                                          int ncomp, const double *__restrict__ phase, double *__restrict__ map,
                                                          const double *__restrict__ wgt)
{
    extern __shared__ double2 lds_all[];
    constexpr int G = N / 8;
    const int k2 = threadIdx.x / G, tl0 = threadIdx.x % G;
    double2 *lds = lds_all + k2 * N;
    const int ip = pairs[blockIdx.x], comp = blockIdx.y;
    const int n = P.nphi[ip], q = n >> 2;
    constexpr bool blue = BLUE;
    const int K = F.K2of[q];
    const double2 *__restrict__ chirp = F.chirp + F.woff[q];
    const double2 *__restrict__ filt = F.A.filt + F.A.coff[q];
    const int ml = min(mlim[ip], P.mmax);
    const bool shifted = P.phi0[ip] != 0.0;
    const double inv_n = 1.0 / n;
    constexpr int estride = 4;
    const double *__restrict__ ph = phase + ((int64_t)ip * ncomp + comp) * P.mstride * estride;
    Tw8<N> tw;
    tw8_load<N>(tw, tl0, F.tw, F.Mtw);
    double2 d[8];
    {   // gather (see k_phase2map_fast): bins 4 k1 + k2 of this group's sub-DFT.  All eight loads of a thread are issued before the first
        // one is used (one round trip to memory per ring instead of eight)
        const int tl = fresh(tl0);
        double4 f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const FastBin b = fast_bin(blue, tl + G * j, N, q, K);
            const int k = 4 * b.k1 + k2;
            const int mm = b.sgn > 0 ? k : n - k;
            const bool have = b.sgn != 0 && mm <= ml;
            f[j] = *reinterpret_cast<const double4 *>(ph + (int64_t)(have ? mm : 0) * estride);  // out-of-band bins: entry 0, zeroed below
        }
        double2 pj = make_double2(1., 0.), pstep = pj, uneg = pj;
        if (shifted) {
            pj = cispi((4.0 * tl + k2) * inv_n); pstep = F.ringc[4 * q + 2];
            uneg = F.ringc[4 * q + 3];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const FastBin b = fast_bin(blue, tl + G * j, N, q, K);
            double2 cw = make_double2(1., 0.);
            if (blue) cw = chirp[b.cabs];
            const int k = 4 * b.k1 + k2;
            const int mm = b.sgn > 0 ? k : n - k;
            const bool have = b.sgn != 0 && mm <= ml;
            double2 fn = make_double2(f[j].x, f[j].y), fs = make_double2(f[j].z, f[j].w);
            if (shifted) {
                const double2 pm = b.sgn > 0 ? pj : cmulc(uneg, pj);
                fn = cmul(fn, pm); fs = cmul(fs, pm);
            }
            double2 z;
            z.x = have ? (b.sgn > 0 ? fn.x - fs.y : fn.x + fs.y) : 0.0;
            z.y = have ? (b.sgn > 0 ? fn.y + fs.x : -fn.y + fs.x) : 0.0;
            if (!blue && j == 4) {  // order n / 2 of a direct ring with mlim = n / 2 (see k_phase2map_fast)
                const bool nyq = have && tl == 0 && k2 == 0;
                z.x = nyq ? 2.0 * fn.x : z.x;
                z.y = nyq ? 2.0 * fs.x : z.y;
            }
            d[j] = cmul(z, cw);
            pj = cmul(pj, pstep);
        }
    }
    if constexpr (SPLIT) {
        fft8<N, true>(d, lds, tl0, tw);
        double2 spec[8], half[4];
        {
            const int tl = fresh(tl0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { spec[j] = d[j]; d[j] = cmul(d[j], filt[tl + G * j]); }
        }
        fft8<N, false>(d, lds, tl0, tw);
        {
            const int tl = fresh(tl0);
#pragma unroll
            for (int j = 0; j < 4; ++j) half[j] = d[j];  // pixels j1 < N / 2
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = cmul(spec[j], filt[N + tl + G * j]);
        }
        fft8<N, false>(d, lds, tl0, tw);
#pragma unroll
        for (int j = 0; j < 4; ++j) { d[j + 4] = d[j]; d[j] = half[j]; }
    } else if (blue) {
        fft8<N, true>(d, lds, tl0, tw);
        const int tl = fresh(tl0);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = cmul(d[j], filt[tl + G * j]);
        fft8<N, false>(d, lds, tl0, tw);
    } else {
        fft8<N, false>(d, lds, tl0, tw);
    }
    // pixel side: y_k2(j1) = d(j1) chirp e^{2 pi i j1 k2 / n} into the exchange buffers, then thread group g combines the four sub-DFTs
    // for the pixels j1 = tl + G j, j = 2 g, 2 g + 1
    const int tl = fresh(tl0);
    {
        double2 ek = cispi(2.0 * k2 * tl * inv_n);
        const double2 eg = F.ringc[4 * q + 1], eg2 = cmul(eg, eg);  // e^{2 pi i G / n}; this group's step is its k2-th power
        const double2 estep = k2 == 0 ? make_double2(1., 0.) : k2 == 1 ? eg : k2 == 2 ? eg2 : cmul(eg2, eg);
        __syncthreads();  // every group is done with its exchange buffer
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int j1 = tl + G * j;
            double2 cw = ek;
            if (blue) cw = cmul(ek, chirp[min(j1, q - 1)]);
            lds_all[k2 * N + j1] = cmul(d[j], cw);
            ek = cmul(ek, estep);
        }
        __syncthreads();
    }
    double *__restrict__ mp = map + (int64_t)comp * P.npix;
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j1 = tl + G * (2 * k2 + u);
        if (j1 < q) {
            double2 y[4] = {lds_all[j1], lds_all[N + j1], lds_all[2 * N + j1], lds_all[3 * N + j1]};
            dft_small<4, false>(y);  // y[j2] = sum_k2 i^(j2 k2) y_k2
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                if constexpr (WGT) {
                    mp[on + j1 + q * j2] = y[j2].x * wgt[on + j1 + q * j2];
                    if (os >= 0) mp[os + j1 + q * j2] = y[j2].y * wgt[os + j1 + q * j2];
                } else {
                    mp[on + j1 + q * j2] = y[j2].x;
                    if (os >= 0) mp[os + j1 + q * j2] = y[j2].y;
                }
            }
        }
    }
}

template <int N, bool BLUE, bool SPLIT = false>
__global__ __launch_bounds__(N / 2) void k_map2phase_quad(DevPlan P, DevFFT F, const int *__restrict__ pairs, const int *__restrict__ mlim,
                                                          int ncomp, const double *map, double *phase)
{
    extern __shared__ double2 lds_all[];
    constexpr int G = N / 8;
    const int k2 = threadIdx.x / G, tl0 = threadIdx.x % G;
    double2 *lds = lds_all + k2 * N;
    const int ip = pairs[blockIdx.x], comp = blockIdx.y;
    const int n = P.nphi[ip], q = n >> 2;
    constexpr bool blue = BLUE;
    const int K = F.K2of[q];
    const double2 *__restrict__ chirp = F.chirp + F.woff[q];
    const double2 *__restrict__ filt = F.A.filt + F.A.coff[q];
    const int ml = min(mlim[ip], P.mmax);
    const bool shifted = P.phi0[ip] != 0.0;
    const double inv_n = 1.0 / n;
    constexpr int estride = 4;
    double *ph = phase + ((int64_t)ip * ncomp + comp) * P.mstride * estride;
    const double wgt = 0.5 * 4.0 * 3.14159265358979323846 / (double)P.npix;  // includes the 1/2 of the N/S split
    const double *mp = fft_input_map(F, P, map, comp);
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    const bool has_s = os >= 0;
    Tw8<N> tw;
    tw8_load<N>(tw, tl0, F.tw, F.Mtw);
    {   // pixels in: thread group g loads the pixels j1 = tl + G j, j = 2 g, 2 g + 1 (all four j2, both rings), does the radix-4 step and
        // hands y_k2(j1) e^{2 pi i j1 k2 / n} chirp to sub-DFT k2's buffer; slots beyond the ring are zero
        const int tl = fresh(tl0);
        const double *mps = has_s ? mp + os : mp + on;
        double vn[2][4], vs[2][4];
        const double2 e1a = cispi(2.0 * (tl + G * 2 * k2) * inv_n);
#pragma unroll
        for (int u = 0; u < 2; ++u) {   // all sixteen loads first
            const int j1c = min(tl + G * (2 * k2 + u), q - 1);
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) { vn[u][j2] = mp[on + j1c + q * j2]; vs[u][j2] = mps[j1c + q * j2]; }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j1 = tl + G * (2 * k2 + u);
            const int j1c = min(j1, q - 1);
            double2 y[4];
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                y[j2].x = j1 < q ? vn[u][j2] : 0.0;
                y[j2].y = (j1 < q && has_s) ? -vs[u][j2] : 0.0;
            }
            dft_small<4, false>(y);  // y[k2] = sum_j2 i^(j2 k2) conj(z)_(j1 + q j2)
            double2 cw = make_double2(1., 0.);
            if (blue) cw = chirp[j1c];
            const double2 e1 = u == 0 ? e1a : cmul(e1a, F.ringc[4 * q + 1]), e2 = cmul(e1, e1), e3 = cmul(e2, e1);  // e^{2 pi i j1 / n}
            lds_all[j1] = cmul(y[0], cw);
            lds_all[N + j1] = cmul(y[1], cmul(cw, e1));
            lds_all[2 * N + j1] = cmul(y[2], cmul(cw, e2));
            lds_all[3 * N + j1] = cmul(y[3], cmul(cw, e3));
        }
        __syncthreads();
    }
    double2 d[8];
    {
        const int tl = fresh(tl0);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = lds[tl + G * j];
    }
    if constexpr (SPLIT) {
        double2 hi[8] = {d[4], d[5], d[6], d[7], make_double2(0., 0.), make_double2(0., 0.), make_double2(0., 0.), make_double2(0., 0.)};
#pragma unroll
        for (int j = 4; j < 8; ++j) d[j] = make_double2(0., 0.);
        fft8<N, true>(d, lds, tl0, tw);
        phase_fence();
        {
            const int tl = fresh(tl0);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = cmul(d[j], filt[(N - (tl + G * j)) & (N - 1)]);
        }
        fft8<N, true>(hi, lds, tl0, tw);
        phase_fence();
        {
            const int tl = fresh(tl0);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = cadd(d[j], cmul(hi[j], filt[N + ((N - (tl + G * j)) & (N - 1))]));
        }
        fft8<N, false>(d, lds, tl0, tw);
        phase_fence();
        {
            const int tl = fresh(tl0);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = cmul(d[j], chirp[fast_bin(blue, tl + G * j, N, q, K).cabs]);
        }
    } else if (blue) {
        fft8<N, true>(d, lds, tl0, tw);
        phase_fence();
        {
            const int tl = fresh(tl0);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = cmul(d[j], filt[(N - (tl + G * j)) & (N - 1)]);  // spectrum of the mirrored filter
        }
        fft8<N, false>(d, lds, tl0, tw);
        phase_fence();
        {
            const int tl = fresh(tl0);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = cmul(d[j], chirp[fast_bin(blue, tl + G * j, N, q, K).cabs]);
        }
    } else {
        fft8<N, false>(d, lds, tl0, tw);
    }
    // F_N, F_S of order m = 4 k1 + k2 need V_m (own) and V_(n - m), which lives in sub-DFT (4 - k2) mod 4: every group publishes its
    // - side values (and bin 0) in its own buffer, indexed by k1, and reads its partner's
    const int tl = fresh(tl0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const FastBin b = fast_bin(blue, tl + G * j, N, q, K);
        if (b.sgn < 0 || (b.sgn > 0 && b.k1 == 0)) lds[swz(b.k1)] = d[j];
    }
    __syncthreads();
    const double2 *other = lds_all + ((4 - k2) & 3) * N;
    double2 pjj = make_double2(1., 0.), pstep = pjj;
    if (shifted) { pjj = cispi((4.0 * tl + k2) * inv_n); pstep = F.ringc[4 * q + 2]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const FastBin b = fast_bin(blue, tl + G * j, N, q, K);
        const int m = 4 * b.k1 + k2;
        const bool nyq = !blue && k2 == 0 && j == 4 && tl == 0;
        if ((b.sgn > 0 || nyq) && m <= ml) {
            const int k1m = k2 == 0 ? (b.k1 == 0 ? 0 : q - b.k1) : q - 1 - b.k1;
            const double2 a = cconj(d[j]);
            const double2 vm = other[swz(k1m)];
            double2 fn = cadd(a, vm);
            const double2 dd = csub(a, vm);
            double2 fs = make_double2(dd.y, -dd.x);   // (conj(V_m) - V_(n-m)) / i
            if (shifted) { fn = cmulc(fn, pjj); fs = cmulc(fs, pjj); }  // e^{-i pi m / n}
            double4 o;
            o.x = fn.x * wgt; o.y = fn.y * wgt;
            o.z = has_s ? fs.x * wgt : 0.0; o.w = has_s ? fs.y * wgt : 0.0;
            *reinterpret_cast<double4 *>(ph + (int64_t)m * estride) = o;
        }
        pjj = cmul(pjj, pstep);
    }
}

// plan-time: natural-order spectrum (times 1/M) of the wrapped conjugate chirp h_d = e^{-i pi d^2 / q}, d in [-K, q-1+K],
// K = K2of[q] (band-limited sub-DFT bins)
template <int NT>
__global__ __launch_bounds__(NT) void k_bluestein_setup2(DevFFT F, const int *__restrict__ qlist, double2 *__restrict__ filt_out)
{
    extern __shared__ double2 ws[];
    const int q = qlist[blockIdx.x];
    const FftSide &sd = F.A;
    const int M = sd.Mof[q];
    const int Kn = F.K2of[q], Kp = F.K2of[q];
    // split rings: two spectra, of h_d for d in [-K, M / 2 - 1 + K] (first half of the pixels) and of h_(d + M / 2) for
    // d in [-K, q - 1 - M / 2 + K] (the others)
    const int nparts = sd.split[q] ? 2 : 1;
    for (int part = 0; part < nparts; ++part) {
        double2 *filt = filt_out + sd.coff[q] + (int64_t)part * M;
        const int shift = part * (M / 2);
        const int len = sd.split[q] ? (part == 0 ? M / 2 : q - M / 2) : q;  // pixels served by this spectrum
        __syncthreads();
        for (int t = threadIdx.x; t < M; t += NT) ws[t] = make_double2(0., 0.);
        __syncthreads();
        for (int t = threadIdx.x; t < len + Kn + Kp; t += NT) {
            const int dd = t - Kn;  // -Kn .. len - 1 + Kp
            const long long da = dd + shift;
            const long long t2 = (da * da) % (2LL * q);
            const double2 w = cispi((double)t2 / (double)q);
            ws[(dd + M) & (M - 1)] = cconj(w);
        }
        fft_dif_fwd<NT>(ws, M, nullptr, F.tw, F.Mtw);
        const double inv = 1.0 / M;
        for (int t = threadIdx.x; t < M; t += NT) {
            const double2 v = ws[digit_reverse(t, M)];
            filt[t] = make_double2(v.x * inv, v.y * inv);
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// host launchers
// -----------------------------------------------------------------------------------------------------
constexpr int kMaxDevices = 64;
static int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
    return d;
}

// dynamic LDS ceiling of the generic kernels: the CU's 160 KiB less their static allocations (k_map2phase keeps the template coefficients there)
static constexpr int kGenericMaxLds = 160 * 1024 - 256;
// largest LDS footprint for which the generic kernel transforms the four sub-DFTs of a ring side by side
static constexpr size_t kB4MaxLds = (size_t)kGenericMaxLds;
static size_t fft_lds_bytes(const DevFFT &F) { return (size_t)(F.Lmax + F.twl_cap) * sizeof(double2); }

template <int NT, int QMAX>
static hipError_t launch_p2m(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, const double *phase, double *map,
                             hipStream_t st, const NinvProj &W)
{
    if (F.A.legacy_n == 0) return hipSuccess;
    const size_t lds4 = (size_t)(4 * F.Lmax + F.twl_cap) * sizeof(double2);
    if (lds4 <= kB4MaxLds) {  // short transforms (coarse grids, short cap rings): the four sub-DFTs side by side
        static bool attr4_done[kMaxDevices] = {};
        const int dv4 = current_device();
        if (!attr4_done[dv4] && lds4 > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map<NT, QMAX, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kGenericMaxLds);
            if (e != hipSuccess) return e;
            attr4_done[dv4] = true;
        }
        hipLaunchKernelGGL((k_phase2map<NT, QMAX, true>), dim3(F.A.legacy_n, ncomp), dim3(NT), lds4, st, P, F, F.A.legacy_pairs, mlim, ncomp,
                           phase, map, W);
        return hipGetLastError();
    }
    const size_t lds = fft_lds_bytes(F);
    static bool attr_done[kMaxDevices] = {};  // function attributes are per device
    const int dv = current_device();
    if (!attr_done[dv] && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map<NT, QMAX, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kGenericMaxLds);
        if (e != hipSuccess) return e;
        attr_done[dv] = true;
    }
    hipLaunchKernelGGL((k_phase2map<NT, QMAX, false>), dim3(F.A.legacy_n, ncomp), dim3(NT), lds, st, P, F, F.A.legacy_pairs, mlim, ncomp, phase,
                       map, W);
    return hipGetLastError();
}

template <int NT, int QMAX>
static hipError_t launch_m2p(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, const double *map, double *phase,
                             hipStream_t st, const NinvProj &W)
{
    if (F.A.legacy_n == 0) return hipSuccess;
    const size_t lds4 = (size_t)(4 * F.Lmax + F.twl_cap) * sizeof(double2);
    if (lds4 <= kB4MaxLds) {  // short transforms (coarse grids, short cap rings): the four sub-DFTs side by side
        static bool attr4_done[kMaxDevices] = {};
        const int dv4 = current_device();
        if (!attr4_done[dv4] && lds4 > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_map2phase<NT, QMAX, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kGenericMaxLds);
            if (e != hipSuccess) return e;
            attr4_done[dv4] = true;
        }
        hipLaunchKernelGGL((k_map2phase<NT, QMAX, true>), dim3(F.A.legacy_n, ncomp), dim3(NT), lds4, st, P, F, F.A.legacy_pairs, mlim, ncomp,
                           map, phase, W);
        return hipGetLastError();
    }
    const size_t lds = fft_lds_bytes(F);
    static bool attr_done[kMaxDevices] = {};
    const int dv = current_device();
    if (!attr_done[dv] && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_map2phase<NT, QMAX, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kGenericMaxLds);
        if (e != hipSuccess) return e;
        attr_done[dv] = true;
    }
    hipLaunchKernelGGL((k_map2phase<NT, QMAX, false>), dim3(F.A.legacy_n, ncomp), dim3(NT), lds, st, P, F, F.A.legacy_pairs, mlim, ncomp, map,
                       phase, W);
    return hipGetLastError();
}

#define PL_FFT_DISPATCH(FN, ...)                                            \
    do {                                                                    \
        const int ns = F.A.legacy_qmax;  /* workgroup threads x points per thread cover the longest quarter-ring of the list */ \
        if (ns <= 256) return FN<256, 1>(__VA_ARGS__);                      \
        if (ns <= 512) return FN<256, 2>(__VA_ARGS__);                      \
        if (ns <= 1024) return FN<256, 4>(__VA_ARGS__);                     \
        if (ns <= 2048) return FN<512, 4>(__VA_ARGS__);                     \
        if (ns <= 4096) return FN<1024, 4>(__VA_ARGS__);                    \
        if (ns <= 8192) return FN<1024, 8>(__VA_ARGS__);                    \
        return hipErrorInvalidValue;                                        \
    } while (0)

template <int NT, int QMAX>
static hipError_t launch_rt(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, double *phase, hipStream_t st, const double *n_inv)
{
    if (F.A.legacy_n == 0) return hipSuccess;
    const size_t lds4 = (size_t)(4 * F.Lmax + F.twl_cap) * sizeof(double2);
    if (lds4 <= kB4MaxLds) {
        static bool attr4_done[kMaxDevices] = {};
        const int dv4 = current_device();
        if (!attr4_done[dv4] && lds4 > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ring_roundtrip<NT, QMAX, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kGenericMaxLds);
            if (e != hipSuccess) return e;
            attr4_done[dv4] = true;
        }
        hipLaunchKernelGGL((k_ring_roundtrip<NT, QMAX, true>), dim3(F.A.legacy_n, ncomp), dim3(NT), lds4, st, P, F, F.A.legacy_pairs, mlim, ncomp,
                           phase, n_inv);
        return hipGetLastError();
    }
    const size_t lds = fft_lds_bytes(F);
    static bool attr_done[kMaxDevices] = {};
    const int dv = current_device();
    if (!attr_done[dv] && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ring_roundtrip<NT, QMAX, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kGenericMaxLds);
        if (e != hipSuccess) return e;
        attr_done[dv] = true;
    }
    hipLaunchKernelGGL((k_ring_roundtrip<NT, QMAX, false>), dim3(F.A.legacy_n, ncomp), dim3(NT), lds, st, P, F, F.A.legacy_pairs, mlim, ncomp,
                       phase, n_inv);
    return hipGetLastError();
}

// phase -> pixels -> n_inv x pixels -> phase in one launch, in place; only for plans whose rings all run in the generic kernel
hipError_t launch_ring_roundtrip(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, double *phase, const double *n_inv, hipStream_t st)
{
    PL_FFT_DISPATCH(launch_rt, P, F, mlim, ncomp, phase, st, n_inv);
}

static hipError_t launch_phase2map_legacy(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, const double *phase, double *map,
                                          hipStream_t st, const NinvProj &W)
{
    PL_FFT_DISPATCH(launch_p2m, P, F, mlim, ncomp, phase, map, st, W);
}

static hipError_t launch_map2phase_legacy(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, const double *map, double *phase,
                                          hipStream_t st, const NinvProj &W)
{
    PL_FFT_DISPATCH(launch_m2p, P, F, mlim, ncomp, map, phase, st, W);
}

// Which classes run in the quad kernels (one thread group per sub-DFT).  Measured per kernel at nside = lmax = 2048, two components
// (tools/fft_kernel_stats.sh, profiles/round4_b_fft_kernel_stats.txt): N = 512 classes 2.0x faster, N = 1024 1.2-1.8x, N = 2048 analysis
// of the direct rings 1.07x; the N = 2048 syntheses and Bluestein analyses are equal or slower (one 1024-thread workgroup per CU:
// no second workgroup to overlap its memory phases with, and the per-thread phase-factor set-up is paid by four times the threads).
// (PLSHTS_DEBUG=1 PLSHTS_FFT_QUAD=0: the one-group kernels everywhere, for the per-kernel comparison of tools/fft_kernel_stats.sh.)
// In a synthesis stage of a grid that has N = 2048 classes the quad kernels of the small classes (512- and 1024-thread workgroups with
// 64-128 KB of LDS) get in the way of the dominant one-group kernel running beside them (stage 0.66 -> 0.70 ms): syntheses use them
// on grids up to nside 1024 only (the coarse and middle levels of the CG chains), analyses everywhere.
template <int N, bool BLUE, bool SPLIT>
static bool fft_quad_enabled(bool synth, int nside)
{
    static const int mode = dbg_env_int("PLSHTS_FFT_QUAD", 1);
    if (mode == 0 || N < 512 || N > 2048) return false;
    if (synth) return nside <= 1024 && N <= 1024;
    return N <= 1024 || !BLUE;  // N = 2048: the analysis of the direct (equatorial) rings only
}

template <int N, bool BLUE, bool SPLIT>
static hipError_t launch_quad_class(const DevPlan &P, const DevFFT &F, int n, const int *pairs, bool synth, const int *mlim, int ncomp, const double *in,
                                    double *out, hipStream_t st, const double *wgt)
{
    constexpr int lds = 4 * N * (int)sizeof(double2);
    if (lds > 48 * 1024) {
        static bool attr_done[kMaxDevices] = {};
        const int dv = current_device();
        if (!attr_done[dv]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map_quad<N, BLUE, false, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map_quad<N, BLUE, true, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_map2phase_quad<N, BLUE, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
            attr_done[dv] = true;
        }
    }
    if (synth && wgt) hipLaunchKernelGGL((k_phase2map_quad<N, BLUE, true, SPLIT>), dim3(n, ncomp), dim3(N / 2), lds, st, P, F, pairs, mlim, ncomp, in, out, wgt);
    else if (synth) hipLaunchKernelGGL((k_phase2map_quad<N, BLUE, false, SPLIT>), dim3(n, ncomp), dim3(N / 2), lds, st, P, F, pairs, mlim, ncomp, in, out, wgt);
    else hipLaunchKernelGGL((k_map2phase_quad<N, BLUE, SPLIT>), dim3(n, ncomp), dim3(N / 2), lds, st, P, F, pairs, mlim, ncomp, in, out);
    return hipGetLastError();
}

template <int N, bool BLUE>
static hipError_t launch_fast_class(const DevPlan &P, const DevFFT &F, int cls, bool synth, const int *mlim, int ncomp, const double *in,
                                    double *out, hipStream_t st, const double *wgt)
{
    const FftSide &sd = F.A;  // both directions use the band-limited classes
    const int n = BLUE ? sd.cls_n[cls] : sd.dir_n[cls];
    const int *pairs = BLUE ? sd.cls_pairs[cls] : sd.dir_pairs[cls];
    if (n == 0) return hipSuccess;
    if constexpr (N == kWaveN && !BLUE) {
        // the direct rings of length 4 N: wavefront-private sub-DFTs (k_phase2map_wave); PLSHTS_DEBUG=1 PLSHTS_FFT_WAVE=0: the one-group kernel
        static const bool wave = dbg_env_int("PLSHTS_FFT_WAVE", 1) != 0;
        if (synth && wave) {
            static bool wattr_done[kMaxDevices] = {};
            const int dv = current_device();
            if (!wattr_done[dv]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map_wave<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kWaveLds);
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map_wave<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kWaveLds);
                if (e != hipSuccess) return e;
                wattr_done[dv] = true;
            }
            if (wgt) hipLaunchKernelGGL((k_phase2map_wave<true>), dim3(n, ncomp), dim3(256), kWaveLds, st, P, F, pairs, mlim, ncomp, in, out, wgt);
            else hipLaunchKernelGGL((k_phase2map_wave<false>), dim3(n, ncomp), dim3(256), kWaveLds, st, P, F, pairs, mlim, ncomp, in, out, wgt);
            return hipGetLastError();
        }
        if (!synth && wave) {
            static bool wattr_a_done[kMaxDevices] = {};
            const int dv = current_device();
            if (!wattr_a_done[dv]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_map2phase_wave), hipFuncAttributeMaxDynamicSharedMemorySize, kWaveLds);
                if (e != hipSuccess) return e;
                wattr_a_done[dv] = true;
            }
            hipLaunchKernelGGL(k_map2phase_wave, dim3(n, ncomp), dim3(256), kWaveLds, st, P, F, pairs, mlim, ncomp, in, out);
            return hipGetLastError();
        }
    }
    if constexpr (N <= 2048 && N >= 512) {
        if (fft_quad_enabled<N, BLUE, false>(synth, P.nside)) return launch_quad_class<N, BLUE, false>(P, F, n, pairs, synth, mlim, ncomp, in, out, st, wgt);
    }
    const size_t lds = (size_t)N * sizeof(double2);
    if (lds > 48 * 1024) {
        static bool attr_done[kMaxDevices] = {};
        const int dv = current_device();
        if (!attr_done[dv]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map_fast<N, BLUE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map_fast<N, BLUE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_map2phase_fast<N, BLUE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_done[dv] = true;
        }
    }
    if (synth && wgt) hipLaunchKernelGGL((k_phase2map_fast<N, BLUE, true>), dim3(n, ncomp), dim3(N / 8), lds, st, P, F, pairs, mlim, ncomp, in, out, wgt);
    else if (synth) hipLaunchKernelGGL((k_phase2map_fast<N, BLUE, false>), dim3(n, ncomp), dim3(N / 8), lds, st, P, F, pairs, mlim, ncomp, in, out, wgt);
    else hipLaunchKernelGGL((k_map2phase_fast<N, BLUE>), dim3(n, ncomp), dim3(N / 8), lds, st, P, F, pairs, mlim, ncomp, in, out);
    return hipGetLastError();
}

hipError_t fft_streams_create(FftStreams &fs)
{
    hipError_t e = hipEventCreateWithFlags(&fs.fork, hipEventDisableTiming);
    for (int i = 0; i < FftStreams::kN && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&fs.s[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&fs.join[i], hipEventDisableTiming);
    }
    fs.ok = e == hipSuccess;
    return e;
}

void fft_streams_destroy(FftStreams &fs)
{
    for (int i = 0; i < FftStreams::kN; ++i) {
        if (fs.s[i]) (void)hipStreamDestroy(fs.s[i]);
        if (fs.join[i]) (void)hipEventDestroy(fs.join[i]);
        fs.s[i] = nullptr; fs.join[i] = nullptr;
    }
    if (fs.fork) (void)hipEventDestroy(fs.fork);
    fs.fork = nullptr; fs.ok = false;
}

// One FFT stage = up to six independent kernels (five register classes + the generic kernel).  The biggest one runs on
// the caller's stream, the others on the plan's side streams between a fork and a join event.
template <int N>
static hipError_t launch_split_class(const DevPlan &P, const DevFFT &F, int cls, bool synth, const int *mlim, int ncomp, const double *in,
                                     double *out, hipStream_t st, const double *wgt)
{
    const int n = F.A.split_n[cls];
    if (n == 0) return hipSuccess;
    if constexpr (N <= 2048 && N >= 512) {
        if (fft_quad_enabled<N, true, true>(synth, P.nside)) return launch_quad_class<N, true, true>(P, F, n, F.A.split_pairs[cls], synth, mlim, ncomp, in, out, st, wgt);
    }
    const size_t lds = (size_t)N * sizeof(double2) * (synth ? 2 : 1);  // synthesis: exchange buffer + the parked forward spectrum
    static bool attr_done[kMaxDevices] = {};
    const int dv = current_device();
    if (!attr_done[dv]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map_fast<N, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * N * sizeof(double2)));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map_fast<N, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * N * sizeof(double2)));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_map2phase_fast<N, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(N * sizeof(double2)));
        if (e != hipSuccess) return e;
        attr_done[dv] = true;
    }
    const int *pairs = F.A.split_pairs[cls];
    if (synth && wgt) hipLaunchKernelGGL((k_phase2map_fast<N, true, true, true>), dim3(n, ncomp), dim3(N / 8), lds, st, P, F, pairs, mlim, ncomp, in, out, wgt);
    else if (synth) hipLaunchKernelGGL((k_phase2map_fast<N, true, false, true>), dim3(n, ncomp), dim3(N / 8), lds, st, P, F, pairs, mlim, ncomp, in, out, wgt);
    else hipLaunchKernelGGL((k_map2phase_fast<N, true, true>), dim3(n, ncomp), dim3(N / 8), lds, st, P, F, pairs, mlim, ncomp, in, out);
    return hipGetLastError();
}

static hipError_t launch_stage(const DevPlan &P, const DevFFT &F, const FftStreams &fs, bool synth, const int *mlim, int ncomp,
                               const double *in, double *out, hipStream_t st, const NinvProj &W)
{
    // PLSHTS_DEBUG=1 PLSHTS_FFT_SERIAL=1: every class on the caller's stream (profiling with PMC counters: one kernel at a time)
    static const bool serial = dbg_env_int("PLSHTS_FFT_SERIAL", 0) != 0;
    const bool par = fs.ok && !serial;
    // Work items: w = 3 c + kind of class c (kind 0: direct, 1: Bluestein, 2: split Bluestein) and the generic list (w = nw).  Cost
    // model: ring pairs x transform size x transforms per sub-DFT, plus a fixed latency (the short-ring kernels are latency-bound).
    // The costliest item stays on the caller's stream; the others go, costliest first, to the side stream with the least work
    // queued so far (longest-processing-time-first), so that no short kernel waits at the end of a queue and becomes the stage's tail.
    constexpr int nw = 3 * kFftClasses;
    auto count = [&](int w) { const int c = w / 3, k = w % 3; return w == nw ? F.A.legacy_n : (k == 0 ? F.A.dir_n[c] : (k == 1 ? F.A.cls_n[c] : F.A.split_n[c])); };
    int order[nw + 1], nitems = 0;
    int64_t cost[nw + 1];
    for (int w = 0; w <= nw; ++w) {
        cost[w] = 0;
        if (count(w) == 0) continue;
        cost[w] = w == nw ? (int64_t)F.A.legacy_n * F.A.legacy_qmax * 8 + (1 << 20)
                          : (int64_t)count(w) * (256 << (w / 3)) * (w % 3 + 1) + (1 << 19);
        order[nitems++] = w;
    }
    for (int a = 1; a < nitems; ++a)  // insertion sort, costliest first
        for (int b = a; b > 0 && cost[order[b]] > cost[order[b - 1]]; --b) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
    hipError_t e = hipSuccess;
    if (par && nitems > 1) e = hipEventRecord(fs.fork, st);
    bool joined[FftStreams::kN] = {false, false, false, false, false};
    int64_t load[FftStreams::kN] = {0, 0, 0, 0, 0};
    const double *wgt = synth ? W.n_inv : nullptr;
    auto run = [&](int w, hipStream_t s) -> hipError_t {
        switch (w) {
        case nw: return synth ? launch_phase2map_legacy(P, F, mlim, ncomp, in, out, s, W) : launch_map2phase_legacy(P, F, mlim, ncomp, in, out, s, W);
        case 14: return launch_split_class<4096>(P, F, 4, synth, mlim, ncomp, in, out, s, wgt);
        case 13: return launch_fast_class<4096, true>(P, F, 4, synth, mlim, ncomp, in, out, s, wgt);
        case 12: return launch_fast_class<4096, false>(P, F, 4, synth, mlim, ncomp, in, out, s, wgt);
        case 11: return launch_split_class<2048>(P, F, 3, synth, mlim, ncomp, in, out, s, wgt);
        case 10: return launch_fast_class<2048, true>(P, F, 3, synth, mlim, ncomp, in, out, s, wgt);
        case 9: return launch_fast_class<2048, false>(P, F, 3, synth, mlim, ncomp, in, out, s, wgt);
        case 8: return launch_split_class<1024>(P, F, 2, synth, mlim, ncomp, in, out, s, wgt);
        case 7: return launch_fast_class<1024, true>(P, F, 2, synth, mlim, ncomp, in, out, s, wgt);
        case 6: return launch_fast_class<1024, false>(P, F, 2, synth, mlim, ncomp, in, out, s, wgt);
        case 5: return launch_split_class<512>(P, F, 1, synth, mlim, ncomp, in, out, s, wgt);
        case 4: return launch_fast_class<512, true>(P, F, 1, synth, mlim, ncomp, in, out, s, wgt);
        case 3: return launch_fast_class<512, false>(P, F, 1, synth, mlim, ncomp, in, out, s, wgt);
        case 2: return launch_split_class<256>(P, F, 0, synth, mlim, ncomp, in, out, s, wgt);
        case 1: return launch_fast_class<256, true>(P, F, 0, synth, mlim, ncomp, in, out, s, wgt);
        default: return launch_fast_class<256, false>(P, F, 0, synth, mlim, ncomp, in, out, s, wgt);
        }
    };
    // items of a handful of ring pairs (the single power-of-two cap rings) are pure latency: they go first, one after the other, on
    // the last side stream, which takes nothing else -- at the end of a queue they were the tail of every stage (45-50 us)
    static const int nside_streams = [] { const int n = dbg_env_int("PLSHTS_FFT_STREAMS", 3); return n < 2 ? 2 : (n > FftStreams::kN ? FftStreams::kN : n); }();
    // (PLSHTS_DEBUG=1) PLSHTS_FFT_STREAMS: side streams in use, 2 ... 5.  Every one costs a join on the caller's stream (the idle gap at the end of a stage
    // grows by ~10 us per joined stream); measured 26.54 ms per reconstruction with 3, 26.64 with 5, 26.89 with 2
    const int kTinyStream = nside_streams - 1;
    // (round 5, measured and dropped: also sending every item under a tenth of the stage's work to that stream, so that the short kernels
    // run at the head of the stage beside the big ones instead of forming its ~70 us tail -- in series on one stream they became the
    // critical path instead: analysis stage 0.57 -> 0.62 ms, spin-0 synthesis 0.37 -> 0.44 ms)
    auto tiny = [&](int w) { return w != nw && count(w) < 8; };
    auto on_side = [&](int i, int w) -> hipStream_t {
        if (joined[i] || hipStreamWaitEvent(fs.s[i], fs.fork, 0) == hipSuccess) {
            joined[i] = true;
            load[i] += cost[w];
            return fs.s[i];
        }
        return st;
    };
    if (par && nitems > 1)
        for (int a = 1; a < nitems && e == hipSuccess; ++a)
            if (tiny(order[a])) e = run(order[a], on_side(kTinyStream, order[a]));
    // assignment: costliest first to the least-loaded side stream (LPT); launch order on a side stream: per `side_small_first` cheapest first,
    // so that a stream's short latency-bound kernels run at the head of the stage, beside the big classes, instead of alone at its end
    static const bool side_small_first = dbg_env_int("PLSHTS_FFT_SMALL_FIRST", 1) != 0;
    int assign[nw + 1];
    for (int a = 0; a < nitems; ++a) {
        const int w = order[a];
        assign[a] = -1;  // the caller's stream
        if (par && a > 0) {
            if (tiny(w)) { assign[a] = -2; continue; }  // launched above
            int best = 0;
            for (int i = 1; i < kTinyStream; ++i) if (load[i] < load[best]) best = i;
            assign[a] = best;
            load[best] += cost[w];
        }
    }
    if (nitems > 0 && e == hipSuccess) e = run(order[0], st);
    for (int i = 0; i < kTinyStream && e == hipSuccess; ++i) {
        for (int a0 = 1; a0 < nitems && e == hipSuccess; ++a0) {
            const int a = side_small_first ? nitems - a0 : a0;  // (order[] is sorted costliest first)
            if (assign[a] != i) continue;
            const int64_t keep = load[i];
            e = run(order[a], on_side(i, order[a]));
            load[i] = keep;  // (on_side adds the cost again: the loads were final after the assignment)
        }
    }
    for (int a = 1; a < nitems && e == hipSuccess; ++a)
        if (assign[a] == -1) e = run(order[a], st);  // (no side streams: everything on the caller's stream)
    for (int i = 0; i < FftStreams::kN; ++i) {
        if (!joined[i]) continue;
        hipError_t e2 = hipEventRecord(fs.join[i], fs.s[i]);
        if (e2 == hipSuccess) e2 = hipStreamWaitEvent(st, fs.join[i], 0);
        if (e == hipSuccess) e = e2;
    }
    return e;
}

bool fft_all_generic(const DevPlan &P, const DevFFT &F) { return F.A.legacy_n == P.npairs; }

hipError_t launch_phase2map(const DevPlan &P, const DevFFT &F, const FftStreams &fs, const int *mlim, int ncomp, const double *phase,
                            double *map, hipStream_t st, const NinvProj *W)
{
    // plain weighting (nmodes = 0) rides in every kernel and for any number of components; the template sums (one set per component:
    // W.parts holds ncomp x nmodes x nparts) need the generic kernel
    if (W && W->n_inv && W->nmodes > 0 && !(fft_all_generic(P, F) && W->nmodes <= kFuseModes && W->nparts >= F.A.legacy_n))
        return hipErrorInvalidValue;
    return launch_stage(P, F, fs, true, mlim, ncomp, phase, map, st, W ? *W : NinvProj());
}

hipError_t launch_map2phase(const DevPlan &P, const DevFFT &F, const FftStreams &fs, const int *mlim, int ncomp, const double *map,
                            double *phase, hipStream_t st, const NinvProj *W, const double *const *map_ind)
{
    if (W && W->rm && !(fft_all_generic(P, F) && W->nmodes <= kFuseModes && W->nparts >= F.A.legacy_n)) return hipErrorInvalidValue;
    if (map_ind) {  // inputs through a pointer table in device memory (one entry per component)
        if (W && (W->rm || W->n_inv)) return hipErrorInvalidValue;
        DevFFT Fi = F;
        Fi.map_ind = map_ind;
        return launch_stage(P, Fi, fs, false, mlim, ncomp, map, phase, st, NinvProj());
    }
    return launch_stage(P, F, fs, false, mlim, ncomp, map, phase, st, W ? *W : NinvProj());
}

hipError_t launch_twiddles(double *tw, int Mtw, hipStream_t st)
{
    hipLaunchKernelGGL(k_twiddles, dim3((Mtw / 2 + 255) / 256), dim3(256), 0, st, reinterpret_cast<double2 *>(tw), Mtw);
    return hipGetLastError();
}

hipError_t launch_bluestein_setup(const DevFFT &F, const int *qlist_dev, int nq, double *chirp, double *filt, hipStream_t st)
{
    if (nq == 0) return hipSuccess;
    const size_t lds = fft_lds_bytes(F);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bluestein_setup<256>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kGenericMaxLds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_bluestein_setup<256>, dim3(nq), dim3(256), lds, st, F, qlist_dev, reinterpret_cast<double2 *>(chirp),
                       reinterpret_cast<double2 *>(filt));
    return hipGetLastError();
}

hipError_t launch_bluestein_setup2(const DevFFT &F, const int *qlist_dev, int nq, int Mmax, double *filt2, hipStream_t st)
{
    if (nq == 0) return hipSuccess;
    const size_t lds = (size_t)Mmax * sizeof(double2);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bluestein_setup2<256>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kGenericMaxLds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_bluestein_setup2<256>, dim3(nq), dim3(256), lds, st, F, qlist_dev, reinterpret_cast<double2 *>(filt2));
    return hipGetLastError();
}

}  // namespace plshts
