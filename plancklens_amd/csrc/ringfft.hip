// Fourier stage of the HEALPix SHTs for gfx950: per-ring Fourier coefficients F_m(ring) <-> RING pixels.
//
// One workgroup per ring pair (north ring + mirror south ring, identical nphi and phi0).  The two real
// rings are transformed by ONE complex DFT of length n = nphi = 4 q (z = north + i south).  That DFT is
// split as 4 sub-DFTs of length q (Cooley-Tukey 4 x q) done in LDS:
//   * q a power of two (all equatorial rings when nside is): radix-2 DIT on bit-reversed input;
//   * otherwise (polar caps, q = ring number): Bluestein chirp-z with a power-of-two LDS convolution of size
//     M >= 2q - 1 whose filter spectrum is precomputed per q at plan creation (bit-reversed order, so the
//     DIF forward / DIT inverse pair needs no reordering pass).
// Synthesis gathers the aliased spectrum bins straight from the phase array, keeps the 4 x q partial
// results in registers and finishes with the radix-4 butterfly while writing pixels; analysis is the exact
// transpose (radix-4 on register-resident pixels first, sub-DFT outputs un-aliased from LDS).
//
// Bound: LDS bandwidth / HBM (the stage does O(npix log n) flops on 8 npix + 32 (mmax+1) nrings bytes).
#include <hip/hip_runtime.h>

#include "device_plan.h"
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

// forward (e^{-}) decimation-in-frequency FFT, natural order in, bit-reversed order out
template <int NT>
__device__ void fft_dif_fwd(double2 *a, int M, const double2 *__restrict__ tw, int Mtw)
{
    for (int h = M >> 1; h >= 1; h >>= 1) {
        __syncthreads();
        const int tstep = Mtw / (2 * h);
        for (int b = threadIdx.x; b < (M >> 1); b += NT) {
            const int pos = b & (h - 1);
            const int i0 = ((b - pos) << 1) + pos, i1 = i0 + h;
            const double2 u = a[i0], v = a[i1];
            const double2 w = tw[pos * tstep];
            a[i0] = cadd(u, v);
            a[i1] = cmul(csub(u, v), w);
        }
    }
    __syncthreads();
}

// inverse (e^{+}, unnormalised) decimation-in-time FFT, bit-reversed order in, natural order out
template <int NT>
__device__ void fft_dit_inv(double2 *a, int M, const double2 *__restrict__ tw, int Mtw)
{
    for (int h = 1; h < M; h <<= 1) {
        __syncthreads();
        const int tstep = Mtw / (2 * h);
        for (int b = threadIdx.x; b < (M >> 1); b += NT) {
            const int pos = b & (h - 1);
            const int i0 = ((b - pos) << 1) + pos, i1 = i0 + h;
            const double2 u = a[i0];
            const double2 v = cmulc(a[i1], tw[pos * tstep]);
            a[i0] = cadd(u, v);
            a[i1] = csub(u, v);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }
__device__ __forceinline__ int bitrev(int k, int lq) { return lq ? (int)(__brev((unsigned)k) >> (32 - lq)) : 0; }

// In: ws[0..q) filled (Bluestein: x_k w_k, zero padded to M by the caller; direct: x at bit-reversed positions).
// Out: ws[j] (times chirp[j] for Bluestein, applied by the caller through sub_value) = sum_k x_k e^{+2 pi i jk/q}.
template <int NT>
__device__ __forceinline__ void sub_dft_inverse(double2 *ws, int q, int M, const double2 *__restrict__ filt, const DevFFT &F)
{
    if (M) {
        fft_dif_fwd<NT>(ws, M, F.tw, F.Mtw);
        for (int t = threadIdx.x; t < M; t += NT) ws[t] = cmul(ws[t], filt[t]);
        fft_dit_inv<NT>(ws, M, F.tw, F.Mtw);
    } else {
        fft_dit_inv<NT>(ws, q, F.tw, F.Mtw);
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
    const int M = F.Mof[q];
    double2 *chirp = chirp_out + F.woff[q];
    double2 *filt = filt_out + F.coff[q];
    for (int t = threadIdx.x; t < M; t += NT) ws[t] = make_double2(0., 0.);
    __syncthreads();
    for (int t = threadIdx.x; t < q; t += NT) {
        const long long t2 = ((long long)t * t) % (2LL * q);
        const double2 w = cispi((double)t2 / (double)q);  // e^{i pi t^2 / q}
        chirp[t] = w;
        const double2 c = cconj(w);
        ws[t] = c;
        if (t > 0) ws[M - t] = c;
    }
    fft_dif_fwd<NT>(ws, M, F.tw, F.Mtw);
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
// synthesis: phase -> pixels
// -----------------------------------------------------------------------------------------------------
template <int NT, int QMAX>
__global__ __launch_bounds__(NT) void k_phase2map(DevPlan P, DevFFT F, const int *__restrict__ mlim, int ncomp,
                                                  const double *__restrict__ phase, double *__restrict__ map)
{
    extern __shared__ double2 ws[];
    const int ip = P.npairs - 1 - blockIdx.x;  // largest rings first
    const int comp = blockIdx.y;
    const int n = P.nphi[ip], q = n >> 2;
    const int M = F.Mof[q];
    const double2 *__restrict__ chirp = F.chirp + F.woff[q];
    const double2 *__restrict__ filt = F.filt + F.coff[q];
    const int ml = min(mlim[ip], P.mmax);
    const bool shifted = P.phi0[ip] != 0.0;
    const double inv_n = 1.0 / n;
    const int estride = 4 * ncomp;
    const double *__restrict__ ph = phase + (int64_t)ip * P.mstride * estride + comp * 4;
    const int lq = ilog2(q);

    double2 acc[4][QMAX], e1[QMAX];
#pragma unroll
    for (int qq = 0; qq < QMAX; ++qq) {
        acc[0][qq] = acc[1][qq] = acc[2][qq] = acc[3][qq] = make_double2(0., 0.);
        const int j1 = threadIdx.x + NT * qq;
        e1[qq] = cispi(2.0 * j1 * inv_n);  // e^{2 pi i j1 / n}
    }

    for (int k2 = 0; k2 < 4; ++k2) {
        if (M) for (int t = q + threadIdx.x; t < M; t += NT) ws[t] = make_double2(0., 0.);
        for (int k1 = threadIdx.x; k1 < q; k1 += NT) {
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
            const double2 z = make_double2(zr, zi);
            if (M) ws[k1] = cmul(z, chirp[k1]);
            else ws[bitrev(k1, lq)] = z;
        }
        sub_dft_inverse<NT>(ws, q, M, filt, F);
#pragma unroll
        for (int qq = 0; qq < QMAX; ++qq) {
            const int j1 = threadIdx.x + NT * qq;
            if (j1 < q) {
                double2 y = ws[j1];
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
        __syncthreads();
    }
    double *__restrict__ mp = map + (int64_t)comp * P.npix;
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
#pragma unroll
        for (int qq = 0; qq < QMAX; ++qq) {
            const int j1 = threadIdx.x + NT * qq;
            if (j1 < q) {
                const int j = j1 + q * j2;
                mp[on + j] = acc[j2][qq].x;
                if (os >= 0) mp[os + j] = acc[j2][qq].y;
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------------
// analysis: pixels -> phase (uniform quadrature weights 4 pi / npix)
// -----------------------------------------------------------------------------------------------------
template <int NT, int QMAX>
__global__ __launch_bounds__(NT) void k_map2phase(DevPlan P, DevFFT F, const int *__restrict__ mlim, int ncomp,
                                                  const double *__restrict__ map, double *__restrict__ phase)
{
    extern __shared__ double2 ws[];
    const int ip = P.npairs - 1 - blockIdx.x;
    const int comp = blockIdx.y;
    const int n = P.nphi[ip], q = n >> 2;
    const int M = F.Mof[q];
    const double2 *__restrict__ chirp = F.chirp + F.woff[q];
    const double2 *__restrict__ filt = F.filt + F.coff[q];
    const int ml = min(mlim[ip], P.mmax);
    const bool shifted = P.phi0[ip] != 0.0;
    const double inv_n = 1.0 / n;
    const int estride = 4 * ncomp;
    double *__restrict__ ph = phase + (int64_t)ip * P.mstride * estride + comp * 4;
    const int lq = ilog2(q);
    const double wgt = 0.5 * 4.0 * 3.14159265358979323846 / (double)P.npix;  // includes the 1/2 of the N/S split
    const double *__restrict__ mp = map + (int64_t)comp * P.npix;
    const int64_t on = P.ofs_n[ip], os = P.ofs_s[ip];
    const bool has_s = os >= 0;

    // conj(z_j) = north - i south, j = j1 + q j2
    double2 zc[4][QMAX], e1[QMAX], hold[QMAX];
#pragma unroll
    for (int qq = 0; qq < QMAX; ++qq) {
        const int j1 = threadIdx.x + NT * qq;
        e1[qq] = cispi(2.0 * j1 * inv_n);
        hold[qq] = make_double2(0., 0.);
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
            double2 v = make_double2(0., 0.);
            if (j1 < q) {
                const int j = j1 + q * j2;
                v.x = mp[on + j];
                v.y = has_s ? -mp[os + j] : 0.0;
            }
            zc[j2][qq] = v;
        }
    }

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

    for (int kk = 0; kk < 4; ++kk) {
        const int k2 = (kk == 0) ? 0 : (kk == 1) ? 2 : (kk == 2) ? 1 : 3;
        if (M) for (int t = q + threadIdx.x; t < M; t += NT) ws[t] = make_double2(0., 0.);
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
                if (M) ws[j1] = cmul(x, chirp[j1]);
                else ws[bitrev(j1, lq)] = x;
            }
        }
        sub_dft_inverse<NT>(ws, q, M, filt, F);
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

// -----------------------------------------------------------------------------------------------------
// host launchers
// -----------------------------------------------------------------------------------------------------
static size_t fft_lds_bytes(const DevFFT &F) { return (size_t)F.Lmax * sizeof(double2); }

template <int NT, int QMAX>
static hipError_t launch_p2m(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, const double *phase, double *map,
                             hipStream_t st)
{
    const size_t lds = fft_lds_bytes(F);
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase2map<NT, QMAX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((k_phase2map<NT, QMAX>), dim3(P.npairs, ncomp), dim3(NT), lds, st, P, F, mlim, ncomp, phase, map);
    return hipGetLastError();
}

template <int NT, int QMAX>
static hipError_t launch_m2p(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, const double *map, double *phase,
                             hipStream_t st)
{
    const size_t lds = fft_lds_bytes(F);
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_map2phase<NT, QMAX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((k_map2phase<NT, QMAX>), dim3(P.npairs, ncomp), dim3(NT), lds, st, P, F, mlim, ncomp, map, phase);
    return hipGetLastError();
}

#define PL_FFT_DISPATCH(FN, ...)                                            \
    do {                                                                    \
        const int ns = P.nside;                                             \
        if (ns <= 256) return FN<256, 1>(__VA_ARGS__);                      \
        if (ns <= 512) return FN<256, 2>(__VA_ARGS__);                      \
        if (ns <= 1024) return FN<256, 4>(__VA_ARGS__);                     \
        if (ns <= 2048) return FN<512, 4>(__VA_ARGS__);                     \
        if (ns <= 4096) return FN<1024, 4>(__VA_ARGS__);                    \
        if (ns <= 8192) return FN<1024, 8>(__VA_ARGS__);                    \
        return hipErrorInvalidValue;                                        \
    } while (0)

hipError_t launch_phase2map(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, const double *phase, double *map,
                            hipStream_t st)
{
    PL_FFT_DISPATCH(launch_p2m, P, F, mlim, ncomp, phase, map, st);
}

hipError_t launch_map2phase(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, const double *map, double *phase,
                            hipStream_t st)
{
    PL_FFT_DISPATCH(launch_m2p, P, F, mlim, ncomp, map, phase, st);
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
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_bluestein_setup<256>, dim3(nq), dim3(256), lds, st, F, qlist_dev, reinterpret_cast<double2 *>(chirp),
                       reinterpret_cast<double2 *>(filt));
    return hipGetLastError();
}

}  // namespace plshts
