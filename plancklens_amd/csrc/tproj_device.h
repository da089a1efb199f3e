// Workgroup bodies of the template-projection coefficient pass (c_k = sum_i P_ki t_i, opfilt_tt.py:196-205), shared by the stand-alone
// kernels of elementwise.hip and by the prologue kernel of the temperature CG operator (legendre.hip: k_prep0_lr), which runs them in
// extra workgroups of its own launch.  One body = one workgroup of NT threads = one partial sum per mode: part `part` of `nparts`.
// Whoever launches them, every sum is formed from the same entries in the same order: bit-identical partial sums.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace plshts {

constexpr int kProjParts = 256, kProjMaxModes = 16;
constexpr int kFuseModesB = 4;  // block-vector bodies: monopole + dipole (more modes: one launch set per map)
constexpr int kProjChunk = 4;   // block vectors: maps per workgroup pass (the template rows are read once for all of them)

// one vector t of n doubles (n_inv: t <- n_inv t on the way; null: t is only read); parts: [kProjMaxModes][kProjParts] of this vector
template <int NT>
__device__ __forceinline__ void tproj_coeffs_wg(int64_t n, int nmodes, double *__restrict__ t, const double *__restrict__ n_inv,
                                                const double *__restrict__ pm, double *__restrict__ parts, int part, int nparts)
{
    __shared__ double red[kProjMaxModes][NT / 64];
    double acc[kProjMaxModes];
#pragma unroll
    for (int k = 0; k < kProjMaxModes; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)part * NT + threadIdx.x; i < n; i += (int64_t)nparts * NT) {
        double u = t[i];
        if (n_inv) { u *= n_inv[i]; t[i] = u; }  // n_inv null: t arrives weighted (the ring-FFT kernels did it), nothing to store
#pragma unroll
        for (int k = 0; k < kProjMaxModes; ++k)
            if (k < nmodes) acc[k] = fma(pm[(int64_t)k * n + i], u, acc[k]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kProjMaxModes; ++k) {
        if (k < nmodes) {  // wave-uniform
            double v = acc[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) red[k][wave] = v;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nmodes) {
        double v = 0.0;
        for (int w = 0; w < NT / 64; ++w) v += red[threadIdx.x][w];
        parts[threadIdx.x * kProjParts + part] = v;
    }
}

// maps b0 ... b0 + kProjChunk - 1 of a block of nb maps (t_: nb x n; parts_: nb x [kProjMaxModes][kProjParts]); nmodes <= kFuseModesB
template <int NT>
__device__ __forceinline__ void tproj_coeffs_b_wg(int64_t n, int nmodes, int nb, double *__restrict__ t_, const double *__restrict__ n_inv,
                                                  const double *__restrict__ pm, double *__restrict__ parts_, int part, int nparts, int chunk)
{
    __shared__ double red[kProjChunk][kFuseModesB][NT / 64];
    const int b0 = chunk * kProjChunk, nbc = min(kProjChunk, nb - b0);
    double acc[kProjChunk][kFuseModesB];
#pragma unroll
    for (int b = 0; b < kProjChunk; ++b)
#pragma unroll
        for (int k = 0; k < kFuseModesB; ++k) acc[b][k] = 0.0;
    for (int64_t i = (int64_t)part * NT + threadIdx.x; i < n; i += (int64_t)nparts * NT) {
        double p[kFuseModesB];
#pragma unroll
        for (int k = 0; k < kFuseModesB; ++k) p[k] = k < nmodes ? pm[(int64_t)k * n + i] : 0.0;
        const double w = n_inv ? n_inv[i] : 1.0;
#pragma unroll
        for (int b = 0; b < kProjChunk; ++b) {
            if (b < nbc) {
                double *tb = t_ + (int64_t)(b0 + b) * n;
                double u = tb[i];
                if (n_inv) { u *= w; tb[i] = u; }
#pragma unroll
                for (int k = 0; k < kFuseModesB; ++k)
                    if (k < nmodes) acc[b][k] = fma(p[k], u, acc[b][k]);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int b = 0; b < kProjChunk; ++b)
#pragma unroll
        for (int k = 0; k < kFuseModesB; ++k) {
            if (b < nbc && k < nmodes) {  // wave-uniform
                double v = acc[b][k];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) red[b][k][wave] = v;
            }
        }
    __syncthreads();
    if ((int)threadIdx.x < nmodes * nbc) {
        const int b = threadIdx.x / nmodes, k = threadIdx.x % nmodes;
        double v = 0.0;
        for (int w = 0; w < NT / 64; ++w) v += red[b][k][w];
        parts_[(int64_t)(b0 + b) * (kProjMaxModes * kProjParts) + k * kProjParts + part] = v;
    }
}

// launch shape of the coefficient pass over vectors of n doubles: threads per workgroup and workgroups (= partial sums) per mode
inline void tproj_coeffs_shape(int64_t n, int *nt, int *nparts)
{
    if (n >= (int64_t)kProjParts * 4096) { *nt = 1024; *nparts = kProjParts; return; }  // fine grids: 256 workgroups of 1024 threads
    int np = (int)((n + 511) / 512);  // coarse grids, 256 threads, two entries per thread: the pass is a latency chain of its loads
    if (np < 1) np = 1;
    if (np > kProjParts) np = kProjParts;
    *nt = 256; *nparts = np;
}

}  // namespace plshts
