// FFT-stage tables (device pointers) and launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "device_plan.h"

namespace plshts {

// Work lists of the ring-FFT stage (both directions use the same band-limited classes): every ring pair is in exactly one list.
constexpr int kFftClasses = 5;  // register-resident transform sizes N = 256 << c, c = 0 .. 4 (256 ... 4096)

struct FftSide {
    const int *legacy_pairs; // generic LDS-resident kernel (short polar rings, aliased rings), longest rings first
    int legacy_n;
    int legacy_qmax;         // longest quarter-ring of the generic list: sizes its workgroups (threads x points per thread >= qmax)
    const int *cls_pairs[kFftClasses]; // register-resident kernels, transform size N = 256 << c: Bluestein rings (q != N)
    int cls_n[kFftClasses];
    const int *dir_pairs[kFftClasses]; // ... and the rings whose own sub-DFT length is N (q == N, a power of two)
    int dir_n[kFftClasses];
    // split Bluestein rings: the convolution of size 2 N runs as two halves of size N = 256 << c sharing one transform -- synthesis
    // 1 forward + 2 inverse (output halves j1 < N / 2 and j1 >= N / 2), analysis 2 forward + 1 inverse (input halves)
    const int *split_pairs[kFftClasses];
    int split_n[kFftClasses];
    const int *Mof;          // [nside + 1] Bluestein convolution size of q in its class (0: direct, or generic list)
    const int64_t *coff;     // [nside + 1] offset of q's natural-order filter spectrum (Mof[q] entries; split rings: two spectra)
    const int *split;        // [nside + 1] 1: q is a split ring
    const double2 *filt;
};

struct DevFFT {
    const double2 *tw;       // e^{-2 pi i t / Mtw}, t < Mtw / 2
    int Mtw;                 // largest power-of-two transform size used by the plan
    int Lmax;                // LDS workspace elements (double2) per workgroup
    int twl_cap;             // LDS twiddle-table elements after the workspace (0: read twiddles from global memory)
    const int *Mof;          // [nside + 1] Bluestein size for ring length 4 q, 0 when q is a power of two
    const int64_t *woff;     // [nside + 1] offset of q's chirp (q entries)
    const int64_t *coff;     // [nside + 1] offset of q's filter spectrum (M entries)
    const double2 *chirp;    // e^{i pi t^2 / q}
    const double2 *filt;     // bit-reversed FFT_M of the wrapped conj chirp, times 1 / M
    // register-resident kernels (ringfft.hip, second half): per-direction class lists and Bluestein tables
    int const *K2of;         // [nside + 1] band half-width of the sub-DFT bins of ring length 4 q: |c| <= K2of[q] (analysis)
    const double2 *ringc;    // [nside + 1][4] wave-uniform phase factors of ring length n = 4 q in its register class of size N (G = N / 8):
                             // e^{i pi / n}, e^{2 pi i G / n}, e^{4 pi i G / n}, e^{4 pi i N / n} (tables instead of sincos in every thread)
    FftSide A;               // class lists and band-limited Bluestein tables, shared by synthesis and analysis
    // analysis stage only, set per launch (launch_map2phase): component c of the input is read at map_ind[c] instead of map + c npix -- a table of
    // device pointers in device memory, dereferenced when the kernel runs: a captured launch can be replayed on other inputs (pl_map2alm_ind)
    const double *const *map_ind = nullptr;
};
__device__ __forceinline__ const double *fft_input_map(const DevFFT &F, const DevPlan &P, const double *map, int comp)
{
    return F.map_ind ? F.map_ind[comp] : map + (int64_t)comp * P.npix;
}

// Optional inverse-noise weighting with template marginalisation folded into the generic ring-FFT kernels (the CG operator of
// opfilt_tt.py:196-205 on the coarse multigrid grids, where every ring runs in the generic kernel):
//   synthesis side:  pixel u = n_inv t is stored instead of t, and the ring pair's share of c_k = sum_i pm[k][i] u_i goes to parts;
//   analysis side:   the pixels are read as u_i - sum_k rm[k][i] c_k, c_k = the sum of parts in ring-pair order (bit-reproducible).
constexpr int kFuseModes = 4;  // monopole + dipole; more template modes take the separate projection kernels
struct NinvProj {
    const double *n_inv = nullptr;  // [npix]; null: plain transform
    const double *pm = nullptr;     // [nmodes][npix]
    const double *rm = nullptr;     // [nmodes][npix]
    double *parts = nullptr;        // [ncomp][nmodes][nparts]: one set per component (= per entry of a batch of temperature maps)
    int nmodes = 0, nparts = 0;
};

// Side streams of a plan: the ring-length classes of one FFT stage are independent kernels of very different sizes
// (the short-ring ones are latency-bound), so they are forked from the caller's stream and joined back with events.
struct FftStreams {
    static constexpr int kN = 5;
    hipStream_t s[kN] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[kN] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ok = false;
};
hipError_t fft_streams_create(FftStreams &fs);
void fft_streams_destroy(FftStreams &fs);

hipError_t launch_phase2map(const DevPlan &P, const DevFFT &F, const FftStreams &fs, const int *mlim, int ncomp, const double *phase, double *map, hipStream_t st,
                            const NinvProj *W = nullptr);
hipError_t launch_map2phase(const DevPlan &P, const DevFFT &F, const FftStreams &fs, const int *mlim, int ncomp, const double *map, double *phase, hipStream_t st,
                            const NinvProj *W = nullptr, const double *const *map_ind = nullptr);
// phase -> pixels -> n_inv x pixels -> phase in one launch and in place (k_ring_roundtrip): plans with fft_all_generic only
hipError_t launch_ring_roundtrip(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, double *phase, const double *n_inv, hipStream_t st);
bool fft_all_generic(const DevPlan &P, const DevFFT &F);  // every ring pair runs in the generic kernel: NinvProj can be fused
hipError_t launch_twiddles(double *tw, int Mtw, hipStream_t st);
hipError_t launch_bluestein_setup(const DevFFT &F, const int *qlist_dev, int nq, double *chirp, double *filt, hipStream_t st);
hipError_t launch_bluestein_setup2(const DevFFT &F, const int *qlist_dev, int nq, int Mmax, double *filt2, hipStream_t st);

}  // namespace plshts
