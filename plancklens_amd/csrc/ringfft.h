// FFT-stage tables (device pointers) and launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "device_plan.h"

namespace plshts {

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
    // workgroup-per-pair kernels with an LDS-resident transform (any ring): the pairs not covered by a register class
    const int *legacy_pairs; // ring pair indices, longest rings first
    int legacy_n;
    // register-resident kernels (ringfft.hip, second half): transform size classes N = 256 << c, c = 0 .. 4
    const int *cls_pairs[5]; // ring pair indices of each class
    int cls_n[5];
    const int *K2of;         // [nside + 1] band half-width of the sub-DFT inputs of ring length 4 q: |c| <= K2of[q]
    const int *M2of;         // [nside + 1] band-limited Bluestein size (0: q itself is the transform size, or legacy class)
    const int64_t *coff2;    // [nside + 1] offset of q's natural-order filter spectrum (M2of[q] entries)
    const double2 *filt2;
};

hipError_t launch_phase2map(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, const double *phase, double *map, hipStream_t st);
hipError_t launch_map2phase(const DevPlan &P, const DevFFT &F, const int *mlim, int ncomp, const double *map, double *phase, hipStream_t st);
hipError_t launch_twiddles(double *tw, int Mtw, hipStream_t st);
hipError_t launch_bluestein_setup(const DevFFT &F, const int *qlist_dev, int nq, double *chirp, double *filt, hipStream_t st);
hipError_t launch_bluestein_setup2(const DevFFT &F, const int *qlist_dev, int nq, int Mmax, double *filt2, hipStream_t st);

}  // namespace plshts
