// Device-side view of a plan (plain pointers into hipMalloc'ed tables), passed by value to kernels.
#pragma once
#include <cstdint>

namespace plshts {

struct DevSpinTab {
    const int64_t *off;      // [mmax + 2]
    const double *ab;        // 2 per entry
    const double *beta;      // 1 per entry
    const double *seedfac_n, *seedfac_p;  // [mmax + 1]
    const int *psin, *phalf, *usecos_n, *usecos_p;
    const int *mlim;         // [npairs]
    const int *gstart;       // [nmgroups] first ring group with any active ring, per group size (see api)
};

struct DevPlan {
    int nside, lmax, mmax, npairs, mstride;
    int64_t npix, nalm;
    // geometry (north member of each ring pair; the last pair is the equator, without partner)
    const double *cth, *sth, *chalf, *shalf, *phi0;
    const int *nphi;
    const int64_t *ofs_n, *ofs_s;
    // spin 0
    const int64_t *off0;     // [mmax + 2]
    const double *ab0;       // 2 per entry
    const double *alpha0;    // 1 per entry
    const double *eps0;      // 2 per entry
    const double *seed0;     // [mmax + 1]
    const int *mlim0;        // [npairs]
    int64_t nent0;           // total spin-0 entries
    // m-block shard (pl_plan_create_shard): the Legendre launches of this plan cover the m-groups mg0, mg0 + mgstride, ... only
    // (an m-group = 4 consecutive orders, the unit of a workgroup); 0 / 1 on an ordinary plan
    int mg0, mgstride;
};

}  // namespace plshts
