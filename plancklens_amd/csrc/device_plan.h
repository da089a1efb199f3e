// Device-side view of a plan (plain pointers into hipMalloc'ed tables), passed by value to kernels.
#pragma once
#include <cstdint>

namespace plshts {

// Recursion state of every (m, ring pair) at the step where phase A (recursion only: no ring of the wave counts yet, legendre_math.h) of
// its wave ends, made once per plan by the same arithmetic (legendre.hip: k_seed_gen0 / k_seed_gens) for one kernel family: a grouping of
// `rg` ring pairs per wave, an activation threshold and a check interval.  A Legendre kernel whose grouping matches starts from the table
// (bit-identical state) instead of recursing from l = m: phase A is ~10 % of all recursion steps and runs at the latency of its dependent
// FMA chains.  rg = 0: no table.
struct DevSeedTab {
    const int *il;      // [mmax + 1][ngroups]: first recursion step of (m, ring group) that is not skipped (a multiple of the check interval)
    const double *st;   // [mmax + 1][npad][2] (spin 0: p0, p1) or [mmax + 1][npad][4] (spin s: n0, n1, p0, p1)
    const int *sc;      // [mmax + 1][npad] (spin 0) or [mmax + 1][npad][2] (spin s: scn, scp)
    int rg, npad;       // npad = ngroups * rg
};

struct DevSpinTab {
    const int64_t *off;      // [mmax + 2]
    const double *ab;        // 2 per entry
    const double *beta;      // 1 per entry
    const double *seedfac_n, *seedfac_p;  // [mmax + 1]
    const int *psin, *phalf, *usecos_n, *usecos_p;
    const int *mlim;         // [npairs]
    const int *gstart;       // [nmgroups] first ring group with any active ring, per group size (see api)
    DevSeedTab seed_syn, seed_ana;  // synthesis / analysis kernel family of this spin
};

// Scalar products formed by the post-processing kernel of an analysis (k_post0 / k_posts) on its way out: with q the alm it writes (nf = 1: spin 0;
// nf = 2: gradient and curl), s1[w] = the share of workgroup w of <d, q> and s2[w] that of <d, r>, summed over the fields, in the weights of the CG
// scalar product (1 on m = 0, 2 elsewhere, 0 below lmin: elementwise.hip alm_dot_weight).  One partial sum per workgroup, batch entry after batch
// entry (post_dots_count of them each); whoever consumes them adds them in index order (k_cg_axpy_pre).  conjugate-directions step of cd_solve.py:66-84
// without its own scalar-product launch.
struct PostDots {
    const double *d[2] = {nullptr, nullptr}, *r[2] = {nullptr, nullptr};
    double *s1 = nullptr, *s2 = nullptr;
    int lmin = 0;
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
    DevSeedTab seed_syn0, seed_ana0;  // spin-0 synthesis / analysis kernel family
};

}  // namespace plshts
