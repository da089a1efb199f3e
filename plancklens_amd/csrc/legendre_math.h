// Per-lane arithmetic of the Legendre kernels: scaled seeds, recursion steps, alm <-> recursion-basis
// transforms.  Everything here is __host__ __device__ so that a host unit test can run the very same
// per-lane code on the CPU.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define PL_HD __host__ __device__ __forceinline__
#else
#define PL_HD inline
#endif

namespace plshts {

// A value is represented as v * 2^(512 s), s <= 0, with |v| kept inside [2^-256, 2^256] while s < 0.
// A lane is "active" (s == 0) once the true value has reached 2^-256; below that its contribution to
// any sum is < 1e-77 of an O(1) term and is dropped, exactly as libsharp does below its own threshold.
constexpr double kFBig = 0x1p+512, kFSmall = 0x1p-512, kTBig = 0x1p+256, kTSmall = 0x1p-256;
constexpr int kNeverActive = -(1 << 28);
// When does a wave start to accumulate?  The kernels run three phases per wave: (A) recursion only while no ring of the wave counts,
// (B) masked accumulation while some rings are still scaled, (C) the pure FMA stream once every ring is in the double range.  A ring
// *counts* once it is active AND its value has reached an activation threshold 2^-k: the rings of a wave become O(1) at different l, and
// until round 5 a wave left phase A as soon as its first ring entered the double range (2^-256) -- 2-3 % of all recursion steps ran their
// accumulation FMAs on terms below 1e-30 of an O(1) term.  (libsharp drops what lies below 2^-60.)  The check is made once per block of
// recursion steps, so a ring that crosses the threshold inside a block is noticed at the block's end; the thresholds are set per kernel
// family from the largest value that is left out of a sum that way, measured over all (m, ring) of an nside = lmax = 2048 grid:
//   synthesis spin s (blocks of  8 l): 2^-112 -> at most 2^-89      synthesis spin 0 (8 two-l steps = 16 l): 2^-128 -> 2^-85
//   analysis  spin s (tiles of  16 l): 2^-128 -> at most 2^-85      analysis  spin 0 (16 two-l steps = 32 l): 2^-160 -> 2^-82
// i.e. below 1e-24 of an O(1) term.  Phases B and C are unchanged (a ring in the double range is summed from there on, however small).
constexpr double kActSynthS = 0x1p-112, kActSynth0 = 0x1p-128, kActAnalS = 0x1p-128, kActAnal0 = 0x1p-160;

PL_HD void renorm(double &v, int &s)
{
    double a = fabs(v);
    if (a > kTBig) { v *= kFSmall; s += 1; }
    else if (a < kTSmall && a != 0.0) { v *= kFBig; s -= 1; }
}

// b^e for 0 <= b <= 1, e >= 0 by binary powering with renormalisation
PL_HD void scaled_pow(double b, int e, double &v, int &s)
{
    double r = 1.0, bb = b;
    int rs = 0, bs = 0;
    while (e) {
        if (e & 1) { r *= bb; rs += bs; renorm(r, rs); }
        bb *= bb; bs *= 2; renorm(bb, bs);
        e >>= 1;
    }
    v = r; s = rs;
}

PL_HD double small_pow(double b, int e)
{
    double r = 1.0;
    for (int i = 0; i < e; ++i) r *= b;
    return r;
}

// ---- spin 0, two-step recursion ---------------------------------------------------------------------
// state: p0 = P_{il-1}, p1 = P_il (scaled by 2^(512 sc)), x2 = cos^2(theta)
struct Rec0 {
    double p0, p1, x2;
    int sc;
};

PL_HD void rec0_init(Rec0 &r, double seed_m, int m, double cth, double sth, bool ring_active)
{
    r.x2 = cth * cth;
    r.p0 = 0.0;
    double v; int s;
    scaled_pow(sth, m, v, s);
    v *= seed_m;
    renorm(v, s);
    if (!ring_active || v == 0.0) { v = 0.0; s = kNeverActive; }
    if (s > 0) { v *= kFBig; s -= 1; }  // cannot happen for |seed| sin^m < 2^256, kept for safety
    r.p1 = v; r.sc = s;
}

// advance il -> il + 1 (A, B of the current il)
PL_HD void rec0_step_fast(Rec0 &r, double A, double B)
{
    double t = fma(A, r.x2, B);
    double pn = fma(t, r.p1, -r.p0);
    r.p0 = r.p1; r.p1 = pn;
}

PL_HD void rec0_step_careful(Rec0 &r, double A, double B)
{
    rec0_step_fast(r, A, B);
    if (r.sc < 0 && fabs(r.p1) > kTBig) { r.p0 *= kFSmall; r.p1 *= kFSmall; r.sc += 1; }
}

// Deferred form of the careful step: a still-scaled lane may run up to 8 fast steps before this check.  One step
// multiplies the value by at most |A| + |B| + 1 (resp. |a| + |b| + 1), which stays below 2^16 for every table entry at
// lmax < 2^15, so at most 2^128 is gained between checks: far from overflow (2^256 * 2^128 << 2^1024) and a single
// rescale is always enough.  A lane that crosses 2^-256 between checks is below 2^-128 when it is noticed; until
// then it is left out of the sums, which is exact at double precision for any sum with an O(1) term.
PL_HD void rec0_renorm_up(Rec0 &r)
{
    if (r.sc < 0 && fabs(r.p1) > kTBig) { r.p0 *= kFSmall; r.p1 *= kFSmall; r.sc += 1; }
}

// does this ring end phase A of its wave?
PL_HD bool rec0_counts(const Rec0 &r, double thr) { return r.sc == 0 && fabs(r.p1) >= thr; }

// value usable in sums for the current il (0 while still scaled)
PL_HD double rec0_value(const Rec0 &r) { return r.sc == 0 ? r.p1 : 0.0; }

// ---- spin s >= 1, one-step recursion of the (n = -s, n = +s) pair --------------------------------------
struct RecS {
    double n0, n1, p0, p1, x;  // S^-_{l-1}, S^-_l, S^+_{l-1}, S^+_l, cos(theta)
    int scn, scp;
};

PL_HD void recs_seed_one(double fac, int psin, int phalf, double sth, double half, bool ring_active, double &v, int &s)
{
    scaled_pow(sth, psin, v, s);
    v *= small_pow(half, phalf);
    renorm(v, s);
    v *= fac;
    renorm(v, s);
    if (!ring_active || v == 0.0) { v = 0.0; s = kNeverActive; }
    if (s > 0) { v *= kFBig; s -= 1; }
}

PL_HD void recs_init(RecS &r, double fac_n, double fac_p, int psin, int phalf, int usecos_n, int usecos_p,
                     double cth, double sth, double chalf, double shalf, bool ring_active)
{
    r.x = cth; r.n0 = 0.0; r.p0 = 0.0;
    recs_seed_one(fac_n, psin, phalf, sth, usecos_n ? chalf : shalf, ring_active, r.n1, r.scn);
    recs_seed_one(fac_p, psin, phalf, sth, usecos_p ? chalf : shalf, ring_active, r.p1, r.scp);
}

PL_HD void recs_step_fast(RecS &r, double a, double b)
{
    double tn = fma(r.x, a, b);   // n = -s:  x a + b
    double tp = fma(r.x, a, -b);  // n = +s:  x a - b
    double nn = fma(tn, r.n1, -r.n0);
    double pn = fma(tp, r.p1, -r.p0);
    r.n0 = r.n1; r.n1 = nn; r.p0 = r.p1; r.p1 = pn;
}

PL_HD void recs_step_careful(RecS &r, double a, double b)
{
    recs_step_fast(r, a, b);
    if (r.scn < 0 && fabs(r.n1) > kTBig) { r.n0 *= kFSmall; r.n1 *= kFSmall; r.scn += 1; }
    if (r.scp < 0 && fabs(r.p1) > kTBig) { r.p0 *= kFSmall; r.p1 *= kFSmall; r.scp += 1; }
}

PL_HD void recs_renorm_up(RecS &r)  // see rec0_renorm_up
{
    if (r.scn < 0 && fabs(r.n1) > kTBig) { r.n0 *= kFSmall; r.n1 *= kFSmall; r.scn += 1; }
    if (r.scp < 0 && fabs(r.p1) > kTBig) { r.p0 *= kFSmall; r.p1 *= kFSmall; r.scp += 1; }
}

PL_HD bool recs_counts(const RecS &r, double thr) { return (r.scn == 0 && fabs(r.n1) >= thr) || (r.scp == 0 && fabs(r.p1) >= thr); }

PL_HD double recs_value_n(const RecS &r) { return r.scn == 0 ? r.n1 : 0.0; }
PL_HD double recs_value_p(const RecS &r) { return r.scp == 0 ? r.p1 : 0.0; }

// ---- alm <-> recursion basis ---------------------------------------------------------------------------
// spin 0 synthesis: sum_l a_l lambda_l = sum_il (c_il + x d_il) P_il with
//   c_il = alpha_l (eps_{l+1} a_l + eps_{l+2} a_{l+2}),  d_il = alpha_l a_{l+1},  l = m + 2 il  (a_{>lmax} = 0)
// spin 0 analysis (adjoint): a_l = eps_{l+1} alpha_l C_il + eps_l alpha_{l-2} C_{il-1} (l - m even),
//                            a_{l+1} = alpha_l D_il
// spin s synthesis: An_l = -1/2 sg beta_l (G + iC)_l, Ap_l = -1/2 beta_l (G - iC)_l, sg = (-1)^s
//   X_N = sum Sn An, Y_N = sum Sp Ap, X_S = sum sigma_l Sp An, Y_S = sum sigma_l Sn Ap, sigma_l = (-1)^{l+m}
//   Q = X + Y, U = i (Y - X)
// spin s analysis (adjoint), Wp = Q + iU, Wm = Q - iU, mirror ring: Sn <-> sigma_l Sp:
//   G'_l = sum_pairs Sn_l (sg Wp_N + sigma_l Wm_S) + Sp_l (Wm_N + sg sigma_l Wp_S)
//   C'_l = sum_pairs Sn_l (sg Wp_N - sigma_l Wm_S) - Sp_l (Wm_N - sg sigma_l Wp_S)
//   G_l = -1/2 beta_l G'_l,  C_l = i/2 beta_l C'_l

}  // namespace plshts
