/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the spin-weighted Legendre stage of the HEALPix SHTs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
 * product path (plancklens_amd/csrc, HIP) never links or calls it.
 *
 * What is restated.  The reference (plancklens/shts.py:12-35) delegates alm2map / map2alm /
 * alm2map_spin / map2alm_spin to third-party healpy (C++ libsharp; un-vendored, un-pinned:
 * pyproject.toml:11-16).  Its published algorithm (Reinecke & Seljebotn 2013, "Libsharp - spherical
 * harmonic transforms revisited") is: per azimuthal order m, a three-term recursion in l of the
 * normalised spin-weighted Legendre functions on every iso-latitude ring, accumulated into per-ring
 * Fourier coefficients F_m(ring), followed by one FFT per ring.  This file is the Legendre stage; the
 * ring FFTs and the HEALPix geometry are in oracle/sht_oracle.py (numpy pocketfft).
 *
 * Conventions (plancklens/utils_spin.py:1-16):
 *   spin 0 :  F_m(theta)         = sum_l a_lm lambda_lm(theta),  Y_lm = lambda_lm e^{i m phi}
 *   spin s>0: _{+-s}a_lm = -(+-1)^s (G_lm +- i C_lm),  _sf = Q + iU = sum _sa_lm _sY_lm
 *             _sY_lm(theta,0) = (-1)^s sqrt((2l+1)/4pi) d^l_{m,-s}(theta)
 *             Fp = -( _slam + (-1)^s _{-s}lam )/2 ,  Fm = -( _slam - (-1)^s _{-s}lam )/2
 *             Q_m = sum_l (G Fp + i C Fm) ,  U_m = sum_l (C Fp - i G Fm)
 *   analysis is the exact adjoint: G_lm = sum_rings (Q_m Fp + i U_m Fm), C_lm = sum_rings (U_m Fp - i Q_m Fm)
 *   (ring quadrature weights and e^{-i m phi0} are applied by the FFT stage).
 *
 * Two arithmetic modes:
 *   mode 0: x87 long double, unscaled textbook Wigner-d recursion seeded from lgamma -- slow,
 *           obviously correct, used to pin mode 1 and the HIP kernels at small sizes.
 *   mode 1: double with libsharp-style 2^(+-512) block scaling, ring-blocked and OpenMP-threaded over m;
 *           this is the "port" CPU baseline timed by bench.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846264338327950288L

static inline int64_t alm_index(int lmax, int l, int m) { return (int64_t)m * (2 * lmax + 1 - m) / 2 + l; }

/* ------------------------------------------------------------------------------------------- */
/* mode 0: long double                                                                         */
/* ------------------------------------------------------------------------------------------- */

/* Normalised Wigner small-d: out[l] = sqrt((2l+1)/4pi) d^l_{m,n}(theta), l = 0..lmax (0 below max(|m|,|n|)). */
static void ld_wigd(int m, int n, int lmax, long double theta, long double *out)
{
    int am = m < 0 ? -m : m, an = n < 0 ? -n : n;
    int j = am > an ? am : an;
    for (int l = 0; l <= lmax; ++l) out[l] = 0.0L;
    if (j > lmax) return;
    long double x = cosl(theta), c = cosl(0.5L * theta), s = sinl(0.5L * theta);
    /* map d^j_{m,n} onto sign * d^j_{j,k} with the symmetries d_{m,n} = (-1)^{m-n} d_{n,m} = d_{-n,-m} */
    int k; long double sign = 1.0L;
    if (j == m)       { k = n; }
    else if (j == n)  { k = m;  if ((j - m) & 1) sign = -1.0L; }        /* d_{m,j} = (-1)^{m-j} d_{j,m} */
    else if (j == -n) { k = -m; }                                       /* d_{m,-j} = d_{j,-m}          */
    else              { k = -n; if ((j + n) & 1) sign = -1.0L; }        /* m = -j: d_{-j,n} = (-1)^{n+j} d_{j,-n} */
    /* d^j_{j,k} = sqrt((2j)!/((j+k)!(j-k)!)) cos^{j+k}(theta/2) (-sin(theta/2))^{j-k} */
    long double lg = 0.5L * (lgammal(2.0L * j + 1) - lgammal((long double)(j + k) + 1) - lgammal((long double)(j - k) + 1));
    long double seed;
    if ((j + k > 0 && c <= 0.0L) || (j - k > 0 && s <= 0.0L)) seed = 0.0L;
    else {
        if (j + k > 0) lg += (j + k) * logl(c);
        if (j - k > 0) lg += (j - k) * logl(s);
        seed = (lg < -11300.0L) ? 0.0L : expl(lg);
    }
    if ((j - k) & 1) sign = -sign;
    long double dm1 = 0.0L, d0 = sign * seed;
    out[j] = sqrtl((2.0L * j + 1) / (4.0L * ORC_PI)) * d0;
    for (int l = j; l < lmax; ++l) {
        long double d1;
        if (l == 0) d1 = x * d0; /* m = n = 0 */
        else {
            long double L = l, L1 = l + 1;
            long double num1 = (2 * L + 1) * (L * L1 * x - (long double)m * n);
            long double num2 = L1 * sqrtl((L * L - (long double)m * m) * (L * L - (long double)n * n));
            long double den = L * sqrtl((L1 * L1 - (long double)m * m) * (L1 * L1 - (long double)n * n));
            d1 = (num1 * d0 - num2 * dm1) / den;
        }
        dm1 = d0; d0 = d1;
        out[l + 1] = sqrtl((2.0L * (l + 1) + 1) / (4.0L * ORC_PI)) * d0;
    }
}

/* fp[l], fm[l], l = 0..lmax for spin s >= 0 and order m >= 0 at colatitude theta. */
static void ld_lam(int s, int m, int lmax, long double theta, long double *fp, long double *fm, long double *tmp)
{
    if (s == 0) {
        ld_wigd(m, 0, lmax, theta, fp);
        for (int l = 0; l <= lmax; ++l) fm[l] = 0.0L;
        return;
    }
    long double sg = (s & 1) ? -1.0L : 1.0L;
    ld_wigd(m, -s, lmax, theta, fp);   /* -> _slam  = sg * this  */
    ld_wigd(m, +s, lmax, theta, tmp);  /* -> _-slam = sg * this  */
    for (int l = 0; l <= lmax; ++l) {
        long double lp = sg * fp[l], lm = sg * tmp[l];
        fp[l] = -0.5L * (lp + sg * lm);
        fm[l] = -0.5L * (lp - sg * lm);
    }
}

/* Exported for tests: double copies of Fp, Fm (or lambda_lm for s = 0). */
void orc_lambda(int s, int m, int lmax, double cth, double sth, double *fp, double *fm)
{
    long double *a = malloc(3 * (size_t)(lmax + 1) * sizeof(long double));
    long double theta = atan2l((long double)sth, (long double)cth);
    ld_lam(s, m, lmax, theta, a, a + (lmax + 1), a + 2 * (lmax + 1));
    for (int l = 0; l <= lmax; ++l) { fp[l] = (double)a[l]; fm[l] = (double)a[lmax + 1 + l]; }
    free(a);
}

/* ------------------------------------------------------------------------------------------- */
/* mode 1: double, scaled                                                                      */
/* ------------------------------------------------------------------------------------------- */
#define NV 8                    /* rings per block (vector direction)        */
#define SCALE_BITS 512
static const double F_BIG = 0x1p+512, F_SMALL = 0x1p-512, T_BIG = 0x1p+256, T_SMALL = 0x1p-256;

/* value = v * 2^(512 sc); keep |v| in [2^-256, 2^256] */
static inline void renorm(double *v, int *sc)
{
    double a = fabs(*v);
    if (a > T_BIG) { *v *= F_SMALL; *sc += 1; }
    else if (a < T_SMALL && a != 0.0) { *v *= F_BIG; *sc -= 1; }
}

/* b^e (e >= 0, 0 < b <= 1 typically) as scaled number */
static void scaled_pow(double b, int e, double *v, int *sc)
{
    double r = 1.0, bb = b; int rs = 0, bs = 0;
    while (e) {
        if (e & 1) { r *= bb; rs += bs; renorm(&r, &rs); }
        bb *= bb; bs *= 2; renorm(&bb, &bs);
        e >>= 1;
    }
    *v = r; *sc = rs;
}

/* Per-m tables of the one-step recursion  R_{l+1} = (x a_l - b_l) R_l - c_l R_{l-1}  for the
 * normalised function  sqrt((2l+1)/4pi) d^l_{m,n}.  (b = 0 for n = 0.)  Computed in long double. */
typedef struct { double *a, *b, *c; double seedfac; int l0; int psin, phalf, usecos; double seedsign; } rectab;

static void rectab_build(rectab *t, int m, int n, int lmax)
{
    int an = n < 0 ? -n : n;
    int j = m > an ? m : an;
    t->l0 = j;
    if (j > lmax) return;
    for (int l = j; l < lmax; ++l) {
        long double L = l, L1 = l + 1, M = m, N = n;
        if (l == 0) { t->a[l] = (double)sqrtl(3.0L); t->b[l] = 0; t->c[l] = 0; continue; }
        long double den = L * sqrtl((L1 * L1 - M * M) * (L1 * L1 - N * N));
        long double nrm1 = sqrtl((2 * L1 + 1) / (2 * L + 1));
        long double nrm2 = sqrtl((2 * L1 + 1) / (2 * L - 1));
        t->a[l] = (double)(nrm1 * (2 * L + 1) * L * L1 / den);
        t->b[l] = (double)(nrm1 * (2 * L + 1) * M * N / den);
        t->c[l] = (l == j) ? 0.0 : (double)(nrm2 * L1 * sqrtl((L * L - M * M) * (L * L - N * N)) / den);
    }
    /* seed: sqrt((2j+1)/4pi) * sign * sqrt((2j)!/((j+k)!(j-k)!)) * cos^{j+k}(th/2) * sin^{j-k}(th/2)
     *     = sign * [sqrt((2j+1)/4pi) sqrt(C(2j,j+k)) 2^-(j-|k|)] * sin^{j-|k|}(th) * (cos or sin)^{2|k|}(th/2) */
    int k; long double sign = 1.0L;
    if (j == m)       { k = n; }
    else if (j == n)  { k = m;  if ((j - m) & 1) sign = -1.0L; }
    else              { k = -m; }
    if ((j - k) & 1) sign = -sign;
    int ak = k < 0 ? -k : k;
    long double lg = 0.5L * (lgammal(2.0L * j + 1) - lgammal((long double)(j + k) + 1) - lgammal((long double)(j - k) + 1))
                     - (long double)(j - ak) * logl(2.0L);
    t->seedfac = (double)(sqrtl((2.0L * j + 1) / (4.0L * ORC_PI)) * expl(lg));
    t->seedsign = (double)sign;
    t->psin = j - ak; t->phalf = 2 * ak; t->usecos = k > 0;
}

/* Fill lam[l*NV + v] (l = l0..lmax) with the IEEE-range values of the function (0 while still below 2^-256),
 * for NV rings.  Returns the first l with a non-zero entry (lmax + 1 if none).
 * Two parts: rec_scaled seeds the recursion and runs it while at least one ring is still below the IEEE window (per-lane scale
 * bookkeeping); rec_ieee continues with the plain three-term step.  rec_fill2 runs the two functions of a spin-weighted transform
 * (n = -s and n = +s) through the IEEE part together: one recursion is a chain of dependent FMAs, two interleaved ones fill the pipe. */
typedef struct { double v1[NV], v2[NV]; int l, first; } recstate;

static void rec_scaled(const rectab *t, int lmax, const double *x, const double *st, const double *ch, const double *sh, int nv, double *lam,
                       recstate *S)
{
    double *v1 = S->v1, *v2 = S->v2; int sc[NV];
    int l0 = t->l0;
    for (int v = 0; v < NV; ++v) { v1[v] = 0; v2[v] = 0; sc[v] = 0; }
    for (int v = 0; v < nv; ++v) {
        double pv, qv; int ps_, qs_;
        scaled_pow(st[v], t->psin, &pv, &ps_);
        scaled_pow(t->usecos ? ch[v] : sh[v], t->phalf, &qv, &qs_);
        double r = pv * qv; int rs = ps_ + qs_;
        renorm(&r, &rs);
        r *= t->seedfac * t->seedsign; renorm(&r, &rs);
        if (r == 0.0) rs = -1000000;
        v2[v] = r; sc[v] = rs;
    }
    for (int v = nv; v < NV; ++v) sc[v] = 0;
    int l = l0, first = lmax + 1;
    /* scaled phase: at least one ring still below the IEEE window */
    for (; l <= lmax; ++l) {
        int allin = 1;
        for (int v = 0; v < NV; ++v) allin &= (sc[v] == 0);
        if (allin) break;
        int any = 0;
        for (int v = 0; v < NV; ++v) { double o = (sc[v] == 0) ? v2[v] : 0.0; lam[(size_t)l * NV + v] = o; any |= (sc[v] == 0); }
        if (any && first > lmax) first = l;
        if (l == lmax) { ++l; break; }
        double a = t->a[l], b = t->b[l], c = t->c[l];
        for (int v = 0; v < NV; ++v) {
            double nw = (x[v] * a - b) * v2[v] - c * v1[v];
            v1[v] = v2[v]; v2[v] = nw;
            if (sc[v] < 0 && fabs(nw) > T_BIG) { v1[v] *= F_SMALL; v2[v] *= F_SMALL; sc[v] += 1; }
        }
    }
    if (l <= lmax && first > lmax) first = l;
    S->l = l; S->first = first;
}

/* IEEE phase from S->l up to (and including) lend <= lmax: stores lam[l] and steps; on return S->l = lend + 1 (state = value at lend + 1
 * unless lend = lmax) */
static void rec_ieee(const rectab *t, int lmax, const double *x, double *lam, recstate *S, int lend)
{
    double *v1 = S->v1, *v2 = S->v2;
    int l = S->l;
    for (; l <= lend; ++l) {
        for (int v = 0; v < NV; ++v) lam[(size_t)l * NV + v] = v2[v];
        if (l == lmax) { ++l; break; }
        double a = t->a[l], b = t->b[l], c = t->c[l];
#pragma omp simd
        for (int v = 0; v < NV; ++v) {
            double nw = (x[v] * a - b) * v2[v] - c * v1[v];
            v1[v] = v2[v]; v2[v] = nw;
        }
    }
    S->l = l;
}

static int rec_fill(const rectab *t, int lmax, const double *x, const double *st, const double *ch, const double *sh, int nv, double *lam)
{
    recstate S;
    rec_scaled(t, lmax, x, st, ch, sh, nv, lam, &S);
    rec_ieee(t, lmax, x, lam, &S, lmax);
    return S.first;
}

typedef struct { double x[NV], st[NV], ch[NV], sh[NV]; int nv; } blkgeom;

/* two recursions side by side: functions ta, tb on ring blocks ga, gb (the same block for the two functions of a spin-weighted
 * transform, two different blocks of one function for spin 0) */
static void rec_fill2(const rectab *ta, const rectab *tb, int lmax, const blkgeom *ga, const blkgeom *gb,
                      double *lama, double *lamb, int *firsta, int *firstb)
{
    recstate A, B;
    const double *x = ga->x, *xb = gb->x;
    rec_scaled(ta, lmax, ga->x, ga->st, ga->ch, ga->sh, ga->nv, lama, &A);
    rec_scaled(tb, lmax, gb->x, gb->st, gb->ch, gb->sh, gb->nv, lamb, &B);
    *firsta = A.first; *firstb = B.first;
    /* the one that reached the IEEE window first goes on alone until the other has */
    if (A.l < B.l) rec_ieee(ta, lmax, x, lama, &A, B.l - 1 < lmax ? B.l - 1 : lmax);
    else if (B.l < A.l) rec_ieee(tb, lmax, xb, lamb, &B, A.l - 1 < lmax ? A.l - 1 : lmax);
    /* the joint loop on local copies (no aliasing with the state structs: the loop vectorises with both chains in registers) */
    double xa_[NV], xb_[NV], p1[NV], p2[NV], q1[NV], q2[NV];
    for (int v = 0; v < NV; ++v) { xa_[v] = x[v]; xb_[v] = xb[v]; p1[v] = A.v1[v]; p2[v] = A.v2[v]; q1[v] = B.v1[v]; q2[v] = B.v2[v]; }
    for (int l = A.l; l <= lmax; ++l) {
        double *restrict oa = lama + (size_t)l * NV, *restrict ob = lamb + (size_t)l * NV;
        for (int v = 0; v < NV; ++v) { oa[v] = p2[v]; ob[v] = q2[v]; }
        if (l == lmax) break;
        const double a1 = ta->a[l], b1 = ta->b[l], c1 = ta->c[l], a2 = tb->a[l], b2 = tb->b[l], c2 = tb->c[l];
#pragma omp simd
        for (int v = 0; v < NV; ++v) {
            const double n1 = (xa_[v] * a1 - b1) * p2[v] - c1 * p1[v], n2 = (xb_[v] * a2 - b2) * q2[v] - c2 * q1[v];
            p1[v] = p2[v]; p2[v] = n1; q1[v] = q2[v]; q2[v] = n2;
        }
    }
}

/* libsharp's polar-optimisation bound: orders m above this contribute < ~1e-30 on the ring */
static int mlim_ring(int lmax, int spin, double sth, double cth)
{
    double ofs = lmax * 0.01;
    if (ofs < 100.) ofs = 100.;
    double b = -2 * spin * fabs(cth);
    double t1 = lmax * sth + ofs;
    double c = (double)spin * spin - t1 * t1;
    double discr = b * b - 4 * c;
    if (discr <= 0) return lmax;
    double res = (-b + sqrt(discr)) / 2.;
    if (res > lmax) res = lmax;
    return (int)(res + 0.5);
}

/* ------------------------------------------------------------------------------------------- */
/* drivers                                                                                     */
/* ------------------------------------------------------------------------------------------- */
/* Ring list: nring rings with cos/sin(theta).  If pair[r] != 0 the ring has a mirror partner at
 * pi - theta whose phases are stored right after the ring's own (slot index 2*r and 2*r + 1 in the
 * phase array); otherwise slot 2*r + 1 is unused.
 *
 * phase layout: [comp][slot = 2*nring][m = 0..mmax] complex (interleaved re, im)
 * alm layout:   [comp][healpy index] complex interleaved; ncomp = 1 (spin 0) or 2 (G, C).
 */

static void get_lam_ld(int spin, int m, int lmax, double cth, double sth, double *fp, double *fm, long double *scr)
{
    long double theta = atan2l((long double)sth, (long double)cth);
    ld_lam(spin, m, lmax, theta, scr, scr + (lmax + 1), scr + 2 * (lmax + 1));
    for (int l = 0; l <= lmax; ++l) { fp[l] = (double)scr[l]; fm[l] = (double)scr[lmax + 1 + l]; }
}

/* generic kernel given tables of Fp/Fm for one (m, ring): synthesis of both hemispheres */
static inline void synth_one(int spin, int lmax, int m, int l0, const double *fp, const double *fm, int stride,
                             const double *almG, const double *almC, /* pointers to (l = 0) of this m, interleaved */
                             double *qn, double *un, double *qs, double *us)
{
    /* accumulate even / odd (l + m + s) parities separately: mirror ring flips the odd Fp and even Fm parts */
    double qe[2] = {0, 0}, qo[2] = {0, 0}, ue[2] = {0, 0}, uo[2] = {0, 0};
    for (int l = l0; l <= lmax; ++l) {
        double p = fp[(size_t)l * stride], q = fm ? fm[(size_t)l * stride] : 0.0;
        double gr = almG[2 * l], gi = almG[2 * l + 1];
        double cr = almC ? almC[2 * l] : 0.0, ci = almC ? almC[2 * l + 1] : 0.0;
        /* Q += G p + i C q ; U += C p - i G q */
        double tq_p[2] = {gr * p, gi * p}, tq_m[2] = {-ci * q, cr * q};
        double tu_p[2] = {cr * p, ci * p}, tu_m[2] = {gi * q, -gr * q};
        if (((l + m + spin) & 1) == 0) { /* Fp even under mirror, Fm odd */
            qe[0] += tq_p[0]; qe[1] += tq_p[1]; qo[0] += tq_m[0]; qo[1] += tq_m[1];
            ue[0] += tu_p[0]; ue[1] += tu_p[1]; uo[0] += tu_m[0]; uo[1] += tu_m[1];
        } else {
            qo[0] += tq_p[0]; qo[1] += tq_p[1]; qe[0] += tq_m[0]; qe[1] += tq_m[1];
            uo[0] += tu_p[0]; uo[1] += tu_p[1]; ue[0] += tu_m[0]; ue[1] += tu_m[1];
        }
    }
    qn[0] = qe[0] + qo[0]; qn[1] = qe[1] + qo[1];
    un[0] = ue[0] + uo[0]; un[1] = ue[1] + uo[1];
    if (qs) { qs[0] = qe[0] - qo[0]; qs[1] = qe[1] - qo[1]; us[0] = ue[0] - uo[0]; us[1] = ue[1] - uo[1]; }
}

static inline void anal_one(int spin, int lmax, int m, int l0, const double *fp, const double *fm, int stride,
                            double *almG, double *almC,
                            const double *qn, const double *un, const double *qs, const double *us)
{
    double qe[2], qo[2], ue[2] = {0, 0}, uo[2] = {0, 0};
    double z[2] = {0, 0};
    if (!qs) { qs = z; us = z; }
    if (!un) { un = z; us = z; }
    qe[0] = qn[0] + qs[0]; qe[1] = qn[1] + qs[1]; qo[0] = qn[0] - qs[0]; qo[1] = qn[1] - qs[1];
    ue[0] = un[0] + us[0]; ue[1] = un[1] + us[1]; uo[0] = un[0] - us[0]; uo[1] = un[1] - us[1];
    for (int l = l0; l <= lmax; ++l) {
        double p = fp[(size_t)l * stride], q = fm ? fm[(size_t)l * stride] : 0.0;
        const double *qp, *qm, *up, *um;
        if (((l + m + spin) & 1) == 0) { qp = qe; qm = qo; up = ue; um = uo; }
        else { qp = qo; qm = qe; up = uo; um = ue; }
        /* G += Q p + i U q ; C += U p - i Q q */
        almG[2 * l] += qp[0] * p - um[1] * q;
        almG[2 * l + 1] += qp[1] * p + um[0] * q;
        if (almC) {
            almC[2 * l] += up[0] * p + qm[1] * q;
            almC[2 * l + 1] += up[1] * p - qm[0] * q;
        }
    }
}


/* ---- mode 1 accumulation, NV rings at a time (the vector direction of rec_fill's tables) ----------------------------------
 * Same sums as synth_one / anal_one, formed for the NV rings of a block together: the Legendre values of one l are one vector
 * (fp[l * NV + v]), the alm coefficients are broadcast scalars (synthesis) or the targets of horizontal sums (analysis).  The l loop
 * runs in steps of two so that the parity roles (which of the even / odd accumulators a term goes to) are fixed inside the body.
 * This is what makes the "port" baseline a vectorised code (AVX2 / AVX-512 through -march=native and omp simd); the scalar
 * per-ring routines above remain the long-double route and the reference the block routines are tested against. */
typedef struct { double qe[2][NV], qo[2][NV], ue[2][NV], uo[2][NV]; } blkacc;

static inline void synth_blk_step(int spin, const double *p, const double *q, const double *g, const double *c,
                                  double (*ep)[NV], double (*om)[NV], double (*uep)[NV], double (*uom)[NV])
{
    /* a term with Legendre value p goes to (ep, uep), one with q to (om, uom): Q += G p + i C q, U += C p - i G q */
    const double gr = g[0], gi = g[1];
    if (spin == 0) {
#pragma omp simd
        for (int v = 0; v < NV; ++v) { ep[0][v] += gr * p[v]; ep[1][v] += gi * p[v]; }
        return;
    }
    const double cr = c[0], ci = c[1];
#pragma omp simd
    for (int v = 0; v < NV; ++v) {
        const double pv = p[v], qv = q[v];
        ep[0][v] += gr * pv; ep[1][v] += gi * pv;
        om[0][v] -= ci * qv; om[1][v] += cr * qv;
        uep[0][v] += cr * pv; uep[1][v] += ci * pv;
        uom[0][v] += gi * qv; uom[1][v] -= gr * qv;
    }
}

static void synth_blk(int spin, int lmax, int m, int first, const double *fp, const double *fm, const double *almG, const double *almC, blkacc *A)
{
    memset(A, 0, sizeof(*A));
    int l = first;
    if (l <= lmax && ((l + m + spin) & 1)) {  /* odd parity: Fp terms to the odd sums, Fm terms to the even ones */
        synth_blk_step(spin, fp + (size_t)l * NV, fm + (size_t)l * NV, almG + 2 * l, almC ? almC + 2 * l : NULL, A->qo, A->qe, A->uo, A->ue);
        ++l;
    }
    for (; l + 1 <= lmax; l += 2) {
        synth_blk_step(spin, fp + (size_t)l * NV, fm + (size_t)l * NV, almG + 2 * l, almC ? almC + 2 * l : NULL, A->qe, A->qo, A->ue, A->uo);
        synth_blk_step(spin, fp + (size_t)(l + 1) * NV, fm + (size_t)(l + 1) * NV, almG + 2 * (l + 1), almC ? almC + 2 * (l + 1) : NULL,
                       A->qo, A->qe, A->uo, A->ue);
    }
    if (l <= lmax)
        synth_blk_step(spin, fp + (size_t)l * NV, fm + (size_t)l * NV, almG + 2 * l, almC ? almC + 2 * l : NULL, A->qe, A->qo, A->ue, A->uo);
}

static inline void anal_blk_step(int spin, const double *p, const double *q, double *g, double *c,
                                 double (*qp)[NV], double (*qm)[NV], double (*up)[NV], double (*um)[NV])
{
    /* G += Q p + i U q ; C += U p - i Q q, with (qp, up) the parity class that meets Fp at this l and (qm, um) the one that meets Fm */
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (spin == 0) {
#pragma omp simd reduction(+ : s0, s1)
        for (int v = 0; v < NV; ++v) { s0 += qp[0][v] * p[v]; s1 += qp[1][v] * p[v]; }
        g[0] += s0; g[1] += s1;
        return;
    }
#pragma omp simd reduction(+ : s0, s1, s2, s3)
    for (int v = 0; v < NV; ++v) {
        const double pv = p[v], qv = q[v];
        s0 += qp[0][v] * pv - um[1][v] * qv;
        s1 += qp[1][v] * pv + um[0][v] * qv;
        s2 += up[0][v] * pv + qm[1][v] * qv;
        s3 += up[1][v] * pv - qm[0][v] * qv;
    }
    g[0] += s0; g[1] += s1; c[0] += s2; c[1] += s3;
}

static void anal_blk(int spin, int lmax, int m, int first, const double *fp, const double *fm, double *almG, double *almC, blkacc *A)
{
    int l = first;
    if (l <= lmax && ((l + m + spin) & 1)) {
        anal_blk_step(spin, fp + (size_t)l * NV, fm + (size_t)l * NV, almG + 2 * l, almC ? almC + 2 * l : NULL, A->qo, A->qe, A->uo, A->ue);
        ++l;
    }
    for (; l + 1 <= lmax; l += 2) {
        anal_blk_step(spin, fp + (size_t)l * NV, fm + (size_t)l * NV, almG + 2 * l, almC ? almC + 2 * l : NULL, A->qe, A->qo, A->ue, A->uo);
        anal_blk_step(spin, fp + (size_t)(l + 1) * NV, fm + (size_t)(l + 1) * NV, almG + 2 * (l + 1), almC ? almC + 2 * (l + 1) : NULL,
                      A->qo, A->qe, A->uo, A->ue);
    }
    if (l <= lmax)
        anal_blk_step(spin, fp + (size_t)l * NV, fm + (size_t)l * NV, almG + 2 * l, almC ? almC + 2 * l : NULL, A->qe, A->qo, A->ue, A->uo);
}


/* ---- spin s > 0 without the Fp / Fm tables --------------------------------------------------------------------------------
 * With D- = sqrt((2l+1)/4pi) d^l_{m,-s}, D+ = ... d^l_{m,+s} (the two tables of rec_fill2), sg = (-1)^s:
 *   Fp = -(sg D- + D+) / 2, Fm = -(sg D- - D+) / 2,   and under theta -> pi - theta:  sg D- -> sigma D+, D+ -> sigma sg D-,
 *   sigma = (-1)^(l+m+s).  Substituting into Q = sum G Fp + i C Fm, U = sum C Fp - i G Fm:
 *   north:  X = sum_l A_l D-,  Y = sum_l B_l D+,   A = -sg (G + iC) / 2,  B = -(G - iC) / 2,   Q = X + Y,  U = i (Y - X)
 *   south:  X_S = sg sum_l sigma A_l D+,  Y_S = sg sum_l sigma B_l D-,  same combinations.
 * The analysis is the adjoint: T- = sum_rings [D- P_N + sigma sg D+ P_S], T+ = sum_rings [D+ M_N + sigma sg D- M_S] with P = Q + iU,
 * M = Q - iU, and G = -(sg T- + T+) / 2, C = -i (T+ - sg T-) / 2.  Same FMA count as the Fp / Fm form, no pass that builds Fp, Fm. */
typedef struct { double x[2][NV], y[2][NV], xe[2][NV], xo[2][NV], ye[2][NV], yo[2][NV]; } blkacc2;

static inline void synth2_step(const double *dm, const double *dp, const double *A, const double *B,
                               double (*X)[NV], double (*Y)[NV], double (*XS)[NV], double (*YS)[NV])
{
    const double ar = A[0], ai = A[1], br = B[0], bi = B[1];
#pragma omp simd
    for (int v = 0; v < NV; ++v) {
        const double m_ = dm[v], p_ = dp[v];
        X[0][v] += ar * m_; X[1][v] += ai * m_; Y[0][v] += br * p_; Y[1][v] += bi * p_;
        XS[0][v] += ar * p_; XS[1][v] += ai * p_; YS[0][v] += br * m_; YS[1][v] += bi * m_;
    }
}

static void synth_blk2(int spin, int lmax, int m, int first, const double *Dm, const double *Dp, const double *A, const double *B, blkacc2 *S)
{
    memset(S, 0, sizeof(*S));
    int l = first;
    if (l <= lmax && ((l + m + spin) & 1)) { synth2_step(Dm + (size_t)l * NV, Dp + (size_t)l * NV, A + 2 * l, B + 2 * l, S->x, S->y, S->xo, S->yo); ++l; }
    for (; l + 1 <= lmax; l += 2) {
        synth2_step(Dm + (size_t)l * NV, Dp + (size_t)l * NV, A + 2 * l, B + 2 * l, S->x, S->y, S->xe, S->ye);
        synth2_step(Dm + (size_t)(l + 1) * NV, Dp + (size_t)(l + 1) * NV, A + 2 * (l + 1), B + 2 * (l + 1), S->x, S->y, S->xo, S->yo);
    }
    if (l <= lmax) synth2_step(Dm + (size_t)l * NV, Dp + (size_t)l * NV, A + 2 * l, B + 2 * l, S->x, S->y, S->xe, S->ye);
}

/* inputs: pn = P_N, mn = M_N, ps = sg P_S, ms = sg M_S per ring ([re / im][ring]); sigma = +1 or -1 for this l */
static inline void anal2_step(const double *dm, const double *dp, double sigma, double (*pn)[NV], double (*mn)[NV], double (*ps)[NV], double (*ms)[NV],
                              double *Tm, double *Tp)
{
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma omp simd reduction(+ : s0, s1, s2, s3)
    for (int v = 0; v < NV; ++v) {
        const double m_ = dm[v], p_ = sigma * dp[v], q_ = dp[v], r_ = sigma * dm[v];
        s0 += m_ * pn[0][v] + p_ * ps[0][v];
        s1 += m_ * pn[1][v] + p_ * ps[1][v];
        s2 += q_ * mn[0][v] + r_ * ms[0][v];
        s3 += q_ * mn[1][v] + r_ * ms[1][v];
    }
    Tm[0] += s0; Tm[1] += s1; Tp[0] += s2; Tp[1] += s3;
}

static void anal_blk2(int spin, int lmax, int m, int first, const double *Dm, const double *Dp, double (*pn)[NV], double (*mn)[NV], double (*ps)[NV],
                      double (*ms)[NV], double *Tm, double *Tp)
{
    for (int l = first; l <= lmax; ++l)
        anal2_step(Dm + (size_t)l * NV, Dp + (size_t)l * NV, ((l + m + spin) & 1) ? -1.0 : 1.0, pn, mn, ps, ms, Tm + 2 * l, Tp + 2 * l);
}

/* direction: 0 = synthesis (alm -> phase), 1 = analysis (phase -> alm, accumulating into zeroed alm).
 * mode: 0 long double, 1 scaled double.  Returns 0. */
int orc_legendre(int direction, int mode, int spin, int lmax, int mmax, int nring,
                 const double *cth, const double *sth, const int *pair,
                 double *alm, double *phase, int nthreads)
{
    const int ncomp = spin == 0 ? 1 : 2;
    const int64_t nalm = alm_index(lmax, lmax, mmax) + 1;
    const int64_t nslot = 2 * (int64_t)nring, mstride = mmax + 1;
    if (direction == 1) memset(alm, 0, sizeof(double) * 2 * nalm * ncomp);
    else memset(phase, 0, sizeof(double) * 2 * nslot * mstride * ncomp);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int *mlim = malloc(sizeof(int) * nring);
    for (int r = 0; r < nring; ++r) mlim[r] = mode == 0 ? mmax : mlim_ring(lmax, spin, sth[r], cth[r]);

#pragma omp parallel
    {
        /* (64-byte aligned: a row of NV doubles is one cache line / one vector) */
        /* one block, the three tables 64-byte aligned and staggered by a few cache lines: equal offsets modulo 4 KB would make the two
         * table rows written per l (and the rows read back) alias in the store buffers */
        const size_t tabn = ((size_t)(lmax + 1) * NV + 7) & ~(size_t)7;
        double *tabs = aligned_alloc(64, sizeof(double) * (3 * tabn + 64 * 3));
        double *fp = tabs, *fm = tabs + tabn + 24, *tm = tabs + 2 * tabn + 88;
        double *Aa = malloc(sizeof(double) * 8 * (size_t)(lmax + 1)), *Ab = Aa + 2 * (lmax + 1), *Tm = Ab + 2 * (lmax + 1), *Tp = Tm + 2 * (lmax + 1);
        long double *scr = malloc(sizeof(long double) * 3 * (size_t)(lmax + 1));
        rectab tp, tn;
        tp.a = malloc(sizeof(double) * 3 * (size_t)(lmax + 2)); tp.b = tp.a + lmax + 2; tp.c = tp.b + lmax + 2;
        tn.a = malloc(sizeof(double) * 3 * (size_t)(lmax + 2)); tn.b = tn.a + lmax + 2; tn.c = tn.b + lmax + 2;
#pragma omp for schedule(dynamic, 1)
        for (int m = 0; m <= mmax; ++m) {
            double *aG = alm + 2 * alm_index(lmax, 0, m);
            double *aC = ncomp == 2 ? alm + 2 * nalm + 2 * alm_index(lmax, 0, m) : NULL;
            int l0 = m > spin ? m : spin;
            if (mode == 1) {
                rectab_build(&tp, m, -spin, lmax);
                if (spin) rectab_build(&tn, m, spin, lmax);
                if (spin && direction == 0) {
                    const double sg = (spin & 1) ? -1.0 : 1.0;
                    for (int l = l0; l <= lmax; ++l) {  /* A = -sg (G + iC) / 2, B = -(G - iC) / 2 */
                        const double gr = aG[2 * l], gi = aG[2 * l + 1], cr = aC[2 * l], ci = aC[2 * l + 1];
                        Aa[2 * l] = -0.5 * sg * (gr - ci); Aa[2 * l + 1] = -0.5 * sg * (gi + cr);
                        Ab[2 * l] = -0.5 * (gr + ci); Ab[2 * l + 1] = -0.5 * (gi - cr);
                    }
                } else if (spin) {
                    memset(Tm, 0, sizeof(double) * 2 * (size_t)(lmax + 1)); memset(Tp, 0, sizeof(double) * 2 * (size_t)(lmax + 1));
                }
            }
            /* mode 1, spin 0: ring blocks go through the recursion two at a time (rec_fill2 on two blocks; the second one's values
             * wait in `tm`); `held` = the block whose values are in tm, to be accumulated next */
            int held = -1, held_first = 0;
            for (int r0 = 0; r0 < nring; r0 += (mode == 1 ? NV : 1)) {
                int nv = 1, first = l0, stride = 1;
                double *lamtab = fp;
                if (mode == 0) {
                    get_lam_ld(spin, m, lmax, cth[r0], sth[r0], fp, fm, scr);
                } else {
                    nv = nring - r0 < NV ? nring - r0 : NV;
                    int anyact = 0;
                    for (int v = 0; v < nv; ++v) anyact |= (m <= mlim[r0 + v]);
                    if (!anyact) continue;
                    stride = NV;
                    if (held == r0) {  /* spin 0: this block was filled together with the previous one */
                        lamtab = tm; first = held_first; held = -1;
                    } else {
                        blkgeom g[2];
                        int nblk = 1;
                        if (spin == 0 && r0 + NV < nring) {  /* a second active block to ride along? */
                            int act2 = 0, nv2 = nring - (r0 + NV) < NV ? nring - (r0 + NV) : NV;
                            for (int v = 0; v < nv2; ++v) act2 |= (m <= mlim[r0 + NV + v]);
                            if (act2) nblk = 2;
                        }
                        for (int b = 0; b < nblk; ++b) {
                            const int rb = r0 + b * NV, nvb = nring - rb < NV ? nring - rb : NV;
                            g[b].nv = nvb;
                            for (int v = 0; v < NV; ++v) {
                                int r = rb + (v < nvb ? v : 0);
                                g[b].x[v] = cth[r]; g[b].st[v] = sth[r];
                                /* half-angle functions without cancellation: cos(th/2)^2 = (1+x)/2, sin(th/2) = sin(th) / (2 cos(th/2)) */
                                if (cth[r] >= 0) { g[b].ch[v] = sqrt(0.5 * (1.0 + cth[r])); g[b].sh[v] = 0.5 * sth[r] / g[b].ch[v]; }
                                else { g[b].sh[v] = sqrt(0.5 * (1.0 - cth[r])); g[b].ch[v] = 0.5 * sth[r] / g[b].sh[v]; }
                            }
                        }
                        int f1, f2 = lmax + 1;
                        if (spin) rec_fill2(&tp, &tn, lmax, &g[0], &g[0], fp, tm, &f1, &f2);   /* fp = D-, tm = D+ */
                        else if (nblk == 2) { rec_fill2(&tp, &tp, lmax, &g[0], &g[1], fp, tm, &f1, &held_first); held = r0 + NV; }
                        else f1 = rec_fill(&tp, lmax, g[0].x, g[0].st, g[0].ch, g[0].sh, nv, fp);
                        first = spin ? (f1 < f2 ? f1 : f2) : f1;
                    }
                    if (first > lmax) continue;
                }
                if (mode == 1 && spin > 0) {  /* the NV rings of the block together, on D- and D+ directly (synth_blk2 / anal_blk2) */
                    const int64_t cs = 2 * nslot * mstride;  /* component stride of the phase array, in doubles */
                    const double sg = (spin & 1) ? -1.0 : 1.0;
                    if (direction == 1) {
                        double pn[2][NV], mn[2][NV], ps[2][NV], ms[2][NV];
                        memset(pn, 0, sizeof(pn)); memset(mn, 0, sizeof(mn)); memset(ps, 0, sizeof(ps)); memset(ms, 0, sizeof(ms));
                        for (int v = 0; v < nv; ++v) {
                            int r = r0 + v;
                            if (m > mlim[r]) continue;
                            const double *qn = phase + 2 * ((2 * (int64_t)r) * mstride + m), *un = qn + cs;
                            pn[0][v] = qn[0] - un[1]; pn[1][v] = qn[1] + un[0];   /* P = Q + i U */
                            mn[0][v] = qn[0] + un[1]; mn[1][v] = qn[1] - un[0];   /* M = Q - i U */
                            if (pair[r]) {
                                const double *qs = phase + 2 * ((2 * (int64_t)r + 1) * mstride + m), *us = qs + cs;
                                ps[0][v] = sg * (qs[0] - us[1]); ps[1][v] = sg * (qs[1] + us[0]);
                                ms[0][v] = sg * (qs[0] + us[1]); ms[1][v] = sg * (qs[1] - us[0]);
                            }
                        }
                        anal_blk2(spin, lmax, m, first, fp, tm, pn, mn, ps, ms, Tm, Tp);
                    } else {
                        blkacc2 S;
                        synth_blk2(spin, lmax, m, first, fp, tm, Aa, Ab, &S);
                        for (int v = 0; v < nv; ++v) {
                            int r = r0 + v;
                            if (m > mlim[r]) continue;
                            double *qn = phase + 2 * ((2 * (int64_t)r) * mstride + m), *un = qn + cs;
                            qn[0] = S.x[0][v] + S.y[0][v]; qn[1] = S.x[1][v] + S.y[1][v];                 /* Q = X + Y */
                            un[0] = -(S.y[1][v] - S.x[1][v]); un[1] = S.y[0][v] - S.x[0][v];              /* U = i (Y - X) */
                            if (pair[r]) {
                                double *qs = phase + 2 * ((2 * (int64_t)r + 1) * mstride + m), *us = qs + cs;
                                const double xr = sg * (S.xe[0][v] - S.xo[0][v]), xi = sg * (S.xe[1][v] - S.xo[1][v]);
                                const double yr = sg * (S.ye[0][v] - S.yo[0][v]), yi = sg * (S.ye[1][v] - S.yo[1][v]);
                                qs[0] = xr + yr; qs[1] = xi + yi; us[0] = -(yi - xi); us[1] = yr - xr;
                            }
                        }
                    }
                    continue;
                }
                if (mode == 1) {  /* spin 0: the NV rings of the block together (synth_blk / anal_blk) */
                    blkacc A;
                    const int64_t cs = 2 * nslot * mstride;  /* component stride of the phase array, in doubles */
                    if (direction == 1) {
                        memset(&A, 0, sizeof(A));
                        for (int v = 0; v < nv; ++v) {
                            int r = r0 + v;
                            if (m > mlim[r]) continue;
                            const double *pn = phase + 2 * ((2 * (int64_t)r) * mstride + m);
                            const double *ps = pair[r] ? phase + 2 * ((2 * (int64_t)r + 1) * mstride + m) : NULL;
                            for (int k = 0; k < 2; ++k) {
                                double qn = pn[k], qs = ps ? ps[k] : 0.0;
                                A.qe[k][v] = qn + qs; A.qo[k][v] = qn - qs;
                                if (ncomp == 2) { double un = pn[cs + k], us = ps ? ps[cs + k] : 0.0; A.ue[k][v] = un + us; A.uo[k][v] = un - us; }
                            }
                        }
                        anal_blk(spin, lmax, m, first, lamtab, fm, aG, aC, &A);
                    } else {
                        synth_blk(spin, lmax, m, first, lamtab, fm, aG, aC, &A);
                        for (int v = 0; v < nv; ++v) {
                            int r = r0 + v;
                            if (m > mlim[r]) continue;
                            double *pn = phase + 2 * ((2 * (int64_t)r) * mstride + m);
                            double *ps = pair[r] ? phase + 2 * ((2 * (int64_t)r + 1) * mstride + m) : NULL;
                            for (int k = 0; k < 2; ++k) {
                                pn[k] = A.qe[k][v] + A.qo[k][v];
                                if (ps) ps[k] = A.qe[k][v] - A.qo[k][v];
                                if (ncomp == 2) { pn[cs + k] = A.ue[k][v] + A.uo[k][v]; if (ps) ps[cs + k] = A.ue[k][v] - A.uo[k][v]; }
                            }
                        }
                    }
                    continue;
                }
                for (int v = 0; v < nv; ++v) {
                    int r = r0 + v;
                    double *pq_n = phase + 2 * ((2 * (int64_t)r) * mstride + m);
                    double *pq_s = pair[r] ? phase + 2 * ((2 * (int64_t)r + 1) * mstride + m) : NULL;
                    double *pu_n = ncomp == 2 ? pq_n + 2 * nslot * mstride : NULL;
                    double *pu_s = (ncomp == 2 && pq_s) ? pq_s + 2 * nslot * mstride : NULL;
                    if (direction == 0) {
                        double du[2], dus[2];
                        synth_one(spin, lmax, m, first, fp + v, spin ? fm + v : NULL, stride, aG, aC,
                                  pq_n, pu_n ? pu_n : du, pq_s ? pq_s : NULL, pq_s ? (pu_s ? pu_s : dus) : NULL);
                    } else {
                        anal_one(spin, lmax, m, first, fp + v, spin ? fm + v : NULL, stride, aG, aC,
                                 pq_n, pu_n, pq_s, pu_s);
                    }
                }
            }
            if (mode == 1 && spin && direction == 1) {  /* G = -(sg T- + T+) / 2, C = -i (T+ - sg T-) / 2 */
                const double sg = (spin & 1) ? -1.0 : 1.0;
                for (int l = l0; l <= lmax; ++l) {
                    const double tmr = sg * Tm[2 * l], tmi = sg * Tm[2 * l + 1], tpr = Tp[2 * l], tpi = Tp[2 * l + 1];
                    aG[2 * l] = -0.5 * (tmr + tpr); aG[2 * l + 1] = -0.5 * (tmi + tpi);
                    aC[2 * l] = 0.5 * (tpi - tmi); aC[2 * l + 1] = -0.5 * (tpr - tmr);
                }
            }
        }
        free(tabs); free(scr); free(tp.a); free(tn.a); free(Aa);
    }
    free(mlim);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Fourier stage in C (the numpy route of sht_oracle.py stays the default of the tests; this one   */
/* is threaded over rings for bench.py's cpu_baseline and is checked against the numpy route)      */
/* ------------------------------------------------------------------------------------------- */
/* In-place complex FFT of power-of-two length n (iterative radix 2); sign = -1 forward, +1 inverse (unnormalised).
 * tw: per-stage twiddles, stage of butterfly span `half` at tw[half + k] = e^{-2 pi i k / (2 half)}, k < half (contiguous in k, so
 * the butterfly loop vectorises). */
typedef struct { int n; double *r, *i; } twtab;

static void twtab_make(twtab *t, int n)
{
    t->n = n;
    t->r = malloc(sizeof(double) * 2 * (size_t)(n > 1 ? n : 2)); t->i = t->r + (n > 1 ? n : 2);
    for (int half = 1; half < n; half <<= 1)
        for (int k = 0; k < half; ++k) { double a = -(double)ORC_PI * k / half; t->r[half + k] = cos(a); t->i[half + k] = sin(a); }
}

static void fft_pow2(double *re, double *im, int n, int sign, const twtab *t)
{
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double x = re[i]; re[i] = re[j]; re[j] = x; x = im[i]; im[i] = im[j]; im[j] = x; }
    }
    const double sg = sign < 0 ? 1.0 : -1.0;
    for (int half = 1; half < n; half <<= 1) {
        const double *wr = t->r + half, *wi = t->i + half;
        for (int i = 0; i < n; i += 2 * half) {
            double *ar = re + i, *ai = im + i, *br = re + i + half, *bi = im + i + half;
#pragma omp simd
            for (int k = 0; k < half; ++k) {
                const double c = wr[k], d = sg * wi[k];
                const double xr = br[k] * c - bi[k] * d, xi = br[k] * d + bi[k] * c;
                br[k] = ar[k] - xr; bi[k] = ai[k] - xi;
                ar[k] += xr; ai[k] += xi;
            }
        }
    }
}

/* per-thread tables: twiddles by transform size (a handful of powers of two), Bluestein chirp and filter spectrum of the current n */
typedef struct { int n, M; twtab tw[32]; double *cr, *ci, *fr, *fi, *ar, *ai; int cap; } ringfft_plan;

static const twtab *plan_tw(ringfft_plan *p, int M)
{
    int lg = 0;
    while ((1 << lg) < M) ++lg;
    if (p->tw[lg].r == NULL) twtab_make(&p->tw[lg], M);
    return &p->tw[lg];
}

/* plan for length n: direct when n is a power of two, otherwise Bluestein (chirp e^{-i pi k^2 / n}, convolution size M >= 2n - 1) */
static void ringfft_make(ringfft_plan *p, int n)
{
    p->n = n;
    int M = 1;
    if ((n & (n - 1)) == 0) M = n; else { while (M < 2 * n - 1) M <<= 1; }
    p->M = M;
    if (M == n) return;
    if (M > p->cap) {
        free(p->cr);
        p->cr = malloc(sizeof(double) * 6 * (size_t)M); p->cap = M;
    }
    p->ci = p->cr + p->cap; p->fr = p->ci + p->cap; p->fi = p->fr + p->cap; p->ar = p->fi + p->cap; p->ai = p->ar + p->cap;
    for (int k = 0; k < n; ++k) {
        long long k2 = ((long long)k * k) % (2LL * n);
        double a = (double)ORC_PI * (double)k2 / n;
        p->cr[k] = cos(a); p->ci[k] = -sin(a);   /* chirp w_k = e^{-i pi k^2 / n} */
    }
    for (int k = 0; k < M; ++k) { p->fr[k] = 0; p->fi[k] = 0; }
    for (int k = 0; k < n; ++k) {                 /* filter conj(w) wrapped, transformed once */
        p->fr[k] = p->cr[k]; p->fi[k] = -p->ci[k];
        if (k) { p->fr[M - k] = p->cr[k]; p->fi[M - k] = -p->ci[k]; }
    }
    fft_pow2(p->fr, p->fi, M, -1, plan_tw(p, M));
}

static void ringfft_free(ringfft_plan *p)
{
    free(p->cr); p->cr = NULL; p->cap = 0;
    for (int i = 0; i < 32; ++i) { free(p->tw[i].r); p->tw[i].r = NULL; }
}

/* X_k = sum_j x_j e^{sign 2 pi i jk/n}, in place on (re, im) of length n; sign = -1 forward, +1 inverse (unnormalised) */
static void ringfft_run(ringfft_plan *p, double *re, double *im, int sign)
{
    const int n = p->n, M = p->M;
    if (M == n) { fft_pow2(re, im, n, sign, plan_tw(p, n)); return; }
    const twtab *t = plan_tw(p, M);
    if (sign > 0) for (int k = 0; k < n; ++k) im[k] = -im[k];   /* inverse = conj(forward(conj(x))) */
    for (int k = 0; k < n; ++k) { p->ar[k] = re[k] * p->cr[k] - im[k] * p->ci[k]; p->ai[k] = re[k] * p->ci[k] + im[k] * p->cr[k]; }
    for (int k = n; k < M; ++k) { p->ar[k] = 0; p->ai[k] = 0; }
    fft_pow2(p->ar, p->ai, M, -1, t);
    for (int k = 0; k < M; ++k) { double r = p->ar[k] * p->fr[k] - p->ai[k] * p->fi[k]; p->ai[k] = p->ar[k] * p->fi[k] + p->ai[k] * p->fr[k]; p->ar[k] = r; }
    fft_pow2(p->ar, p->ai, M, +1, t);
    const double inv = 1.0 / M;
    for (int k = 0; k < n; ++k) {
        double r = p->ar[k] * inv, i = p->ai[k] * inv;
        re[k] = r * p->cr[k] - i * p->ci[k]; im[k] = r * p->ci[k] + i * p->cr[k];
    }
    if (sign > 0) for (int k = 0; k < n; ++k) im[k] = -im[k];
}

/* direction 0: phase[slot][m] -> ring pixels (x_j = sum_m w_m Re(F_m e^{i m phi_j}), aliasing folded explicitly);
 * direction 1: ring pixels -> phase[slot][m] = (4 pi / npix) sum_j x_j e^{-i m phi_j}.
 * ring[slot] = ring index (0 .. 4 nside - 2) or -1; nphi / phi0 / ofs per ring index as sht_oracle.ring_geometry.
 * Slots 2 i and 2 i + 1 that hold two rings of the same length and offset angle (a ring and its mirror) go through ONE complex
 * transform, z = x_a + i x_b: both are real, so the two spectra separate by Hermitian symmetry. */
int orc_ring_fft(int direction, int64_t npix, int mmax, int nslot, const int64_t *ring, const int64_t *nphi, const double *phi0,
                 const int64_t *ofs, double *phase, double *map, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const double w = 4.0 * (double)ORC_PI / (double)npix;
    const int npairs = (nslot + 1) / 2;
#pragma omp parallel
    {
        ringfft_plan pl; memset(&pl, 0, sizeof(pl));
        double *re = NULL, *im = NULL, *pc = malloc(sizeof(double) * 2 * (size_t)(mmax + 1)), *ps = pc + mmax + 1; int cap = 0;
#pragma omp for schedule(dynamic, 2)
        for (int ip = 0; ip < npairs; ++ip) {
            int sa = 2 * ip, sb = 2 * ip + 1;
            int64_t ra = ring[sa], rb = sb < nslot ? ring[sb] : -1;
            if (ra < 0 && rb < 0) continue;
            if (ra < 0) { ra = rb; sa = sb; rb = -1; }
            const int two = rb >= 0 && nphi[rb] == nphi[ra] && phi0[rb] == phi0[ra];
            for (int pass = 0; pass < (two || rb < 0 ? 1 : 2); ++pass) {  /* (rings that do not match go one by one) */
                const int64_t r = pass == 0 ? ra : rb, r2 = two ? rb : -1;
                const int s1 = pass == 0 ? sa : sb;
                const int n = (int)nphi[r];
                if (pl.n != n) ringfft_make(&pl, n);
                if (n > cap) { free(re); re = malloc(sizeof(double) * 2 * (size_t)n); im = re + n; cap = n; } else im = re + cap;
                double *ph = phase + 2 * (int64_t)s1 * (mmax + 1), *ph2 = two ? phase + 2 * (int64_t)sb * (mmax + 1) : NULL;
                double *px = map + ofs[r], *px2 = two ? map + ofs[r2] : NULL;
                const int shifted = phi0[r] != 0.0;
                if (shifted) for (int m = 0; m <= mmax; ++m) { double a = phi0[r] * m; pc[m] = cos(a); ps[m] = sin(a); }
                if (direction == 0) {
                    for (int k = 0; k < n; ++k) { re[k] = 0; im[k] = 0; }
                    for (int m = 0; m <= mmax; ++m) {
                        const double c = shifted ? pc[m] : 1.0, sn = shifted ? ps[m] : 0.0;
                        double fr = ph[2 * m] * c - ph[2 * m + 1] * sn, fi = ph[2 * m] * sn + ph[2 * m + 1] * c;
                        const int k = m % n, kn = (n - k) % n;
                        re[k] += fr; im[k] += fi;
                        if (m) { re[kn] += fr; im[kn] -= fi; }
                        if (two) {  /* + i (spectrum of the second ring) */
                            double gr = ph2[2 * m] * c - ph2[2 * m + 1] * sn, gi = ph2[2 * m] * sn + ph2[2 * m + 1] * c;
                            re[k] -= gi; im[k] += gr;
                            if (m) { re[kn] += gi; im[kn] += gr; }
                        }
                    }
                    ringfft_run(&pl, re, im, +1);
                    for (int j = 0; j < n; ++j) px[j] = re[j];
                    if (two) for (int j = 0; j < n; ++j) px2[j] = im[j];
                } else {
                    for (int j = 0; j < n; ++j) { re[j] = px[j]; im[j] = two ? px2[j] : 0.0; }
                    ringfft_run(&pl, re, im, -1);
                    for (int m = 0; m <= mmax; ++m) {
                        const double c = shifted ? pc[m] : 1.0, sn = shifted ? -ps[m] : 0.0;
                        const int k = m % n, kn = (n - k) % n;
                        double zr = re[k], zi = im[k];
                        if (two) {  /* X_a = (Z_k + conj Z_{n-k}) / 2, X_b = (Z_k - conj Z_{n-k}) / (2 i) */
                            const double yr = re[kn], yi = im[kn];
                            const double ar = 0.5 * (zr + yr), ai = 0.5 * (zi - yi), br = 0.5 * (zi + yi), bi = 0.5 * (yr - zr);
                            ph2[2 * m] = (br * c - bi * sn) * w; ph2[2 * m + 1] = (br * sn + bi * c) * w;
                            zr = ar; zi = ai;
                        }
                        ph[2 * m] = (zr * c - zi * sn) * w; ph[2 * m + 1] = (zr * sn + zi * c) * w;
                    }
                }
            }
        }
        ringfft_free(&pl);
        free(re); free(pc);
    }
    return 0;
}
