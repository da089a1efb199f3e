// Host-side construction of the plan tables (long double, rounded once to double).
// Geometry: public HEALPix RING definition.  Recursions: DESIGN.md.
#include "plshts_internal.h"

#include <cmath>
#include <cstdlib>

namespace plshts {

static const long double kPi = 3.14159265358979323846264338327950288L;

void build_geometry(int nside, RingGeom &g)
{
    g.nside = nside;
    g.npairs = 2 * nside;  // rings 1 .. 2 nside; the last one (equator) has no partner
    const int np = g.npairs;
    g.cth.resize(np); g.sth.resize(np); g.chalf.resize(np); g.shalf.resize(np); g.phi0.resize(np);
    g.nphi.resize(np); g.ofs_n.resize(np); g.ofs_s.resize(np);
    const int64_t npix = 12LL * nside * nside;
    const int64_t ncap = 2LL * nside * (nside - 1);
    for (int ip = 0; ip < np; ++ip) {
        const int i = ip + 1;  // ring number, north hemisphere incl. equator
        long double omz, z;    // 1 - z kept separately: no cancellation in the caps
        int nphi; bool shifted; int64_t ofs;
        if (i < nside) {
            omz = (long double)i * i / (3.0L * nside * nside);
            nphi = 4 * i; shifted = true; ofs = 2LL * i * (i - 1);
        } else {
            omz = 1.0L - (4.0L / 3.0L - 2.0L * i / (3.0L * nside));
            nphi = 4 * nside; shifted = ((i - nside) & 1) == 0; ofs = ncap + 4LL * nside * (i - nside);
        }
        z = 1.0L - omz;
        long double s = sqrtl(omz * (1.0L + z));
        g.cth[ip] = (double)z; g.sth[ip] = (double)s;
        long double ch = sqrtl(0.5L * (1.0L + z));
        g.chalf[ip] = (double)ch; g.shalf[ip] = (double)(0.5L * s / ch);
        g.nphi[ip] = nphi;
        g.phi0[ip] = shifted ? (double)(kPi / nphi) : 0.0;
        g.ofs_n[ip] = ofs;
        g.ofs_s[ip] = (i == 2 * nside) ? -1 : npix - ofs - nphi;
    }
}

int mlim_ring(int lmax, int spin, double sth, double cth)
{
    double ofs = lmax * 0.01;
    if (ofs < 100.) ofs = 100.;
    double b = -2 * spin * std::fabs(cth);
    double t1 = lmax * sth + ofs;
    double c = (double)spin * spin - t1 * t1;
    double discr = b * b - 4 * c;
    if (discr <= 0) return lmax;
    double res = (-b + std::sqrt(discr)) / 2.;
    if (res > lmax) res = lmax;
    return (int)(res + 0.5);
}

void build_spin0_tables(int lmax, int mmax, Spin0Tables &t)
{
    t.off.assign(mmax + 2, 0);
    for (int m = 0; m <= mmax; ++m) t.off[m + 1] = t.off[m] + ((lmax - m) / 2 + 1);
    const int64_t ntot = t.off[mmax + 1];
    t.ab.assign(2 * ntot, 0.0); t.alpha.assign(ntot, 0.0); t.eps.assign(2 * ntot, 0.0); t.seed.assign(mmax + 1, 0.0);
    long double mfac = 1.0L / sqrtl(4.0L * kPi);
    for (int m = 0; m <= mmax; ++m) {
        if (m > 0) mfac = -mfac * sqrtl((2.0L * m + 1) / (2.0L * m));
        t.seed[m] = (double)(mfac * sqrtl(2.0L * m + 3));  // lambda_mm / eps_{m+1}
        auto eps = [m](int l) -> long double {
            return sqrtl(((long double)l * l - (long double)m * m) / (4.0L * l * l - 1.0L));
        };
        const int nil = (lmax - m) / 2 + 1;
        long double alpha = 1.0L;
        for (int il = 0; il < nil; ++il) {
            const int l = m + 2 * il;
            const int64_t e = t.off[m] + il;
            long double e1 = eps(l + 1), e2 = eps(l + 2), e3 = eps(l + 3);
            t.alpha[e] = (double)alpha;
            t.ab[2 * e] = (double)(alpha * alpha);
            t.ab[2 * e + 1] = (double)(-(e2 * e2 + e1 * e1) * alpha * alpha);
            t.eps[2 * e] = (double)e1; t.eps[2 * e + 1] = (double)e2;
            alpha = 1.0L / (alpha * e2 * e3);
        }
    }
}

void build_spin_tables(int spin, int lmax, int mmax, SpinTables &t)
{
    t.spin = spin;
    t.off.assign(mmax + 2, 0);
    for (int m = 0; m <= mmax; ++m) {
        int l0 = m > spin ? m : spin;
        t.off[m + 1] = t.off[m] + (l0 <= lmax ? lmax - l0 + 1 : 0);
    }
    const int64_t ntot = t.off[mmax + 1];
    t.ab.assign(2 * ntot, 0.0); t.beta.assign(ntot, 0.0);
    t.seedfac_n.assign(mmax + 1, 0.0); t.seedfac_p.assign(mmax + 1, 0.0);
    t.psin.assign(mmax + 1, 0); t.phalf.assign(mmax + 1, 0); t.usecos_n.assign(mmax + 1, 0); t.usecos_p.assign(mmax + 1, 0);
    for (int m = 0; m <= mmax; ++m) {
        const int j = m > spin ? m : spin;
        if (j > lmax) continue;
        const long double M = m, N = spin;
        long double beta_prev = 1.0L, beta_cur = 1.0L;  // beta_{l-1}, beta_l
        for (int l = j; l <= lmax; ++l) {
            const int64_t e = t.off[m] + (l - j);
            const long double L = l, L1 = l + 1;
            long double den = L * sqrtl((L1 * L1 - M * M) * (L1 * L1 - N * N));
            long double nrm1 = sqrtl((2 * L1 + 1) / (2 * L + 1));
            long double a = nrm1 * (2 * L + 1) * L * L1 / den;
            long double b = nrm1 * (2 * L + 1) * M * N / den;
            long double c = (l == j) ? 0.0L
                                     : sqrtl((2 * L1 + 1) / (2 * L - 1)) * L1 * sqrtl((L * L - M * M) * (L * L - N * N)) / den;
            long double beta_next = (l == j) ? 1.0L : c * beta_prev;
            t.beta[e] = (double)beta_cur;
            t.ab[2 * e] = (double)(a * beta_cur / beta_next);
            t.ab[2 * e + 1] = (double)(b * beta_cur / beta_next);
            beta_prev = beta_cur; beta_cur = beta_next;
        }
        // seeds: sqrt((2j+1)/4pi) d^j_{m,n} = sign * [sqrt((2j+1)/4pi) sqrt(C(2j,j+k)) 2^-(j-|k|)] sin^{j-|k|}(th) (cos|sin)^{2|k|}(th/2)
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            const int n = sgn * spin;
            int k; long double sign = 1.0L;
            if (j == m)      { k = n; }
            else if (j == n) { k = m; if ((j - m) & 1) sign = -1.0L; }
            else             { k = -m; }
            if ((j - k) & 1) sign = -sign;
            const int ak = k < 0 ? -k : k;
            long double lg = 0.5L * (lgammal(2.0L * j + 1) - lgammal((long double)(j + k) + 1) - lgammal((long double)(j - k) + 1))
                             - (long double)(j - ak) * logl(2.0L);
            double fac = (double)(sign * sqrtl((2.0L * j + 1) / (4.0L * kPi)) * expl(lg));
            t.psin[m] = j - ak; t.phalf[m] = 2 * ak;
            if (sgn < 0) { t.seedfac_n[m] = fac; t.usecos_n[m] = k > 0; }
            else         { t.seedfac_p[m] = fac; t.usecos_p[m] = k > 0; }
        }
    }
}

const char *dbg_env(const char *name)
{
    static const bool on = [] { const char *v = getenv("PLSHTS_DEBUG"); return v && atoi(v) != 0; }();
    return on ? getenv(name) : nullptr;
}

int dbg_env_int(const char *name, int dflt)
{
    const char *v = dbg_env(name);
    return v ? atoi(v) : dflt;
}

}  // namespace plshts
