// Internal declarations shared by the host-side table builder and the HIP kernels.
#pragma once
#include <cstdint>
#include <vector>

namespace plshts {

constexpr int kMaxSpin = 3;

// Development knobs of the launchers (kernel-variant fallbacks, stream counts, rings per lane: INTEGRATION.md section 4) are read from the
// environment ONLY when PLSHTS_DEBUG is set to a non-zero value; a production process has none of them.  Returns NULL otherwise.
const char *dbg_env(const char *name);
int dbg_env_int(const char *name, int dflt);

// ---- host-side tables (tables.cpp), built in long double and rounded once ---------------------------
struct Spin0Tables {
    // Two-step recursion of the spin-0 Legendre functions for every m (see DESIGN.md "Legendre kernels"):
    //   P_{il+1} = (A_il x^2 + B_il) P_il - P_{il-1},  l = m + 2 il,  M_l := lambda_{l+1,m} / x = alpha_l P_il
    std::vector<int64_t> off;      // [mmax + 2] entry offsets per m (nil(m) = (lmax - m) / 2 + 1 entries)
    std::vector<double> ab;        // 2 per entry: A, B
    std::vector<double> alpha;     // 1 per entry
    std::vector<double> eps;       // 2 per entry: eps_{l+1}, eps_{l+2}  (alm <-> (c,d) transforms)
    std::vector<double> seed;      // [mmax + 1]: P_0 = seed[m] * sin^m(theta)
};

struct SpinTables {
    // One-step recursion of S_l = sqrt((2l+1)/4pi) d^l_{m,+-s} / beta_l:
    //   S^{+-}_{l+1} = (x a_l -+ b_l) S^{+-}_l - S^{+-}_{l-1},  l >= l0 = max(m, s)
    int spin = 0;
    std::vector<int64_t> off;      // [mmax + 2] entry offsets per m (lmax - l0 + 1 entries, 0 if l0 > lmax)
    std::vector<double> ab;        // 2 per entry: a_l, b_l (b for n = +s)
    std::vector<double> beta;      // 1 per entry
    // seeds at l0: S^{+-}_{l0} = seedfac * sin^{psin}(theta) * (cos or sin)^{phalf}(theta / 2)
    std::vector<double> seedfac_n, seedfac_p;  // n = -s, n = +s
    std::vector<int> psin, phalf, usecos_n, usecos_p;
};

void build_spin0_tables(int lmax, int mmax, Spin0Tables &t);
void build_spin_tables(int spin, int lmax, int mmax, SpinTables &t);

// libsharp's polar-optimisation bound (orders above it contribute < 1e-30 on the ring)
int mlim_ring(int lmax, int spin, double sth, double cth);

struct RingGeom {
    int nside = 0, npairs = 0;
    std::vector<double> cth, sth, chalf, shalf, phi0;  // [npairs] north member of each pair
    std::vector<int> nphi;                             // [npairs]
    std::vector<int64_t> ofs_n, ofs_s;                 // first pixel of north / south ring (-1: no partner)
};
void build_geometry(int nside, RingGeom &g);

}  // namespace plshts
