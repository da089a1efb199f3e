import sys, numpy as np
sys.path.insert(0, '.')
from plancklens_amd import shts, hp
rng = np.random.default_rng(0)
for (nside, lmax) in [(16, 32), (16, 47), (8, 16), (16, 31), (16, 33)]:
    for spin in (2, 1):
        m = rng.standard_normal((2, 12*nside**2))
        for rep in range(3):
            g, c = shts.map2alm_spin(m, spin, lmax)
            bad = np.nonzero(~np.isfinite(g) | ~np.isfinite(c))[0]
            if bad.size:
                l, mm = hp.Alm.getlm(lmax, bad)
                print(nside, lmax, spin, rep, 'NaN count', bad.size, 'l', l[:10], 'm', mm[:10], flush=True)
            else:
                print(nside, lmax, spin, rep, 'ok', flush=True)
        # synth then anal
        a = [rng.standard_normal(hp.Alm.getsize(lmax)) + 0j, rng.standard_normal(hp.Alm.getsize(lmax)) + 0j]
        q, u = shts.alm2map_spin(a, nside, spin, lmax)
        print('   synth finite', np.isfinite(q).all() and np.isfinite(u).all())
