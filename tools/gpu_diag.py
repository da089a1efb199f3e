"""Stage-by-stage GPU-vs-oracle diagnostics (development aid; run on the GPU box through gpurun)."""
import ctypes
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, '.')
from oracle import sht_oracle as so
from plancklens_amd import _lib, shts

L = _lib.lib()


def relrms(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / max(np.sum(np.abs(b) ** 2), 1e-300)))


def ralm(rng, lmax, lmin=0):
    n = so.alm_size(lmax)
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    a[:lmax + 1] = a[:lmax + 1].real
    ls = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)])
    a[ls < lmin] = 0
    return a


def gpu_phase_to_oracle(ph, npairs, mstride, ncomp, lmax):
    """[pair][ncomp][mstride][4] doubles -> [comp][slot=2*pair][m] complex"""
    p = ph.reshape(npairs, ncomp, mstride, 4).transpose(0, 2, 1, 3)[:, :lmax + 1]
    out = np.zeros((ncomp, 2 * npairs, lmax + 1), dtype=complex)
    for c in range(ncomp):
        out[c, 0::2] = p[:, :, c, 0] + 1j * p[:, :, c, 1]
        out[c, 1::2] = p[:, :, c, 2] + 1j * p[:, :, c, 3]
    return out


def oracle_phase_to_gpu(po, npairs, mstride, ncomp, lmax):
    p = np.zeros((npairs, mstride, ncomp, 4))
    for c in range(ncomp):
        p[:, :lmax + 1, c, 0] = po[c, 0::2].real
        p[:, :lmax + 1, c, 1] = po[c, 0::2].imag
        p[:, :lmax + 1, c, 2] = po[c, 1::2].real
        p[:, :lmax + 1, c, 3] = po[c, 1::2].imag
    return np.ascontiguousarray(p.transpose(0, 2, 1, 3))  # device layout [pair][ncomp][mstride][4]


def stage_tests(nside, lmax, spins=(0, 1, 2, 3)):
    rng = np.random.default_rng(nside * 1000 + lmax)
    plan = shts.get_plan(nside, lmax)
    npairs = 2 * nside
    mstride = (lmax + 1 + 3) // 4 * 4
    c, s, pair, slots = so._pair_geometry(nside, True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for spin in spins:
        ncomp = 1 if spin == 0 else 2
        try:
            alm = np.stack([ralm(rng, lmax, spin) for _ in range(ncomp)])
            po = so.legendre(0, 1, spin, lmax, lmax, c, s, pair, alm=alm)
            # equator slot: oracle leaves the (unused) south slot at zero; the GPU stores F_N there
            nd = plan.phase_doubles(spin)
            d_alm = torch.from_numpy(alm.copy()).cuda()
            d_ph = torch.zeros(nd, dtype=torch.float64, device='cuda')
            _lib.check(L.pl_legendre_synth(plan.h, spin, d_alm.data_ptr(), None, d_ph.data_ptr(), st))
            torch.cuda.synchronize()
            pg = gpu_phase_to_oracle(d_ph.cpu().numpy(), npairs, mstride, ncomp, lmax)
            pg[:, -1, :] = 0
            print('nside %d lmax %d spin %d  legendre_synth relrms %.3e' % (nside, lmax, spin, relrms(pg, po)), flush=True)
            # phase2map from oracle phases
            maps_o = np.stack([so._phase2map(po[i], nside, lmax, slots) for i in range(ncomp)])
            d_ph2 = torch.from_numpy(oracle_phase_to_gpu(po, npairs, mstride, ncomp, lmax).ravel().copy()).cuda()
            d_map = torch.zeros(ncomp * 12 * nside ** 2, dtype=torch.float64, device='cuda')
            _lib.check(L.pl_phase2map(plan.h, spin, d_ph2.data_ptr(), d_map.data_ptr(), st))
            torch.cuda.synchronize()
            mg = d_map.cpu().numpy().reshape(ncomp, -1)
            print('   phase2map relrms %.3e' % relrms(mg, maps_o), flush=True)
            # map2phase
            maps_in = rng.standard_normal((ncomp, 12 * nside ** 2))
            pho = np.stack([so._map2phase(maps_in[i], nside, lmax, slots) for i in range(ncomp)])
            d_min = torch.from_numpy(maps_in.copy()).cuda()
            d_ph3 = torch.zeros(nd, dtype=torch.float64, device='cuda')
            _lib.check(L.pl_map2phase(plan.h, spin, d_min.data_ptr(), d_ph3.data_ptr(), st))
            torch.cuda.synchronize()
            pg3 = gpu_phase_to_oracle(d_ph3.cpu().numpy(), npairs, mstride, ncomp, lmax)
            # compare only m <= mlim of each ring: beyond it the GPU leaves the entries untouched
            ml = np.array([so_mlim(lmax, spin, s[i], c[i]) for i in range(npairs)])
            mask = (np.arange(lmax + 1)[None, :] <= np.repeat(ml, 2)[:, None])
            print('   map2phase relrms %.3e' % relrms(pg3 * mask, pho * mask), flush=True)
            # legendre analysis from oracle phases
            ao = so.legendre(1, 1, spin, lmax, lmax, c, s, pair, phase=pho)
            d_ph4 = torch.from_numpy(oracle_phase_to_gpu(pho, npairs, mstride, ncomp, lmax).ravel().copy()).cuda()
            d_aout = torch.zeros(ncomp * plan.nalm, dtype=torch.complex128, device='cuda')
            _lib.check(L.pl_legendre_anal(plan.h, spin, d_ph4.data_ptr(), d_aout.data_ptr(), None, st))
            torch.cuda.synchronize()
            ag = d_aout.cpu().numpy().reshape(ncomp, -1)
            print('   legendre_anal relrms %.3e' % relrms(ag, ao), flush=True)
        except Exception:
            traceback.print_exc()


def so_mlim(lmax, spin, sth, cth):
    ofs = max(lmax * 0.01, 100.)
    b = -2 * spin * abs(cth)
    t1 = lmax * sth + ofs
    cc = spin * spin - t1 * t1
    discr = b * b - 4 * cc
    if discr <= 0:
        return lmax
    return int(min((-b + np.sqrt(discr)) / 2., lmax) + 0.5)


def full_tests(nside, lmax, spins=(0, 1, 2, 3), timing=False):
    rng = np.random.default_rng(7)
    npix = 12 * nside ** 2
    for spin in spins:
        try:
            if spin == 0:
                a = ralm(rng, lmax)
                t0 = time.time(); mo = so.alm2map(a, nside, lmax=lmax); t_or = time.time() - t0
                mg = shts.alm2map(a, nside, lmax=lmax)
                e1 = relrms(mg, mo)
                mi = rng.standard_normal(npix)
                ao = so.map2alm(mi, lmax=lmax); ag = shts.map2alm(mi, lmax=lmax)
                print('FULL nside %d lmax %d spin 0: alm2map %.3e map2alm %.3e (oracle synth %.2fs)' % (nside, lmax, e1, relrms(ag, ao), t_or), flush=True)
            else:
                g, c = ralm(rng, lmax, spin), ralm(rng, lmax, spin)
                mo = so.alm2map_spin([g, c], nside, spin, lmax)
                mg = shts.alm2map_spin([g, c], nside, spin, lmax)
                e1 = relrms(np.stack(mg), np.stack(mo))
                mi = rng.standard_normal((2, npix))
                ao = so.map2alm_spin(mi, spin, lmax); ag = shts.map2alm_spin(mi, spin, lmax)
                print('FULL nside %d lmax %d spin %d: alm2map_spin %.3e map2alm_spin %.3e' % (nside, lmax, spin, e1, relrms(np.stack(ag), np.stack(ao))), flush=True)
        except Exception:
            traceback.print_exc()


def timing(nside, lmax):
    plan = shts.get_plan(nside, lmax)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(1)
    for spin in (0, 2):
        ncomp = 1 if spin == 0 else 2
        alm = torch.from_numpy(np.stack([ralm(rng, lmax, spin) for _ in range(ncomp)])).cuda()
        mp = torch.zeros((ncomp, 12 * nside ** 2), dtype=torch.float64, device='cuda')
        ph = torch.zeros(plan.phase_doubles(spin), dtype=torch.float64, device='cuda')
        a2 = torch.zeros_like(alm)
        def run(fn, n=3):
            fn(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        t_ls = run(lambda: _lib.check(L.pl_legendre_synth(plan.h, spin, alm.data_ptr(), None, ph.data_ptr(), st)))
        t_pm = run(lambda: _lib.check(L.pl_phase2map(plan.h, spin, ph.data_ptr(), mp.data_ptr(), st)))
        t_mp = run(lambda: _lib.check(L.pl_map2phase(plan.h, spin, mp.data_ptr(), ph.data_ptr(), st)))
        t_la = run(lambda: _lib.check(L.pl_legendre_anal(plan.h, spin, ph.data_ptr(), a2.data_ptr(), None, st)))
        steps = (lmax + 1) * (lmax + 2) / 2 * 2 * nside
        fl = (8 if spin == 0 else 24) * steps
        print('TIMING nside %d lmax %d spin %d: leg_synth %.3f ms (%.1f TF/s alg) phase2map %.3f ms map2phase %.3f ms leg_anal %.3f ms (%.1f TF/s alg)'
              % (nside, lmax, spin, t_ls, fl / t_ls / 1e9, t_pm, t_mp, t_la, fl / t_la / 1e9), flush=True)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    print('devices', _lib.device_count(), torch.cuda.get_device_name(0), flush=True)
    if which in ('all', 'stage'):
        stage_tests(8, 16)
        stage_tests(16, 47)
        stage_tests(64, 128)
    if which in ('all', 'full'):
        full_tests(8, 16)
        full_tests(32, 95)
        full_tests(64, 150)
        full_tests(256, 512, spins=(0, 2))
        full_tests(1024, 1024, spins=(0, 1, 3))
    if which in ('all', 'timing'):
        print('fp64 fma peak TF/s', L.pl_fma64_peak_tflops(20000, None), L.pl_fma64_peak_tflops(100000, None), flush=True)
        timing(512, 512)
        timing(2048, 2048)
