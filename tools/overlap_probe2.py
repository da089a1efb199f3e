"""Do the ring-FFT kernels of one transform run under the Legendre kernel of another?  Stream A: spin-2 Legendre syntheses back
to back; stream B (a fork of the plan): spin-2 ring-FFT synthesis stages back to back.  Alone, and together."""
import ctypes, sys, time
import torch
sys.path.insert(0, '.')
from plancklens_amd import shts, _lib
nside = lmax = 2048
L = _lib.lib()
plan = shts.get_plan(nside, lmax)
fork = plan.fork(1)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
n = plan.nalm
alm = torch.randn((2, n), dtype=torch.complex128, device='cuda')
ph = torch.zeros(plan.phase_doubles(2), dtype=torch.float64, device='cuda')
ph2 = torch.randn(plan.phase_doubles(2), dtype=torch.float64, device='cuda')
mp = torch.empty((2, 12 * nside ** 2), dtype=torch.float64, device='cuda')
NA, NB = 6, 24
def leg(k):
    for _ in range(k):
        _lib.check(L.pl_legendre_synth(plan.h, 2, alm.data_ptr(), None, ph.data_ptr(), ctypes.c_void_p(sa.cuda_stream)))
def fft(k):
    for _ in range(k):
        _lib.check(L.pl_phase2map(fork.h, 2, ph2.data_ptr(), mp.data_ptr(), ctypes.c_void_p(sb.cuda_stream)))
leg(1); fft(1); torch.cuda.synchronize()
def timed(fa, fb):
    torch.cuda.synchronize(); t0 = time.time(); fa(); fb(); torch.cuda.synchronize(); return 1e3 * (time.time() - t0)
ta = timed(lambda: leg(NA), lambda: None)
tb = timed(lambda: None, lambda: fft(NB))
tab = timed(lambda: leg(NA), lambda: fft(NB))
print('%d Legendre syntheses alone %.2f ms (%.2f each); %d FFT stages alone %.2f ms (%.2f each); together %.2f ms (sum %.2f, max %.2f)'
      % (NA, ta, ta / NA, NB, tb, tb / NB, tab, ta + tb, max(ta, tb)))
