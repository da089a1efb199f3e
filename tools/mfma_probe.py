"""FP64 pipe probe: sustained TFLOP/s of v_fma_f64 by operand mix (modes 0-2), of v_mfma_f64_16x16x4 alone (3) and of both
interleaved (4, combined flops)."""
import sys
sys.path.insert(0, '.')
from plancklens_amd import _lib
L = _lib.lib()
for mode, name in ((1, 'v_fma_f64, two scalar sources'), (0, 'v_fma_f64, one vector + one scalar'), (2, 'v_fma_f64, three vector sources'),
                   (3, 'v_mfma_f64_16x16x4 only'), (4, 'v_mfma_f64_16x16x4 + v_fma_f64 interleaved (combined)')):
    r = [L.pl_fma64_rate_tflops(mode, 4000, None) for _ in range(3)]
    print('mode %d %-55s %6.1f TFLOP/s' % (mode, name, max(r)))
