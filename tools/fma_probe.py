"""FP64 FMA issue rate by operand kind (pl_fma64_rate_tflops): 0 vector + scalar, 1 two scalar, 2 three vector sources"""
import sys
sys.path.insert(0, '.')
import torch
from plancklens_amd import _lib
torch.zeros(1, device='cuda')
L = _lib.lib()
for m in (1, 0, 2):
    print('mode', m, '%.1f TFLOP/s' % L.pl_fma64_rate_tflops(m, 20000, None))
