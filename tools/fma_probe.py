import sys; sys.path.insert(0,'.')
from plancklens_amd import _lib
import torch
torch.zeros(1, device='cuda')
L=_lib.lib(); L.pl_fma64_peak_tflops.restype=__import__('ctypes').c_double
print(L.pl_fma64_peak_tflops(20000, None))
