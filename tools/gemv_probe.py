"""k_gemv at the size of the temperature dense preconditioner (4290 x 4290), timed with events; prints us per mat-vec and TB/s"""
import sys
import torch
sys.path.insert(0, '.')
from plancklens_amd import dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4290
A = torch.randn(n, n, dtype=torch.float64, device='cuda')
x = torch.randn(n, dtype=torch.float64, device='cuda')
y = dev.gemv(A, x)
assert torch.allclose(y, A @ x, rtol=1e-12, atol=1e-9)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        dev.gemv(A, x, out=y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print('gemv %d: %.1f us, %.2f TB/s' % (n, us, n * n * 8 / us / 1e6))
