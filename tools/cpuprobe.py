import os, time, sys
sys.path.insert(0, '.')
import numpy as np
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' ; cat /proc/loadavg")
from oracle import sht_oracle as so
so.build()
nside = lmax = 1024
c, s, pair, slots = so._pair_geometry(nside, True)
rng = np.random.default_rng(0)
nalm = so.alm_size(lmax)
alm2 = rng.standard_normal((2, nalm)) + 1j * rng.standard_normal((2, nalm))
sel = np.arange(0, 2 * nside, 4)
for nt in (8, 32, 64, 128, 256):
    t0 = time.perf_counter(); so.legendre(0, 1, 2, lmax, lmax, c[sel], s[sel], pair[sel], alm=alm2, nthreads=nt); t1 = time.perf_counter()
    print('legendre spin 2 nside 1024 every 4th ring, %3d threads: %.3f s' % (nt, t1 - t0), flush=True)
