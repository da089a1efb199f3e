"""Worker of tests/test_gpu_shard.py::test_two_ranks_share_one_transform: one rank of a transform sharded by m-group / ring pair
(parallel.sharded_sht).  Launched twice with RANK / WORLD_SIZE in the environment (gloo rendezvous, both ranks on GPU 0);
writes its results to <out>/rank<r>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    out, nside, lmax = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import torch
    from plancklens_amd import parallel
    from plancklens_amd.helpers import mpi
    mpi.init()
    rng = np.random.default_rng(42)  # the same inputs on every rank
    n = (lmax + 1) * (lmax + 2) // 2

    def ralm(lmin):
        a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        a[:lmax + 1] = a[:lmax + 1].real
        ls = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)])
        a[ls < lmin] = 0.
        return a
    t, e, b = ralm(0), ralm(2), ralm(2)
    maps = rng.standard_normal((3, 12 * nside ** 2))
    sh = parallel.sharded_sht(nside, lmax)
    res = {'rank': mpi.rank, 'size': mpi.size}
    res['tmap'] = sh.alm2map(torch.from_numpy(t).cuda(), 0).cpu().numpy()
    res['qumap'] = sh.alm2map(torch.from_numpy(np.stack([e, b])).cuda(), 2).cpu().numpy()
    own = sh.alm2map(torch.from_numpy(t).cuda(), 0, gather=False)
    res['own_ok'] = bool((own[~sh.own_pixels()] == 0).all()) and bool((own[sh.own_pixels()] == torch.from_numpy(res['tmap']).cuda()[sh.own_pixels()]).all())
    res['tlm'] = sh.map2alm(torch.from_numpy(maps[0]).cuda(), 0).cpu().numpy()
    res['eblm'] = sh.map2alm(torch.from_numpy(maps[1:]).cuda(), 2).cpu().numpy()
    np.savez(os.path.join(out, 'rank%d.npz' % mpi.rank), **res)
    mpi.barrier()
    mpi.finalize()


if __name__ == '__main__':
    main()
