"""One transform sharded over several ranks (BASELINE.json north_star: "m-blocks shard across the GPUs"):
Legendre stage by interleaved m-group, ring FFTs by interleaved ring pair, phase slices exchanged in between
(pl_plan_create_shard, pl_phase_pack / pl_phase_unpack, pl_alm_keep_mgroups; plancklens_amd.parallel.sharded_sht).
Must equal the single-plan transform (tolerance 1e-13 relative rms; the only difference is the grouping of the ring partial
sums of the analysis when the shard plan picks another number of rings per lane)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import random_alm, relrms

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-13


@pytest.mark.parametrize('nside,lmax,R', [(16, 40, 2), (64, 128, 3), (512, 512, 2), (512, 700, 4)])
def test_shards_in_one_process_equal_the_whole(nside, lmax, R):
    """R shard plans side by side on one GPU, the all-to-all done by hand with the pack / unpack kernels: every m-group and every
    ring pair is handled by exactly one shard, and the assembled transforms equal the ordinary ones (spins 0 and 2)."""
    import torch
    from plancklens_amd import _lib, dev, hp, shts
    L = _lib.lib()
    rng = np.random.default_rng(7 + R)
    st = dev.stream_ptr()
    plans = [shts.get_shard_plan(nside, lmax, r, R) for r in range(R)]
    npix = 12 * nside ** 2
    pair = np.minimum(hp.pix2ring(nside), 4 * nside - hp.pix2ring(nside)) - 1
    for spin in (0, 2):
        nc = 1 if spin == 0 else 2
        alm = np.stack([random_alm(rng, lmax, lmin=spin) for _ in range(nc)])
        a = torch.from_numpy(alm).cuda()
        ref = shts.alm2map(a[0], nside, lmax=lmax) if spin == 0 else torch.stack(shts.alm2map_spin([a[0], a[1]], nside, spin, lmax))
        phases = [torch.full((p.phase_doubles(spin),), float('nan'), dtype=torch.float64, device='cuda') for p in plans]
        for r, p in enumerate(plans):
            _lib.check(L.pl_legendre_synth(p.h, spin, a.data_ptr(), None, phases[r].data_ptr(), st))
        for r in range(R):       # sender
            for s in range(R):   # receiver: the ring pairs of s, the m-groups of r
                n = int(L.pl_phase_pack_doubles(plans[r].h, nc, s, R, r, R))
                buf = torch.empty(max(n, 1), dtype=torch.float64, device='cuda')
                _lib.check(L.pl_phase_pack(plans[r].h, nc, phases[r].data_ptr(), buf.data_ptr(), s, R, r, R, st))
                if s != r:
                    _lib.check(L.pl_phase_unpack(plans[s].h, nc, phases[s].data_ptr(), buf.data_ptr(), s, R, r, R, st))
        out = torch.full((nc, npix), float('nan'), dtype=torch.float64, device='cuda')
        for r, p in enumerate(plans):
            m = torch.zeros((nc, npix), dtype=torch.float64, device='cuda')
            _lib.check(L.pl_phase2map(p.h, spin, phases[r].data_ptr(), m.data_ptr(), st))
            own = torch.from_numpy(pair % R == r).cuda()
            assert bool((m[:, ~own] == 0).all()), 'a shard wrote pixels of rings it does not own'
            out[:, own] = m[:, own]
        assert relrms(dev.to_host(out), dev.to_host(ref).reshape(nc, npix)) < TOL, (spin, 'synthesis')
        # analysis
        mp = torch.from_numpy(rng.standard_normal((nc, npix))).cuda()
        refa = shts.map2alm(mp[0], lmax=lmax) if spin == 0 else torch.stack(shts.map2alm_spin([mp[0], mp[1]], spin, lmax))
        phases = [torch.full((p.phase_doubles(spin),), float('nan'), dtype=torch.float64, device='cuda') for p in plans]
        for r, p in enumerate(plans):
            _lib.check(L.pl_map2phase(p.h, spin, mp.data_ptr(), phases[r].data_ptr(), st))
        for r in range(R):       # sender: its ring pairs, the m-groups of the receiver
            for s in range(R):
                if s == r:
                    continue
                n = int(L.pl_phase_pack_doubles(plans[r].h, nc, r, R, s, R))
                buf = torch.empty(max(n, 1), dtype=torch.float64, device='cuda')
                _lib.check(L.pl_phase_pack(plans[r].h, nc, phases[r].data_ptr(), buf.data_ptr(), r, R, s, R, st))
                _lib.check(L.pl_phase_unpack(plans[s].h, nc, phases[s].data_ptr(), buf.data_ptr(), r, R, s, R, st))
        tot = torch.zeros((nc, hp.Alm.getsize(lmax)), dtype=torch.complex128, device='cuda')
        for r, p in enumerate(plans):
            alm_r = torch.empty((nc, hp.Alm.getsize(lmax)), dtype=torch.complex128, device='cuda')
            _lib.check(L.pl_legendre_anal(p.h, spin, phases[r].data_ptr(), alm_r.data_ptr(), None, st))
            _lib.check(L.pl_alm_keep_mgroups(lmax, nc, alm_r.data_ptr(), r, R, st))
            assert bool(torch.isfinite(torch.view_as_real(alm_r)).all())
            assert bool(((tot != 0) & (alm_r != 0)).sum() == 0), 'two shards produced the same alm entry'
            tot += alm_r
        assert relrms(dev.to_host(tot), dev.to_host(refa).reshape(nc, -1)) < TOL, (spin, 'analysis')


def test_two_ranks_share_one_transform(tmp_path):
    """parallel.sharded_sht under torch.distributed: two processes (gloo rendezvous, both on the box's one GPU, the all-to-all
    staged through the host as RCCL would do over xGMI with one GPU per rank) transform the same inputs at nside = lmax = 512;
    both get the single-process results."""
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    nside = lmax = 512
    base = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        base.pop(k, None)
    worker = os.path.join(ROOT, 'tests', 'workers', 'shard_worker.py')
    procs = [subprocess.Popen([sys.executable, worker, str(tmp_path), str(nside), str(lmax)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(base, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                                       PLENS_DIST_BACKEND='gloo')) for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    import torch
    from plancklens_amd import dev, shts
    rng = np.random.default_rng(42)   # the worker's inputs
    t, e, b = random_alm(rng, lmax, 0), random_alm(rng, lmax, 2), random_alm(rng, lmax, 2)
    maps = rng.standard_normal((3, 12 * nside ** 2))
    ref = {'tmap': shts.alm2map(t, nside, lmax=lmax), 'qumap': np.stack(shts.alm2map_spin([e, b], nside, 2, lmax)),
           'tlm': shts.map2alm(maps[0], lmax=lmax), 'eblm': np.stack(shts.map2alm_spin([maps[1], maps[2]], 2, lmax))}
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        assert int(got['rank']) == r and int(got['size']) == 2 and bool(got['own_ok'])
        for k, v in ref.items():
            assert relrms(got[k], v) < TOL, (r, k)


@pytest.mark.parametrize('nside,R', [(1, 2), (4, 3), (16, 2), (64, 5), (512, 8)])
def test_ring_pack_covers_the_map_once(nside, R):
    """pl_map_pack_rings / pl_map_unpack_rings (the all-gather that completes a sharded synthesis): the packed buffers of the R ranks
    have the announced sizes, together hold every pixel exactly once, and unpacking them rebuilds the map bit for bit (two components)."""
    import torch
    from plancklens_amd import _lib, dev, hp, shts
    L = _lib.lib()
    st = dev.stream_ptr()
    lmax = 2 * nside
    plan = shts.get_plan(nside, lmax)
    npix = 12 * nside ** 2
    rng = np.random.default_rng(nside + R)
    ref = torch.from_numpy(rng.standard_normal((2, npix))).cuda()
    out = torch.full((2, npix), float('nan'), dtype=torch.float64, device='cuda')
    pair = np.minimum(hp.pix2ring(nside), 4 * nside - hp.pix2ring(nside)) - 1
    total = 0
    for r in range(R):
        n = int(L.pl_map_pack_doubles(plan.h, r, R))
        assert n == int((pair % R == r).sum()), (r, n)
        total += n
        buf = torch.full((2, max(n, 1)), float('nan'), dtype=torch.float64, device='cuda')
        _lib.check(L.pl_map_pack_rings(plan.h, 2, ref.data_ptr(), buf.data_ptr(), r, R, st))
        assert n == 0 or bool(torch.isfinite(buf[:, :n]).all())
        # the packed values are this rank's pixels in ring order
        own = torch.from_numpy(pair % R == r).cuda()
        assert sorted(dev.to_host(buf[0, :n]).tolist()) == sorted(dev.to_host(ref[0][own]).tolist())
        _lib.check(L.pl_map_unpack_rings(plan.h, 2, out.data_ptr(), buf.data_ptr(), r, R, st))
    assert total == npix
    assert bool((out == ref).all())
