"""Round 5, second half: stream placement on the hardware queues, executed-step accounting of the seed tables."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_stream_is_one_that_overlaps():
    """dev.concurrent_stream: the side stream of the T || P solves and the result-copy stream are picked by measurement -- two spin kernels
    started together on the caller's stream and on the candidate take the time of one (different hardware queues) or of two (the HIP
    runtime put both streams on one queue: the 'concurrent' solves would run one after the other).  The pick overlaps the caller's
    stream; a stream never overlaps itself; of torch's first eight pool streams at least one shares a queue class with another
    (four hardware queues by default) -- the case the probe exists for -- unless the runtime was told to use more queues."""
    import torch
    from plancklens_amd import dev
    cur = torch.cuda.current_stream()
    assert not dev.streams_overlap(cur, cur)
    s = dev.concurrent_stream()
    assert s.cuda_stream != cur.cuda_stream
    assert dev.streams_overlap(cur, s)
    assert dev.streams_overlap(s, cur)
    side = dev.concurrent_stream(beside=[cur, s])
    assert dev.streams_overlap(cur, side) and dev.streams_overlap(s, side)


def test_executed_steps_of_a_plan_follow_its_seed_tables():
    """pl_plan_executed_steps (what bench.py prices a Legendre launch with): without tables -1; with them fewer steps than the count after
    polar pruning alone, more than half of it, and the same for both directions up to the families' different ring-group sizes."""
    from plancklens_amd import _lib, shts
    L = _lib.lib()
    nside = lmax = 512
    plain = shts.Plan(nside, lmax, opts={'seed_tables': 0})
    seeded = shts.Plan(nside, lmax)
    nalm = (lmax + 1) * (lmax + 2) // 2
    for spin in (0, 2):
        assert L.pl_plan_executed_steps(plain.h, spin, 0) == -1
        full = nalm * 2 * nside / (2 if spin == 0 else 1)  # every (l, m, ring pair): two-l steps for spin 0
        for fam in (0, 1):
            st = int(L.pl_plan_executed_steps(seeded.h, spin, fam))
            assert 0.5 * full < st < 0.95 * full, (spin, fam, st / full)
