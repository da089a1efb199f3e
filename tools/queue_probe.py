"""Which streams of this process share a hardware queue?  (dev.streams_overlap: two spin kernels started together take the time of one or of two.)
Prints, for torch's default stream, a few streams of torch's pool and the side streams of an nside-2048 plan, the classes of streams whose kernels
do NOT overlap one another.    python tools/queue_probe.py [nside]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plancklens_amd import _lib, dev, shts

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
plan = shts.Plan(nside, nside)
L = _lib.lib()
names, streams = ['default'], [torch.cuda.default_stream()]
for i in range(6):
    names.append('torch%d' % i)
    streams.append(torch.cuda.Stream())
h = plan.h
for i in range(5):
    ptr = L.pl_plan_side_stream(h, i)
    if ptr:
        names.append('plan.s%d' % i)
        streams.append(torch.cuda.ExternalStream(ptr))
classes = []
for n, s in zip(names, streams):
    for c in classes:
        if not dev.streams_overlap(c[0][1], s):
            c.append((n, s))
            break
    else:
        classes.append([(n, s)])
for c in classes:
    print('queue class:', ' '.join(n for n, _ in c))
