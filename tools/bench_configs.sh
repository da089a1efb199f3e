#!/bin/bash
# the other BASELINE / SURVEY 8(d) configurations through the same bench driver (run on the GPU box)
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --no-cg --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-60s %8.2f rec/s %8.2f ms' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step']))" "$@"; }
run --key ptt --nside 512 --lmax 512 --steps 100 --warmup 20
run --key ptt --steps 10 --warmup 2
run --key p_p --steps 10 --warmup 2
#run --key p --steps 10 --warmup 2
run --key p --steps 10 --warmup 2 --lmax-qlm 4096
run --key p --steps 4 --warmup 1 --qe-only
run --key p_p --steps 4 --warmup 1 --qe-only
run --key ptt --steps 4 --warmup 1 --qe-only
run --key p --nside 4096 --lmax 4096 --steps 4 --warmup 2
