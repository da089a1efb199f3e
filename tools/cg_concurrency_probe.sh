#!/bin/bash
# would the T and the P solve of BASELINE config 4 overlap on one GPU?  Two processes, one solve each, alone and together.
cd "$GRAFT_REPO_ROOT"
one() { CG_BENCH_ONLY=$1 python3 tools/cg_bench.py 2048 2048 100 2>/dev/null | grep "^$1 " | python3 -c "
import sys, json
for l in sys.stdin:
    k, js = l.split(' ', 1); d = json.loads(js); print(k, 'seconds %.3f it/s %.2f' % (d['seconds'], d['iters_per_s']))"; }
echo "== alone"; one t; one p
echo "== together (timed sections start together)"; export CG_BENCH_BARRIER_DIR=$(mktemp -d); one t & one p & wait
