#!/bin/bash
# Two solves at once on one GPU: cinv_t and cinv_p (or two of one kind) in two processes whose timed solves start together
# (CG_BENCH_BARRIER_DIR).  Measures whether the dependent microsecond launches of one solve's coarse levels hide under the
# other's.   usage: tools/cg_concurrent.sh [kindA] [kindB] [iters]
A=${1:-t}; B=${2:-p}; N=${3:-100}
cd "$GRAFT_REPO_ROOT"
d=$(mktemp -d)
CG_BENCH_BATCHES= CG_BENCH_BARRIER_DIR=$d CG_BENCH_ONLY=$A python3 tools/cg_bench.py 2048 2048 $N > gpurun_out/conc_a.log 2>&1 &
CG_BENCH_BATCHES= CG_BENCH_BARRIER_DIR=$d CG_BENCH_ONLY=$B python3 tools/cg_bench.py 2048 2048 $N > gpurun_out/conc_b.log 2>&1 &
wait
grep -h "^[tp] {" gpurun_out/conc_a.log gpurun_out/conc_b.log | cut -c1-120
rm -rf $d
