#!/bin/bash
# rocprofv3 kernel-trace summary of the headline bench (run on the GPU box; output under gpurun_out/)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
head -40 "$f" | cut -c1-180
tail -2 gpurun_out/prof_bench.log | cut -c1-400
