#!/bin/bash
# rocprofv3 kernel trace of the CG benchmark (T and P solves, 6 iterations each)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_cg
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cg -o cg -- python3 tools/cg_bench.py 2048 2048 6 > gpurun_out/prof_cg.log 2>&1
tail -3 gpurun_out/prof_cg.log | cut -c1-300
ls -la gpurun_out/prof_cg | head
