#!/bin/bash
# round-6 GPU session (run on the GPU box): the seam probe, the default bench line, the -m gpu suite
TAG=${1:-round6_a}
WHAT=${2:-probe,bench,tests}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
if [[ $WHAT == *probe* ]]; then
  timeout 300 tools/probes/seam_probe > gpurun_out/$TAG/seam_probe.txt 2>&1; echo "seam probe rc $?"; cat gpurun_out/$TAG/seam_probe.txt
fi
if [[ $WHAT == *bench* ]]; then
  timeout 900 python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc $?"; head -c 1500 gpurun_out/$TAG/bench.json; echo
fi
if [[ $WHAT == *tests* ]]; then
  timeout ${TEST_TIMEOUT:-3000} python3 -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/$TAG/pytest_gpu.log
fi
