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
  timeout ${TEST_TIMEOUT:-3000} python3 -m pytest ${PYTEST_PATHS:-tests} -m gpu -x -q ${PYTEST_ARGS:-} > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/$TAG/pytest_gpu.log
fi
if [[ $WHAT == *s0pair* ]]; then
  for sz in "2048 2048" "1024 1024" "4096 4096"; do timeout 300 python3 tools/s0_pair_ab.py $sz 10; done > gpurun_out/$TAG/s0_pair_ab.txt 2>&1; cat gpurun_out/$TAG/s0_pair_ab.txt
fi
if [[ $WHAT == *config5* ]]; then
  timeout 900 python3 bench.py --config 5 --steps 8 > gpurun_out/$TAG/bench_config5.json 2> gpurun_out/$TAG/bench_config5.err; echo "config5 rc $?"; head -c 700 gpurun_out/$TAG/bench_config5.json; echo
fi
if [[ $WHAT == *probe2* ]]; then
  for o in 210 102 012; do timeout 300 tools/probes/seam_probe 189 $o; done > gpurun_out/$TAG/seam_probe_orders.txt 2>&1; cat gpurun_out/$TAG/seam_probe_orders.txt
fi
if [[ $WHAT == *benchslots* ]]; then
  # the headline with its inputs copied into static slots (the route of rounds 5 / 6a-d) against the address-table route, same box, QE leg only
  for rep in 1 2; do for o in 1 0; do
    PLENS_OPTIONS=qe_indirect=$o timeout 600 python3 bench.py --no-cg --no-cpu-baseline --no-from-sims --no-plan-stats --steps 20 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('qe_indirect=$o: %.3f ms/step = %.2f rec/s (eager pass %.3f ms), resident map sets %d, selfcheck %s' % (d['ms_per_step'], d['value'], d['eager_pass']['ms_per_step'], d['config']['resident_map_sets'], d['selfcheck_max_abs_diff']))"
  done; done > gpurun_out/$TAG/bench_indirect_ab.txt 2>&1; cat gpurun_out/$TAG/bench_indirect_ab.txt
fi
if [[ $WHAT == *configs* ]]; then
  timeout 1200 bash tools/bench_configs.sh > gpurun_out/$TAG/bench_configs.log 2>&1; cat gpurun_out/$TAG/bench_configs.log
fi
if [[ $WHAT == *twostreams* ]]; then
  timeout 900 python3 tools/qe_two_streams.py 2048 5 > gpurun_out/$TAG/qe_two_streams.txt 2>&1; tail -5 gpurun_out/$TAG/qe_two_streams.txt
fi
