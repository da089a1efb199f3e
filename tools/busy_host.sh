#!/bin/bash
# How much does the reconstruction rate depend on the host's launch thread?  The bench's QE leg on a quiet host and beside NSPIN busy-loop
# processes (default 32: twice the CPU quota of a GPU slot), with the replayed-graph route (default) and with eager launches
# (PLENS_OPTIONS=qe_graph=0).  Run on the GPU box; prints one line per arm.
cd "$GRAFT_REPO_ROOT"
NSPIN=${1:-32}
STEPS=${2:-20}
ARGS="--no-cg --no-cpu-baseline --no-from-sims --steps $STEPS --warmup 4 ${@:3}"
one() {  # label, env assignment
  env $2 python3 bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
e=d.get('eager_pass') or {}
print('%-34s %8.2f rec/s %8.3f ms/step   graph_replay=%s  eager pass of the same run: %s ms/step  selfcheck %s' % (sys.argv[1], d['value'], d['ms_per_step'], d.get('graph_replay'), ('%.3f' % e['ms_per_step']) if e else '-', d.get('selfcheck_max_abs_diff')))" "$1"
}
spin_start() { PIDS=""; for i in $(seq $NSPIN); do python3 -c "
while True: pass" & PIDS="$PIDS $!"; done; sleep 1; }
spin_stop() { for p in $PIDS; do kill $p 2>/dev/null; done; wait 2>/dev/null; }
for rep in 1 2; do
  one "quiet host, graph replay" "PLENS_OPTIONS=qe_graph=1"
  one "quiet host, eager launches" "PLENS_OPTIONS=qe_graph=0"
  spin_start
  one "$NSPIN spinners, graph replay" "PLENS_OPTIONS=qe_graph=1"
  one "$NSPIN spinners, eager launches" "PLENS_OPTIONS=qe_graph=0"
  spin_stop
done
