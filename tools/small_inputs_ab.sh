#!/bin/bash
# BASELINE config 1 ('ptt', nside = lmax = 512) through bench.py's QE leg: the replayed pair graph (graph_min_nside=0) against eager launches (1024), one or two resident map sets,
# inputs through the address table or copied into slots.  First figure: the timed region (results retained, as a mean-field run keeps them); in brackets the eager pass after it.
cd "$GRAFT_REPO_ROOT"
run() { env "PLENS_OPTIONS=$1" python3 bench.py --no-cg --no-cpu-baseline --no-from-sims --no-plan-stats --key ptt --nside 512 --lmax 512 --steps 100 --warmup 20 --resident-sets $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
e=d.get('eager_pass') or {}
print('%-44s sets=$2: %8.2f rec/s %8.3f ms (graph replay %s; eager pass %s ms)' % ('$1', d['value'], d['ms_per_step'], d['graph_replay'], e.get('ms_per_step')))"; }
for rep in 1 2; do
run qe_graph_min_nside=0,qe_indirect=1 2
run qe_graph_min_nside=0,qe_indirect=1 1
run qe_graph_min_nside=0,qe_indirect=0 2
run qe_graph_min_nside=1024 2
run qe_graph_min_nside=1024 1
done
