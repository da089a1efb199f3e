# CG rates (T, P one after the other; T || P) under runtime queue settings; B = 4 block solves run first, as in bench.py's flow
cd $GRAFT_REPO_ROOT
run() {
  echo "=== $*"
  env "$@" CG_BENCH_BATCHES=${BATCHES-4} timeout 400 python tools/cg_bench.py 2048 2048 100 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    c=d['TP_concurrent']
    print('T %.1f P %.1f TP %.1f conc %.1f (x%.3f) overlap %s B4conc %s' % (d['T_iters_per_s'], d['P_iters_per_s'], d['TP_iters_per_s'], c['iters_per_s'], c['speedup_vs_one_after_the_other'], c.get('streams_overlap'), (c.get('block_solves') or {}).get('4', {}).get('iters_per_s_per_sim')))
except Exception as e:
    print('failed', e)
"
}
for cfg in "$@"; do run $(echo $cfg | tr ',' ' '); done
