# BASELINE config 1 ('ptt', nside = lmax = 512): replayed pair graph against eager launches under runtime settings of the graph launch path
cd "$GRAFT_REPO_ROOT"
run() { env $1 python3 bench.py --no-cg --no-cpu-baseline --no-from-sims --key ptt --nside 512 --lmax 512 --steps 100 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-60s %8.2f rec/s %8.3f ms  eager %.3f' % (sys.argv[1], d['value'], d['ms_per_step'], d.get('eager_pass',{}).get('ms_per_step',0)))" "$1"; }
for e in "$@"; do run $e; done
