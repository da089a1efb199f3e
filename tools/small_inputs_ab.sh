cd "$GRAFT_REPO_ROOT"
run() { env "$1" python3 bench.py --no-cg --no-cpu-baseline --no-from-sims --no-plan-stats --key ptt --nside 512 --lmax 512 --steps 100 --warmup 20 --resident-sets $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$1 sets=$2: %8.2f rec/s %8.3f ms (eager %.3f ms)' % (d['value'], d['ms_per_step'], d['eager_pass']['ms_per_step']))"; }
for rep in 1 2; do
run PLENS_OPTIONS=qe_indirect=1 2
run PLENS_OPTIONS=qe_indirect=1 1
run PLENS_OPTIONS=qe_indirect=0 2
run PLENS_OPTIONS=qe_indirect=0 1
done
