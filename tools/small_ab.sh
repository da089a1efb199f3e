# BASELINE config 1 ('ptt', nside = lmax = 512) with and without the plan's seed tables, alternating
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --no-cg --no-cpu-baseline --no-from-sims "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('%-90s %8.2f rec/s %8.3f ms  eager %.3f' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d.get('eager_pass',{}).get('ms_per_step',0)))" "$@"; }
for i in 1 2; do
run --key ptt --nside 512 --lmax 512 --steps 100 --warmup 20
run --key ptt --nside 512 --lmax 512 --steps 100 --warmup 20 --plan-opt seed_tables=0
done
run --key ptt --nside 1024 --lmax 1024 --steps 40 --warmup 10
run --key ptt --nside 1024 --lmax 1024 --steps 40 --warmup 10 --plan-opt seed_tables=0
