# the headline QE leg (replayed pair graph) under runtime queue settings: how much do the parallel branches of a ring-FFT stage buy, and
# does the spread of graph branches over hardware queues matter?
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo "=== $cfg"
  env $(echo $cfg | tr ',' ' ') timeout 400 python bench.py --no-cg --no-cpu-baseline --no-from-sims --no-plan-stats 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    k=d['kernels']
    print('value %.2f rec/s  %.3f ms/step  eager %.3f  fft_synth %.3f fft_anal %.3f ms/comp' % (d['value'], d['ms_per_step'], d['eager_pass']['ms_per_step'], k['fft_synth']['ms_per_component'], k['fft_anal']['ms_per_component']))
except Exception as e:
    print('failed', e)
"
done
