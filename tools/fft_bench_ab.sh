#!/bin/bash
# the headline bench (QE leg only) under the ring-FFT development switches, alternating on one box: wave kernels on / off, cheapest-first
# launch order on the side streams on / off.  Prints step time and the per-stage FFT figures of the eager pass.  usage (GPU box): bash tools/fft_bench_ab.sh
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for cfg in "1 1" "0 0" "1 0" "0 1"; do
  set -- $cfg
  PLSHTS_DEBUG=1 PLSHTS_FFT_WAVE=$1 PLSHTS_FFT_SMALL_FIRST=$2 python3 bench.py --no-cg --no-cpu-baseline --no-from-sims --no-plan-stats --steps 20 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['kernels']
print('WAVE=$1 SMALL_FIRST=$2: %.3f ms/step (eager pass %.3f)  fft_synth %.3f ms/launch (%.3f per component)  fft_anal %.3f ms/launch (%.3f per component)' % (d['ms_per_step'], d['eager_pass']['ms_per_step'], k['fft_synth']['avg_ms'], k['fft_synth']['ms_per_component'], k['fft_anal']['avg_ms'], k['fft_anal']['ms_per_component']))"
done; done
