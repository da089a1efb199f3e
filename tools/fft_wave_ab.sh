#!/bin/bash
# A/B of the wavefront-private ring-FFT kernels of the direct N = 2048 rings (k_phase2map_wave / k_map2phase_wave, PLSHTS_FFT_WAVE=1, the
# default) against the one-group / quad kernels they replace (PLSHTS_FFT_WAVE=0): per-kernel durations with the classes serialised, and the
# stage times of tools/kernel_bench.py with the classes on their side streams.  usage (GPU box): bash tools/fft_wave_ab.sh [nside] [lmax] [spin]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
NS=${1:-2048}; LM=${2:-2048}; SP=${3:-2}
for W in 0 1; do
  rm -rf gpurun_out/fwab_$W
  PLSHTS_DEBUG=1 PLSHTS_FFT_WAVE=$W PLSHTS_FFT_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fwab_$W -o t -- python3 tools/kernel_bench.py $NS $LM 5 ps,pa $SP > gpurun_out/fwab_$W.log 2>&1
  python3 - $W <<'PY'
import csv, glob, sys, collections
q = sys.argv[1]
kt = glob.glob('gpurun_out/fwab_%s/**/*kernel_trace.csv' % q, recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    n = r['Kernel_Name']
    if 'phase2map' in n or 'map2phase' in n:
        d[n.split('(')[0].replace('void plshts::', '')[:60]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('WAVE=%s (classes serialised, isolated kernel durations)' % q)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]:
    v = sorted(v)
    print('  %8.1f us (median of %2d)  %s' % (v[len(v) // 2] / 1e3, len(v), n))
PY
  rm -rf gpurun_out/fwab_$W
  echo "WAVE=$W stage times (classes on side streams):"
  for rep in 1 2; do PLSHTS_DEBUG=1 PLSHTS_FFT_WAVE=$W python3 tools/kernel_bench.py $NS $LM 20 ps,pa 0,$SP 2>/dev/null | grep -i "ps\|pa"; done
done
