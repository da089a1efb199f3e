#!/bin/bash
# development aid: per-kernel durations of the ring-FFT stage (one kernel at a time: PLSHTS_FFT_SERIAL=1) for the one-group and the
# quad kernels.  usage (GPU box): bash tools/fft_kernel_stats.sh [nside] [lmax] [spin]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
NS=${1:-2048}; LM=${2:-2048}; SP=${3:-2}
for Q in 0 1; do
  rm -rf gpurun_out/fks_$Q
  PLSHTS_DEBUG=1 PLSHTS_FFT_QUAD=$Q PLSHTS_FFT_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fks_$Q -o t -- python3 tools/kernel_bench.py $NS $LM 5 ps,pa $SP > gpurun_out/fks_$Q.log 2>&1
  python3 - $Q <<'PY'
import csv, glob, sys, collections
q = sys.argv[1]
kt = glob.glob('gpurun_out/fks_%s/**/*kernel_trace.csv' % q, recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    n = r['Kernel_Name']
    if 'phase2map' in n or 'map2phase' in n:
        d[n.split('(')[0].replace('void plshts::', '')[:60] + ' grid=' + r.get('Grid_Size_X', r.get('Grid_Size', '?'))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('QUAD=%s' % q)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print('  %8.1f us (median of %2d)  %s' % (v[len(v) // 2] / 1e3, len(v), n))
PY
  rm -rf gpurun_out/fks_$Q
done
