#!/bin/bash
# isolated durations of the ring-FFT class kernels (classes serialised) under a set of development switches given as VAR=VALUE arguments
# usage (GPU box): bash tools/fft_class_times.sh [nside] [lmax] [spin] [VAR=VALUE ...]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
NS=${1:-2048}; LM=${2:-2048}; SP=${3:-2}; shift 3
rm -rf gpurun_out/fct
env PLSHTS_DEBUG=1 PLSHTS_FFT_SERIAL=1 "$@" true
export PLSHTS_DEBUG=1 PLSHTS_FFT_SERIAL=1 "$@"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fct -o t -- python3 tools/kernel_bench.py $NS $LM 5 ps,pa $SP > gpurun_out/fct.log 2>&1
python3 - "$@" <<'PY'
import csv, glob, sys, collections
kt = glob.glob('gpurun_out/fct/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    n = r['Kernel_Name']
    if 'phase2map' in n or 'map2phase' in n:
        d[n.split('(')[0].replace('void plshts::', '')[:60]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print(' '.join(sys.argv[1:]) or '(defaults)')
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:10]:
    v = sorted(v)
    print('  %8.1f us (median of %2d)  %s' % (v[len(v) // 2] / 1e3, len(v), n))
PY
rm -rf gpurun_out/fct
