#!/bin/bash
# per-kernel latency on the coarse CG grids by rings per lane; run on the GPU box
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for cfg in "128 256" "256 512" "512 1024"; do
    set -- $cfg
    rm -rf gpurun_out/coarse_$1
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/coarse_$1 -o c -- python3 tools/coarse_probe.py $1 $2 > gpurun_out/coarse_$1.log 2>&1
    rm -f gpurun_out/coarse_$1/c_kernel_trace.csv
    echo "== nside $1 lmax $2"
    python3 - $1 <<'PY'
import csv, sys
rows = list(csv.DictReader(open('gpurun_out/coarse_%s/c_kernel_stats.csv' % sys.argv[1])))
for r in sorted(rows, key=lambda r: r['Name']):
    if 'k_leg' in r['Name'] or 'k_p' in r['Name'] or 'map' in r['Name']:
        print('%-70s %5s calls %8.1f us avg %8.1f min' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
