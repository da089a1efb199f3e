#!/bin/bash
# generic ring-FFT kernel with / without the side-by-side sub-DFTs at nside 512 / 1024 / 2048; run on the GPU box
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for kb in 48 160; do
for cfg in "512 1024" "1024 2048" "2048 2048"; do
    set -- $cfg
    rm -rf gpurun_out/b4_$1
    PLSHTS_FFT_B4_KB=$kb rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b4_$1 -o c -- python3 tools/coarse_probe.py $1 $2 1 > gpurun_out/b4_$1.log 2>&1
    echo "== B4 limit $kb KB, nside $1 lmax $2"
    python3 - $1 <<'PY'
import csv, sys
rows = list(csv.DictReader(open('gpurun_out/b4_%s/c_kernel_stats.csv' % sys.argv[1])))
for r in sorted(rows, key=lambda r: r['Name']):
    if 'k_phase2map<' in r['Name'] or 'k_map2phase<' in r['Name']:
        print('%-70s %5s calls %8.1f us avg %8.1f min' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
done
