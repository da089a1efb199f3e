#!/bin/bash
# kernel timeline of the ring-FFT stages (one component and two), from a kernel trace of tools/kernel_bench.py; run on the GPU box
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ffttl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ffttl -o t -- python3 tools/kernel_bench.py ${1:-2048} ${2:-2048} 2 ps,pa 0,2 > gpurun_out/ffttl.log 2>&1
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/ffttl/**/*kernel_trace.csv', recursive=True)[0])))
rows = [r for r in rows if 'phase2map' in r['Kernel_Name'] or 'map2phase' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# group launches into stages: a gap of more than 200 us starts a new stage
stages, cur = [], []
for r in rows:
    if cur and int(r['Start_Timestamp']) - max(int(x['End_Timestamp']) for x in cur) > 100000:
        stages.append(cur); cur = []
    cur.append(r)
stages.append(cur)
for st in stages[-4:]:
    t0 = min(int(r['Start_Timestamp']) for r in st)
    print('stage: %d kernels, %.1f us wall' % (len(st), (max(int(r['End_Timestamp']) for r in st) - t0) / 1e3))
    for r in st:
        print('   %7.1f -> %7.1f us  grid %-12s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
                                                      r.get('Grid_Size', '?') + '/' + r.get('Workgroup_Size', '?'), r['Kernel_Name'][:75]))
PY
