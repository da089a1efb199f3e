#!/bin/bash
# kernel statistics (call counts, durations) of the temperature CG (tools/cg_profile.py); run on the GPU box
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_cg
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cg -o cg -- python3 tools/cg_profile.py 6 > gpurun_out/prof_cg.log 2>&1
rm -f gpurun_out/prof_cg/cg_kernel_trace.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_cg/cg_kernel_stats.csv')))
rows.sort(key=lambda r: -int(r['Calls']))
for r in rows[:32]:
    print('%8d calls  %9.1f us avg  %8.2f ms total  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, int(r['TotalDurationNs']) / 1e6, r['Name'][:90]))
PY
