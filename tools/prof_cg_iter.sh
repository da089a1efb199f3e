#!/bin/bash
# kernel statistics of ONE top-level temperature CG iteration: difference of two kernel-trace runs with 4 and 12
# iterations per solve (3 solves each), so that the dense-preconditioner build and set-up drop out.  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for n in 4 12; do
    rm -rf gpurun_out/prof_cg$n
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cg$n -o cg -- python3 tools/cg_profile.py $n > gpurun_out/prof_cg$n.log 2>&1
    rm -f gpurun_out/prof_cg$n/cg_kernel_trace.csv
done
python3 - <<'PY'
import csv
def load(n):
    return {r['Name']: (int(r['Calls']), int(r['TotalDurationNs'])) for r in csv.DictReader(open('gpurun_out/prof_cg%d/cg_kernel_stats.csv' % n))}
a, b = load(4), load(12)
nit = 3 * (12 - 4)
rows = []
for k in b:
    c0, t0 = a.get(k, (0, 0))
    c1, t1 = b[k]
    if c1 - c0 > 0:
        rows.append((k, (c1 - c0) / nit, (t1 - t0) / nit / 1e3))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
with open('gpurun_out/cg_iter_kernels.csv', 'w') as f:
    f.write('kernel,calls_per_iteration,us_per_iteration\n')
    for k, c, t in rows:
        f.write('"%s",%.1f,%.1f\n' % (k[:100], c, t))
print('kernel time per top-level iteration: %.2f ms in %.0f launches' % (tot / 1e3, sum(r[1] for r in rows)))
for k, c, t in rows[:40]:
    print('%8.1f calls %9.1f us  %s' % (c, t, k[:95]))
PY
grep -h "un-instrumented" gpurun_out/prof_cg4.log gpurun_out/prof_cg12.log
