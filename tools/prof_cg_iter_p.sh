#!/bin/bash
# kernel statistics of ONE top-level polarization CG iteration: difference of two kernel-trace runs of tools/cg_bench.py
# (CG_BENCH_ONLY=p) with 6 and 18 iterations per solve (2 solves each).  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export CG_BENCH_ONLY=p
for n in 6 18; do
    rm -rf gpurun_out/prof_cgp$n
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cgp$n -o cg -- python3 tools/cg_bench.py 2048 2048 $n > gpurun_out/prof_cgp$n.log 2>&1
    rm -f gpurun_out/prof_cgp$n/cg_kernel_trace.csv
done
python3 - <<'PY'
import csv
def load(n):
    return {r['Name']: (int(r['Calls']), int(r['TotalDurationNs'])) for r in csv.DictReader(open('gpurun_out/prof_cgp%d/cg_kernel_stats.csv' % n))}
a, b = load(6), load(18)
nit = 2 * (18 - 6)
rows = []
for k in b:
    c0, t0 = a.get(k, (0, 0))
    c1, t1 = b[k]
    if c1 - c0 > 0:
        rows.append((k, (c1 - c0) / nit, (t1 - t0) / nit / 1e3))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print('kernel time per top-level iteration: %.2f ms in %.0f launches' % (tot / 1e3, sum(r[1] for r in rows)))
for k, c, t in rows[:36]:
    print('%8.1f calls %9.1f us  %s' % (c, t, k[:95]))
PY
