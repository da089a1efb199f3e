#!/bin/bash
# kernel statistics of ONE top-level CG iteration of a block solve of B right-hand sides: difference of two kernel-trace runs with
# 4 and 12 iterations per solve (4 solves each: one warm-up + 3).   usage: tools/prof_cg_batch.sh B t|p     (run on the GPU box)
B=${1:-4}; K=${2:-t}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for n in 4 12; do
    rm -rf gpurun_out/prof_cgb$n
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cgb$n -o cg -- python3 tools/cg_profile_b.py $n $B $K > gpurun_out/prof_cgb$n.log 2>&1
    rm -f gpurun_out/prof_cgb$n/cg_kernel_trace.csv
done
python3 - "$B" "$K" <<'PY'
import csv, sys
B, K = sys.argv[1], sys.argv[2]
def load(n):
    import glob
    fn = glob.glob('gpurun_out/prof_cgb%d/**/cg_kernel_stats.csv' % n, recursive=True)[0]
    return {r['Name']: (int(r['Calls']), int(r['TotalDurationNs'])) for r in csv.DictReader(open(fn))}
a, b = load(4), load(12)
nit = 4 * (12 - 4)  # cg_profile_b.py runs 4 solves per trace (1 warm-up + 3 timed)
rows = []
for k in b:
    c0, t0 = a.get(k, (0, 0))
    c1, t1 = b[k]
    if c1 - c0 > 0:
        rows.append((k, (c1 - c0) / nit, (t1 - t0) / nit / 1e3))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
out = 'gpurun_out/cg_iter_kernels_%s_B%s.csv' % (K, B)
with open(out, 'w') as f:
    f.write('kernel,calls_per_iteration,us_per_iteration\n')
    for k, c, t in rows:
        f.write('"%s",%.1f,%.1f\n' % (k[:100], c, t))
print('B = %s %s: kernel time per top-level iteration: %.2f ms in %.0f launches' % (B, K, tot / 1e3, sum(r[1] for r in rows)))
for k, c, t in rows[:30]:
    print('%8.1f calls %9.1f us  %s' % (c, t, k[:95]))
PY
grep -h "un-instrumented" gpurun_out/prof_cgb4.log gpurun_out/prof_cgb12.log
