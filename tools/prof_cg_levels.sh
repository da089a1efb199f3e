#!/bin/bash
# kernel time of ONE top-level CG iteration by kernel AND grid size (= multigrid level): difference of two kernel traces with 4 and 12
# iterations per solve.   usage: tools/prof_cg_levels.sh t|p     (run on the GPU box; writes gpurun_out/cg_iter_levels_<k>.csv)
K=${1:-t}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for n in 4 12; do
    rm -rf gpurun_out/prof_cgl$n
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_cgl$n -o cg -- python3 tools/cg_profile_b.py $n 1 $K > gpurun_out/prof_cgl$n.log 2>&1
done
python3 - "$K" <<'PY'
import csv, glob, sys, collections
K = sys.argv[1]
def load(n):
    fn = glob.glob('gpurun_out/prof_cgl%d/**/cg_kernel_trace.csv' % n, recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(fn)):
        key = (r['Kernel_Name'][:70], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Y', ''))
        acc[key][0] += 1
        acc[key][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return acc
a, b = load(4), load(12)
nit = 4 * (12 - 4)
rows = []
for k in b:
    c0, t0 = a.get(k, (0, 0))
    c1, t1 = b[k]
    if c1 - c0 > 0:
        rows.append((k, (c1 - c0) / nit, (t1 - t0) / nit / 1e3))
rows.sort(key=lambda r: -r[2])
with open('gpurun_out/cg_iter_levels_%s.csv' % K, 'w') as f:
    f.write('kernel,grid_x,grid_y,calls_per_iteration,us_per_iteration,us_per_call\n')
    for (k, gx, gy), c, t in rows:
        f.write('"%s",%s,%s,%.1f,%.1f,%.1f\n' % (k, gx, gy, c, t, t / c))
print('total %.2f ms in %.0f launches' % (sum(r[2] for r in rows) / 1e3, sum(r[1] for r in rows)))
PY
rm -rf gpurun_out/prof_cgl4 gpurun_out/prof_cgl12
