#!/bin/bash
# kernel timeline of the replayed pair graph at BASELINE config 1 ('ptt', nside = lmax = 512): per-kernel totals and the busy / idle split
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
KEY=${1:-ptt}; NS=${2:-512}
rm -rf gpurun_out/trace_small
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_small -o t -- python3 bench.py --no-cg --no-cpu-baseline --no-from-sims --key $KEY --nside $NS --lmax $NS --steps 40 --warmup 8 > gpurun_out/trace_small.log 2>&1
python3 - <<'PY'
import csv, glob, collections
fn = glob.glob('gpurun_out/trace_small/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:70]) for r in csv.DictReader(open(fn))]
rows.sort()
# last 30 % of the trace = steady replay (+ the eager pass): take the window of the 2000 kernels before the last 40 %
n = len(rows)
win = rows[int(0.35 * n):int(0.55 * n)]
t0, t1 = win[0][0], max(r[1] for r in win)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('window: %d kernels, %.3f ms wall, union of kernels %.3f ms (%.0f %%)' % (len(win), (t1 - t0) / 1e6, busy / 1e6, 100. * busy / (t1 - t0)))
acc = collections.defaultdict(lambda: [0, 0])
for s, e, k in win:
    acc[k][0] += e - s; acc[k][1] += 1
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:25]:
    print('%9.3f ms %6d x %7.2f us  %s' % (t / 1e6, c, t / c / 1e3, k))
PY
rm -rf gpurun_out/trace_small
