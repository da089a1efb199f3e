#!/bin/bash
# development aid: kernel timeline of the headline bench -- where is the GPU idle inside a reconstruction? (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/trace_gaps
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_gaps -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-cg > gpurun_out/trace_gaps.log 2>&1
python3 - <<'PY'
import csv, glob
kt = glob.glob('gpurun_out/trace_gaps/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:70]) for r in csv.DictReader(open(kt))]
mc = glob.glob('gpurun_out/trace_gaps/**/*memory_copy_trace.csv', recursive=True)
copies = []
if mc:
    for r in csv.DictReader(open(mc[0])):
        copies.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '') + ' ' + r.get('Size', '')))
rows.sort()
# last 4 reconstructions: find starts of k_leg_anal0 (one per reconstruction: the temperature filter)
starts = [s for s, e, n in rows if 'k_leg_anal0' in n]
t0 = starts[-4] - 2000000
rows = [r for r in rows if r[0] >= t0]
copies = [c for c in copies if c[0] >= t0]
ev = sorted(rows)
busy, cur_end, gaps = 0, ev[0][0], []
prev = None
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n, cur_end))
        busy += e - s
        cur_end = e
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    prev = n
span = cur_end - ev[0][0]
out = open('gpurun_out/trace_gaps_summary.txt', 'w')
def p(*a):
    s = ' '.join(str(x) for x in a)
    print(s); out.write(s + '\n')
p('span %.2f ms over 4 reconstructions + tail, busy (union of kernels) %.2f ms, idle %.2f ms' % (span / 1e6, busy / 1e6, (span - busy) / 1e6))
p('largest gaps (us): after -> before')
for g, a, b, t in sorted(gaps, reverse=True)[:40]:
    inflight = [c[2] for c in copies if c[0] < t + g and c[1] > t]
    p('%8.1f  %-60s -> %-60s %s' % (g / 1e3, a, b, inflight[:2]))
p('gap histogram: >1ms %d, 100us-1ms %d, 20-100us %d, <20us %d (sum %.2f ms)' % (
    sum(g[0] > 1e6 for g in gaps), sum(1e5 < g[0] <= 1e6 for g in gaps), sum(2e4 < g[0] <= 1e5 for g in gaps), sum(g[0] <= 2e4 for g in gaps),
    sum(g[0] for g in gaps if g[0] <= 2e4) / 1e6))
import collections
tot = collections.defaultdict(lambda: [0, 0])
tend = [s_ for s_, e_, n_ in ev if 'k_fma_peak' in n_]
tend = tend[0] if tend else ev[-1][1]
for s_, e_, n_ in ev:
    if s_ < tend:
        tot[n_][0] += e_ - s_; tot[n_][1] += 1
p('per-kernel totals inside the 4 timed reconstructions (ms per reconstruction, launches per reconstruction):')
acc = 0.
for n_, (t_, c_) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:32]:
    acc += t_ / 4e6
    p('%8.3f %6.1f  %s' % (t_ / 4e6, c_ / 4., n_))
p('sum of all kernel durations per reconstruction: %.2f ms' % (sum(t_ for t_, c_ in tot.values()) / 4e6))
for c in copies[-12:]:
    p('copy %.3f ms %s' % ((c[1] - c[0]) / 1e6, c[2]))
PY
tail -2 gpurun_out/trace_gaps.log | cut -c1-300
rm -rf gpurun_out/trace_gaps
