#!/bin/bash
# development aid: kernel timeline of the bench for one estimator key (where do the milliseconds outside the Legendre / FFT kernels go?)
# usage (GPU box): bash tools/trace_key.sh ptt|p_p|p [steps]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
KEY=${1:-ptt}
STEPS=${2:-6}
D=gpurun_out/trace_$KEY
rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 bench.py --key $KEY --steps $STEPS --warmup 4 --no-cpu-baseline --no-cg > gpurun_out/trace_$KEY.log 2>&1
python3 - "$KEY" "$STEPS" <<'PY'
import csv, glob, sys, collections, json
key, steps = sys.argv[1], int(sys.argv[2])
D = 'gpurun_out/trace_%s' % key
kt = glob.glob(D + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r["Kernel_Name"][:110]) for r in csv.DictReader(open(kt)))
copies = []
for f in glob.glob(D + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        copies.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
line = [l for l in open('gpurun_out/trace_%s.log' % key) if l.startswith('{')][0]
ms_step = json.loads(line)['ms_per_step']
# the timed region: the last `steps` reconstructions before the FMA-rate probe kernels of the bench
tend = [s for s, e, n in rows if 'k_fma_peak' in n]
tend = tend[0] if tend else rows[-1][1]
anals = [s for s, e, n in rows if s < tend and ('k_leg_anals' in n or (key == 'ptt' and 'k_leg_anals' in n))]
per = {'ptt': 1, 'p_p': 1, 'p': 1}[key]
t0 = tend - int(ms_step * 1e6 * steps)
win = [r for r in rows if t0 <= r[0] < tend]
out = open('gpurun_out/trace_%s_summary.txt' % key, 'w')
def p(*a):
    s = ' '.join(str(x) for x in a); print(s); out.write(s + '\n')
busy, ce, gaps, prev = 0, win[0][0], [], None
for s, e, n in win:
    if s > ce:
        gaps.append((s - ce, prev, n)); busy += e - s; ce = e
    elif e > ce:
        busy += e - ce; ce = e
    prev = n
span = ce - win[0][0]
p("key %s: bench line %.2f ms per reconstruction; window %.2f ms for %d reconstructions; union of kernels %.2f ms, idle %.2f ms per reconstruction"
  % (key, ms_step, span / 1e6, steps, busy / 1e6 / steps, (span - busy) / 1e6 / steps))
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    tot[n][0] += e - s; tot[n][1] += 1
p('per-kernel totals (ms per reconstruction, launches per reconstruction); sum of durations %.2f ms per reconstruction' % (sum(t for t, c in tot.values()) / 1e6 / steps))
for n, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:28]:
    p('%8.3f %6.1f  %s' % (t / 1e6 / steps, c / steps, n))
p('largest gaps (us): after -> before')
for g, a, b in sorted(gaps, reverse=True)[:14]:
    p('%8.1f  %-50s -> %s' % (g / 1e3, a, b))
cw = [c for c in copies if t0 <= c[0] < tend]
p('memory copies in the window: %d, %.2f ms per reconstruction' % (len(cw), sum(e - s for s, e, _ in cw) / 1e6 / steps))
PY
rm -rf $D
