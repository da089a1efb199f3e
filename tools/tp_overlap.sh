#!/bin/bash
# development aid: do the kernels of cinv_t and cinv_p overlap when the two solves run on two streams of one process
# (filt_cinv.apply_ivf_tp)?  Kernel trace of tools/cg_bench.py with few iterations; the last solve pair is the overlapped one.
# usage (on the GPU box): bash tools/tp_overlap.sh [iters] [tag]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
IT=${1:-12}
TAG=${2:-tp_overlap}
rm -rf gpurun_out/$TAG
CG_BENCH_REPS=1 CG_BENCH_BATCHES=${BATCHES-} rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG -o t -- python3 tools/cg_bench.py 2048 2048 $IT > gpurun_out/$TAG.log 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
kt = glob.glob('gpurun_out/%s/**/*kernel_trace.csv' % tag, recursive=True)[0]
rd = list(csv.DictReader(open(kt)))
qk = 'Queue_Id' if 'Queue_Id' in rd[0] else None
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60], r.get(qk, '?') if qk else '?') for r in rd)
log = open('gpurun_out/%s.log' % tag).read()
conc = None
for line in log.splitlines():
    if line.startswith('tp_concurrent'):
        conc = json.loads(line[len('tp_concurrent'):])
    if line.startswith('t {'):
        tt = json.loads(line[2:])
    if line.startswith('p {'):
        pp = json.loads(line[2:])
out = open('gpurun_out/%s_summary.txt' % tag, 'w')
def p(*a):
    s = ' '.join(str(x) for x in a); print(s); out.write(s + '\n')
p('cg_bench: T %.3f s, P %.3f s, overlapped %.3f s' % (tt['seconds'], pp['seconds'], conc['seconds']))
tend = rows[-1][1]
win = [r for r in rows if r[0] >= tend - int(conc['seconds'] * 1e9)]
def union(rs):
    b, ce = 0, None
    for s, e, _, _ in sorted(rs):
        if ce is None or s > ce:
            b += e - s; ce = e
        elif e > ce:
            b += e - ce; ce = e
    return b
span = win[-1][1] - win[0][0]
p('window %.1f ms, %d kernels, sum of durations %.1f ms, union busy %.1f ms' % (span / 1e6, len(win), sum(e - s for s, e, _, _ in win) / 1e6, union(win) / 1e6))
byq = collections.defaultdict(list)
for r in win:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    names = collections.Counter(n for _, _, n, _ in rs).most_common(3)
    p('queue %s: %6d kernels, busy %.1f ms; top: %s' % (q, len(rs), union(rs) / 1e6, names))
# classify by solve: T kernels carry spin-0 names / P kernels spin-s names; the time both kinds are in flight
isT = lambda n: ('synth0' in n or 'anal0' in n or 'post0' in n or 'prep0' in n)
isP = lambda n: ('synths' in n or 'anals' in n or 'posts' in n or 'preps' in n)
T = [r for r in win if isT(r[2])]
P = [r for r in win if isP(r[2])]
ev = sorted([(s, 1, 'T') for s, e, _, _ in T] + [(e, -1, 'T') for s, e, _, _ in T] + [(s, 1, 'P') for s, e, _, _ in P] + [(e, -1, 'P') for s, e, _, _ in P])
c = {'T': 0, 'P': 0}; last = ev[0][0]; both = 0
for t, d, k in ev:
    if c['T'] > 0 and c['P'] > 0:
        both += t - last
    c[k] += d; last = t
# the same pair of solves one after the other (the warm-up call of apply_ivf_tp just before): per-kernel mean durations, alone vs overlapped
t1 = win[0][0]
seq = [r for r in rows if t1 - int((tt['seconds'] + pp['seconds']) * 1.02e9) <= r[0] < t1]
def means(rs):
    d = collections.defaultdict(lambda: [0, 0])
    for s_, e_, n_, q_ in rs:
        d[n_][0] += e_ - s_; d[n_][1] += 1
    return d
ms, mo = means(seq), means(win)
p('sequential window: %d kernels, sum of durations %.1f ms, union %.1f ms' % (len(seq), sum(e - s for s, e, _, _ in seq) / 1e6, union(seq) / 1e6))
p('%-62s %8s %10s %10s %7s' % ('kernel', 'calls', 'alone us', 'overl. us', 'ratio'))
for n_, (t_, c_) in sorted(mo.items(), key=lambda kv: -kv[1][0])[:40]:
    if n_ in ms and ms[n_][1]:
        a = ms[n_][0] / ms[n_][1] / 1e3; o = t_ / c_ / 1e3
        p('%-62s %8d %10.1f %10.1f %7.2f' % (n_, c_, a, o, o / a))
p('Legendre-type kernels: T busy %.1f ms, P busy %.1f ms, both in flight %.1f ms' % (union(T) / 1e6, union(P) / 1e6, both / 1e6))
PY
rm -rf gpurun_out/$TAG
