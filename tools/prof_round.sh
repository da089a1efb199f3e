#!/bin/bash
# round evidence set (run on the GPU box): the bench line, rocprofv3 kernel stats of the same command, PMC traffic of the SHT stages
TAG=${1:-round3_a}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
rm -rf gpurun_out/prof_bench gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cg > gpurun_out/$TAG/prof_bench.log 2>&1
cp $(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/bench_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 tools/kernel_bench.py 2048 2048 2 ls,la,ps,pa 0,2 > gpurun_out/$TAG/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 tools/kernel_bench.py 2048 2048 2 ls,la,ps,pa 0,2 > gpurun_out/$TAG/pmc_w.log 2>&1
TAG=$TAG python3 - <<'PY'
import csv, collections, glob, os
tag = os.environ['TAG']
res = collections.defaultdict(dict)
for t, col in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE')):
    fn = glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % t, recursive=True)
    if not fn:
        continue
    acc = collections.defaultdict(lambda: [0., 0])
    for r in csv.DictReader(open(fn[0])):
        if r.get('Counter_Name') == col:
            k = r['Kernel_Name'].split('(')[0][:60].replace(',', ';')
            acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for k, (v, n) in acc.items():
        res[k][col] = v / n; res[k]['n'] = n
with open('gpurun_out/%s/pmc_traffic.csv' % tag, 'w') as f:
    f.write('# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/kernel_bench.py 2048 2048 2 ls,la,ps,pa 0,2  (tools/prof_round.sh)\n')
    f.write('# MI355X, nside = lmax = 2048; rocprofv3 FETCH_SIZE / WRITE_SIZE in KB, mean per launch.  gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 1/2 of the bytes of\n')
    f.write('# 16-B-per-lane streaming reads; other widths (scalar table streams, 8-B pixel stores) are uncalibrated; Infinity-Cache hits included.\n')
    f.write('kernel,FETCH_SIZE_KB,WRITE_SIZE_KB,launches\n')
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get('FETCH_SIZE', 0) + kv[1].get('WRITE_SIZE', 0))):
        f.write('%s,%.0f,%.0f,%d\n' % (k, v.get('FETCH_SIZE', 0), v.get('WRITE_SIZE', 0), v.get('n', 0)))
print(open('gpurun_out/%s/pmc_traffic.csv' % tag).read()[:2500])
PY
rm -rf gpurun_out/prof_bench gpurun_out/pmc_f gpurun_out/pmc_w
head -12 gpurun_out/$TAG/bench_kernel_stats.csv | cut -c1-160
head -c 600 gpurun_out/$TAG/bench.json; echo
