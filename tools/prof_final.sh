#!/bin/bash
# round-end evidence: kernel-trace stats of the headline bench + PMC traffic of the SHT stages (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_bench gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
rm -f gpurun_out/prof_bench/bench_kernel_trace.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 tools/kernel_bench.py 2048 2048 2 ls,la,ps,pa 0,2 > gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 tools/kernel_bench.py 2048 2048 2 ls,la,ps,pa 0,2 > gpurun_out/pmc_w.log 2>&1
python3 - <<'PY'
import csv, collections, glob
for tag, col in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE')):
    fn = glob.glob('gpurun_out/pmc_%s/*counter_collection.csv' % tag)
    if not fn:
        print('no counter file for', tag); continue
    acc = collections.defaultdict(lambda: [0., 0])
    for r in csv.DictReader(open(fn[0])):
        if r.get('Counter_Name') == col:
            k = r['Kernel_Name'].split('(')[0][:60]
            acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    with open('gpurun_out/pmc_%s_summary.csv' % tag, 'w') as f:
        f.write('kernel,%s_KB_mean_per_launch,launches\n' % col)
        for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            f.write('%s,%.0f,%d\n' % (k.replace(',', ';'), v / n, n))
    print(open('gpurun_out/pmc_%s_summary.csv' % tag).read()[:1500])
PY
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
head -30 gpurun_out/prof_bench/bench_kernel_stats.csv | cut -c1-150
