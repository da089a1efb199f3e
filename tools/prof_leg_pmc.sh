#!/bin/bash
# SQ counters of the Legendre kernels at nside = lmax = 2048, with the plan's seed tables and without them (plan option seed_tables = 0):
# instructions issued, wave cycles and where they wait -- run on the GPU box
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for opt in "seed_tables=1" "seed_tables=0"; do
echo "=== plan option $opt"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
rm -rf gpurun_out/pmc_leg
rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_leg -o v -- python3 tools/kernel_bench.py 2048 2048 2 ls,la 0,2 $opt > gpurun_out/pmc_leg.log 2>&1
python3 - <<'PY'
import csv, collections, glob
fn = glob.glob('gpurun_out/pmc_leg/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(fn[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void plshts::', '')[:36]
    if 'k_leg' in k:
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
for k, v in sorted(acc.items()):
    print('%-36s' % k, ' '.join('%s=%.4g' % (a.replace('SQ_', ''), b / n[(k, a)]) for a, b in v.items()))
PY
done
done
rm -rf gpurun_out/pmc_leg
