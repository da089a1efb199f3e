#!/bin/bash
# LDS bank-conflict counters of the FFT stage kernels (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_lds
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_lds -o l -- python3 tools/kernel_bench.py 2048 2048 1 ps,pa,la 0 > gpurun_out/pmc_lds.log 2>&1
python3 - <<'PY'
import csv, collections, glob
fn = glob.glob('gpurun_out/pmc_lds/*counter_collection.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fn[0])):
    acc[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:16]:
    a, c = v.get('SQ_LDS_IDX_ACTIVE', 0), v.get('SQ_LDS_BANK_CONFLICT', 0)
    print('%-62s active %12.0f conflict %12.0f  (%.1f %%)' % (k, a, c, 100 * c / max(a, 1)))
PY
rm -rf gpurun_out/pmc_lds
