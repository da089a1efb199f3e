#!/bin/bash
# VALU utilisation counters of the Legendre kernels (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAVES"; do
rm -rf gpurun_out/pmc_v
rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_v -o v -- python3 tools/kernel_bench.py 2048 2048 1 ls,la 2 > gpurun_out/pmc_v.log 2>&1
python3 - <<'PY'
import csv, collections, glob
fn = glob.glob('gpurun_out/pmc_v/*counter_collection.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fn[0])):
    k = r['Kernel_Name'].split('(')[0][:40]
    if 'k_leg_' in k:
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in acc.items():
    print(k, {a: '%.4g' % b for a, b in v.items()})
PY
done
rm -rf gpurun_out/pmc_v
