#!/bin/bash
# SQ counters and isolated durations of the ring-FFT kernels (classes serialised on one stream) -- run on the GPU box
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export PLSHTS_DEBUG=1 PLSHTS_FFT_SERIAL=1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
rm -rf gpurun_out/pmc_fft
rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_fft -o v -- python3 tools/kernel_bench.py 2048 2048 2 ps,pa 2 > gpurun_out/pmc_fft.log 2>&1
python3 - <<'PY'
import csv, collections, glob
fn = glob.glob('gpurun_out/pmc_fft/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(fn[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void plshts::', '')[:48]
    if 'phase' in k:
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
for k, v in sorted(acc.items()):
    print('%-48s' % k, ' '.join('%s=%.3g' % (a.replace('SQ_', ''), b / n[(k, a)]) for a, b in v.items()))
PY
done
rm -rf gpurun_out/pmc_fft
unset PLSHTS_FFT_SERIAL
PLSHTS_DEBUG=1 PLSHTS_FFT_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_fft -o v -- python3 tools/kernel_bench.py 2048 2048 3 ps,pa 2 > gpurun_out/pmc_fft.log 2>&1
f=$(find gpurun_out/pmc_fft -name "*kernel_stats.csv" | head -1); echo "== isolated durations (classes serialised), 2 components per launch"; cut -d, -f1-4,6 "$f" | grep phase | head -24
rm -rf gpurun_out/pmc_fft
