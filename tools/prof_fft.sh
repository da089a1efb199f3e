cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fft -o fft -- python3 tools/kernel_bench.py 2048 2048 5 ps,pa 0 > gpurun_out/prof_fft.log 2>&1
find gpurun_out/prof_fft -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof_fft -name "*kernel_stats.csv" | head -1)
head -30 "$f" | cut -c1-200
