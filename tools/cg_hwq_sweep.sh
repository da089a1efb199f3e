# CG iteration rate against the number of hardware queues the HIP runtime spreads its streams over (GPU_MAX_HW_QUEUES, default 4)
cd $GRAFT_REPO_ROOT
for q in 1 2 3 4 6 8; do
  for only in t p; do
    echo "=== GPU_MAX_HW_QUEUES=$q only=$only"
    GPU_MAX_HW_QUEUES=$q CG_BENCH_ONLY=$only CG_BENCH_BATCHES= timeout 300 python tools/cg_bench.py 2048 2048 100 2>&1 | tail -2 | cut -c1-400
  done
done
