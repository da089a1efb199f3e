cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_round4.py tests/test_gpu_cg.py tests/test_gpu_cgbatch.py tests/test_gpu_cgvec.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -5
for v in new off new off; do
if [ $v = off ]; then export PLENS_CG_POST_DOTS=0; else unset PLENS_CG_POST_DOTS; fi
echo "--- $v"; CG_BENCH_REPS=1 python3 tools/cg_bench.py 2048 2048 40 2>&1 | tail -1 | cut -c100-260
done
