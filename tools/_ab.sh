cd $GRAFT_REPO_ROOT
for v in new off new off new off; do
if [ $v = off ]; then export PLENS_CG_POST_DOTS=0; else unset PLENS_CG_POST_DOTS; fi
echo "--- $v"; CG_BENCH_REPS=1 python3 tools/cg_bench.py 2048 2048 40 2>&1 | tail -1 | cut -c100-260
done
unset PLENS_CG_POST_DOTS
bash tools/prof_cg_levels.sh t | tail -1
