# rings per lane of the Legendre kernel families re-measured with the seed tables (the recursion-only phase no longer costs anything, so the
# coarser polar pruning of a large ring group is what a larger R pays): Legendre stage times at nside = lmax = 2048
cd $GRAFT_REPO_ROOT
export PLSHTS_DEBUG=1
for v in "PLSHTS_R0=2" "PLSHTS_R0=3" "PLSHTS_R0=4" "PLSHTS_R0A=4" "PLSHTS_R0A=5" "PLSHTS_R0A=6" "PLSHTS_R0A=8" "PLSHTS_RS=1" "PLSHTS_RS=2" "PLSHTS_RS=3" "PLSHTS_RSA=2" "PLSHTS_RSA=3" "PLSHTS_RSA=4"; do
  case $v in PLSHTS_R0=*) a="ls 0";; PLSHTS_R0A=*) a="la 0";; PLSHTS_RS=*) a="ls 2";; PLSHTS_RSA=*) a="la 2";; esac
  set -- $a
  echo "== $v: $(env $v python3 tools/kernel_bench.py ${NSIDE:-2048} ${NSIDE:-2048} ${REPS:-5} $1 $2 2>&1 | grep -i " $1\b\|^$1\|ms" | tail -2 | tr '\n' ' ')"
done
