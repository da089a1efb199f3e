#!/bin/bash
# development aid: k_leg_anals timing by rings-per-lane and by experiment build (PL_EXP_ANALS bit flags, wrong results)
cd "$(dirname "$0")/.."
kb() { env "$@" python3 tools/kernel_bench.py 2048 2048 5 la 2 2>&1 | grep -v amdgpu.ids | grep " la:"; }
for r in 4 3 2 1; do echo "== default lib RSA=$r"; kb PLSHTS_RSA=$r; done
for v in 1 2 4 7; do
  for r in 4 2; do echo "== exp$v RSA=$r"; kb PLSHTS_LIB=$PWD/plancklens_amd/csrc/libplshts_exp$v.so PLSHTS_RSA=$r; done
done
echo "== spin 0 analysis by R0A"
for r in 6 4 3 2; do echo "== default lib R0A=$r"; env PLSHTS_R0A=$r python3 tools/kernel_bench.py 2048 2048 5 la 0 2>&1 | grep " la:"; done
