#!/bin/bash
# development aid: time the Legendre stages for the ring-count-per-lane choices (and alternative builds via PLSHTS_LIB)
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python3 tools/kernel_bench.py 2048 2048 3 ${ST:-ls,la} ${SP:-0,2} 2>&1 | grep -v amdgpu.ids; }
ST=ls,la SP=0,2 run X=1
for r in 1 2 4; do ST=la SP=2 run PLSHTS_RSA=$r; done
for r in 2 3 5 6; do ST=la SP=0 run PLSHTS_R0A=$r; done
for v in $VARIANTS; do ST=ls,la SP=0,2 run PLSHTS_LIB=$PWD/plancklens_amd/csrc/$v.so; done
