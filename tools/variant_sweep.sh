#!/bin/bash
# development aid: time SHT stages with alternative builds of the library (PLSHTS_LIB), e.g. ablation builds
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python3 tools/kernel_bench.py 2048 2048 3 ${ST:-ps,pa} ${SP:-0} 2>&1 | grep -v amdgpu.ids; }
run X=1
for v in $VARIANTS; do run PLSHTS_LIB=$PWD/plancklens_amd/csrc/$v.so; done
