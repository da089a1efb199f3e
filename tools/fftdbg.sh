for d in 0 1 2 4 6; do echo "== FFTDBG=$d"; PLSHTS_FFTDBG=$d python3 tools/kernel_bench.py 2048 2048 3 ps,pa 0,2 2>&1 | grep -v amdgpu.ids; done
