#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp31.raw | grep -vE "amdgpu.ids" | grep -E "phase ticks|iter 1|cores/s|passed|failed|  eig|Error|error|^v1|^v2" | cut -c1-300; if grep -q "GPU core dump" gpurun_out/exp31.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp31.raw
run timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_wg512.py -x -q -k "eig" || exit 1
grep -q "failed" gpurun_out/exp31.raw && exit 1
run timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "headline or bench_batch or eigen" || exit 1
grep -q "failed" gpurun_out/exp31.raw && exit 1
run bash scratch/ab2.sh v1 v2
TTN_PROF_STEP=10 run python tools/diag_fine.py 1
TTN_LIB=$PWD/tensortrainnumerics.jl_amd/libttn_v1.so TTN_PROF_STEP=10 run python tools/diag_fine.py 1
