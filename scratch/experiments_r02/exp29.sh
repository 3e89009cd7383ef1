#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp29.raw | grep -vE "amdgpu.ids" | grep -E "phase ticks|iter 1|cores/s|passed|failed|  eig|Error|error|^base|^tw" | cut -c1-300; if grep -q "GPU core dump" gpurun_out/exp29.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp29.raw
run timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_wg512.py -x -q -k "eig or headline or bench_batch" || exit 1
grep -q "failed" gpurun_out/exp29.raw && exit 1
run bash scratch/ab2.sh base tw
TTN_PROF_STEP=10 TTN_WG512=1 run python tools/diag_fine.py 512
run timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py tests/test_gpu_kernels.py -x -q || exit 1
