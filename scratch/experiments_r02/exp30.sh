#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp30.raw | grep -vE "amdgpu.ids" | grep -E "phase ticks|iter 1|cores/s|passed|failed|  F|Error|error" | cut -c1-200; if grep -q "GPU core dump" gpurun_out/exp30.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp30.raw
run timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py -x -q -k "headline or bench_batch or randomized" || exit 1
grep -q "failed" gpurun_out/exp30.raw && exit 1
for i in 1 2 3; do run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 || exit 1; TTN_FAST=4097 run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 || exit 1; done
run timeout -k 10 100 python tools/diag_batch.py 1
TTN_FAST=4097 run timeout -k 10 100 python tools/diag_batch.py 1
TTN_PROF_STEP=40 TTN_WG512=1 run python tools/diag_fine.py 512
run timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py tests/test_gpu_kernels.py -x -q || exit 1
