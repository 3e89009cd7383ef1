#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp28.raw | grep -vE "amdgpu.ids" | grep -E "phase ticks|iter 1|cores/s|passed|failed|per step|  G|Error|error" | cut -c1-1300; if grep -q "GPU core dump" gpurun_out/exp28.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp28.raw
run timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py -x -q -k "headline or bench_batch" || exit 1
grep -q "failed" gpurun_out/exp28.raw && exit 1
for i in 1 2 3; do run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 || exit 1; TTN_FAST=2049 run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 || exit 1; done
TTN_PROF=1 TTN_WG512=1 run timeout -k 10 120 python tools/diag_batch.py 512 || exit 1
TTN_PROF_STEP=10 TTN_WG512=1 run python tools/diag_fine.py 512
run timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py tests/test_gpu_kernels.py -x -q || exit 1
