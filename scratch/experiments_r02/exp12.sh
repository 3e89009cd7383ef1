#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp12.raw | grep -E "phase ticks|iter 1|cores/s|passed|failed|per step" | cut -c1-1800; if grep -q "GPU core dump" gpurun_out/exp12.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp12.raw
for i in 1 2; do
run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 4 || exit 1
TTN_FAST=33 run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 4 || exit 1
done
TTN_PROF=1 TTN_WG512=1 run timeout -k 10 120 python tools/diag_batch.py 512 || exit 1
TTN_PROF=1 run timeout -k 10 120 python tools/diag_batch.py 1 || exit 1
run timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py tests/test_gpu_kernels.py -x -q || exit 1
