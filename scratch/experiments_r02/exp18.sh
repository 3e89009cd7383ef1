#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp18.raw | grep -vE "amdgpu.ids" | grep -E "phase ticks|iter 1|cores/s|passed|failed|step |  F |  G |  eig" | cut -c1-330; if grep -q "GPU core dump" gpurun_out/exp18.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp18.raw
run timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or headline or apply_compress or bench_batch or randomized" || exit 1
grep -q "failed" gpurun_out/exp18.raw && exit 1
for i in 1 2 3; do run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 || exit 1; TTN_FAST=513 run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 || exit 1; done
TTN_PROF=1 run timeout -k 10 120 python tools/diag_batch.py 1 || exit 1
TTN_FAST=513 TTN_PROF=1 run timeout -k 10 120 python tools/diag_batch.py 1 || exit 1
TTN_PROF=1 TTN_WG512=1 run timeout -k 10 120 python tools/diag_batch.py 512 || exit 1
TTN_FAST=513 TTN_PROF=1 TTN_WG512=1 run timeout -k 10 120 python tools/diag_batch.py 512 || exit 1
run timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py -x -q || exit 1
