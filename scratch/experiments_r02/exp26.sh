#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp26.raw | grep -vE "amdgpu.ids" | grep -E "phase ticks|iter 1|cores/s|passed|failed|per step|Error|error" | cut -c1-1500; if grep -q "GPU core dump" gpurun_out/exp26.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp26.raw
run timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "headline or bench_batch" || exit 1
grep -q "failed" gpurun_out/exp26.raw && exit 1
TTN_PROF=1 run timeout -k 10 120 python tools/diag_batch.py 1 || exit 1
for i in 1 2 3; do run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 || exit 1; TTN_FAST=1025 run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 || exit 1; done
TTN_FAST=1025 TTN_PROF=1 run timeout -k 10 120 python tools/diag_batch.py 1 || exit 1
run timeout -k 10 1100 python -m pytest tests -x -q -m gpu || exit 1
