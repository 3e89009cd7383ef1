#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp13.raw | grep -E "phase ticks|iter 1|cores/s|passed|failed|per step|clk|build" | cut -c1-1800; if grep -q "GPU core dump" gpurun_out/exp13.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp13.raw
run timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q || exit 1
grep -q "failed" gpurun_out/exp13.raw && exit 1
TTN_DIAG_SYRK_ONLY=1 run python tools/diag_gemm.py || exit 1
TTN_DIAG_SYRK_ONLY=1 TTN_WG512_SELFTEST=1 run python tools/diag_gemm.py || exit 1
TTN_DIAG_SYRK_ONLY=1 TTN_WG512_SELFTEST=1 TTN_BENCH_GRID=512 run python tools/diag_gemm.py || exit 1
run timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py -x -q || exit 1
grep -q "failed" gpurun_out/exp13.raw && exit 1
for i in 1 2; do run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 4 || exit 1; done
TTN_PROF=1 TTN_WG512=1 run timeout -k 10 120 python tools/diag_batch.py 512 || exit 1
TTN_PROF=1 run timeout -k 10 120 python tools/diag_batch.py 1 || exit 1
