#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp23.raw | grep -vE "amdgpu.ids" | grep -E "phase ticks|iter 1|cores/s|passed|failed|step |  F |  G |  eig|  merge|clk|build" | cut -c1-200; if grep -q "GPU core dump" gpurun_out/exp23.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp23.raw
run timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q || exit 1
grep -q "failed" gpurun_out/exp23.raw && exit 1
TTN_DIAG_SYRK_ONLY=1 TTN_WG512_SELFTEST=1 run python tools/diag_gemm.py || exit 1
TTN_DIAG_SYRK_ONLY=1 TTN_WG512_SELFTEST=1 TTN_BENCH_GRID=512 run python tools/diag_gemm.py || exit 1
run timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused or headline or apply_compress or bench_batch or randomized" || exit 1
grep -q "failed" gpurun_out/exp23.raw && exit 1
L=$PWD/tensortrainnumerics.jl_amd
for i in 1 2 3; do run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 || exit 1; TTN_LIB=$L/libttn_new.so run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 || exit 1; done
TTN_PROF_STEP=10 TTN_WG512=1 run python tools/diag_fine.py 512
