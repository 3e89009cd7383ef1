#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp33.raw | grep -vE "amdgpu.ids" | grep -E "iter 1|cores/s|passed|failed|Error|error|^base|^s8|^s16|per step" | cut -c1-1200; if grep -q "GPU core dump" gpurun_out/exp33.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp33.raw
run timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or headline or bench_batch or randomized" || exit 1
grep -q "failed" gpurun_out/exp33.raw && exit 1
TTN_LIB=$PWD/tensortrainnumerics.jl_amd/libttn_s16.so run timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or headline or bench_batch or randomized" || exit 1
grep -q "failed" gpurun_out/exp33.raw && exit 1
run bash scratch/ab2.sh base s8 s16
TTN_LIB=$PWD/tensortrainnumerics.jl_amd/libttn_s16.so TTN_PROF=1 run python tools/diag_batch.py 1
TTN_LIB=$PWD/tensortrainnumerics.jl_amd/libttn_s8.so TTN_PROF=1 run python tools/diag_batch.py 1
