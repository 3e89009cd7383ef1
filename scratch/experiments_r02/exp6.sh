#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_wg512.py -x -q --timeout 200 -k "gemm" 2>&1 | tail -3
python tools/diag_gemm.py 2>&1 | grep -v "LDS n="
TTN_WG512_SELFTEST=1 python tools/diag_gemm.py 2>&1 | grep -v "LDS n="
TTN_WG512_SELFTEST=1 TTN_BENCH_GRID=512 python tools/diag_gemm.py 2>&1 | grep -v "LDS n="
bash scratch/ab.sh base new
