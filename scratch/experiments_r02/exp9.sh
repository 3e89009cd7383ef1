#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -5
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
TTN_DIAG_SYRK_ONLY=1 python tools/diag_gemm.py 2>&1 | grep -v LDS
TTN_DIAG_SYRK_ONLY=1 TTN_WG512_SELFTEST=1 python tools/diag_gemm.py 2>&1 | grep -v LDS
TTN_DIAG_SYRK_ONLY=1 TTN_WG512_SELFTEST=1 TTN_BENCH_GRID=512 python tools/diag_gemm.py 2>&1 | grep -v LDS
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py -x -q 2>&1 | tail -5
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
for st in 5 10 34 40; do echo "== step $st"; TTN_PROF=1 TTN_PROF_STEP=$st TTN_WG512=1 python tools/diag_batch.py 512 2>&1 | grep -E "phase ticks"; done
TTN_PROF=1 TTN_WG512=1 python tools/diag_batch.py 512 2>&1 | grep -E "iter 1|per step|phase ticks"
TTN_PROF=1 python tools/diag_batch.py 1 2>&1 | grep -E "iter 1|per step"
python bench.py --no-cpu --steps 4 2>/dev/null | tail -1
TTN_FAST=33 python bench.py --no-cpu --no-single --no-verify --steps 4 2>/dev/null | tail -1 | cut -c1-200
