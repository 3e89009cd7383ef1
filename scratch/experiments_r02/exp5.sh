#!/bin/bash
python tools/diag_gemm.py 2>&1 | grep -v "LDS n="
TTN_WG512_SELFTEST=1 python tools/diag_gemm.py 2>&1 | grep -v "LDS n="
TTN_WG512_SELFTEST=1 TTN_BENCH_GRID=512 python tools/diag_gemm.py 2>&1 | grep -v "LDS n="
