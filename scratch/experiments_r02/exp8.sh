#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -5
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
for st in 4 5 33 34 52 53; do echo "== step $st"; TTN_PROF=1 TTN_PROF_STEP=$st TTN_WG512=1 python tools/diag_batch.py 512 2>&1 | grep -E "phase ticks"; done
TTN_PROF=1 TTN_WG512=1 python tools/diag_batch.py 512 2>&1 | grep -E "iter 1|per step"
TTN_PROF=1 python tools/diag_batch.py 1 2>&1 | grep -E "iter 1|per step"
python bench.py --no-cpu --steps 4 2>/dev/null | tail -1
TTN_FAST=33 python bench.py --no-cpu --no-single --no-verify --steps 4 2>/dev/null | tail -1 | cut -c1-200
