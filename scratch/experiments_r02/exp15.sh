#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp15.raw | grep -vE "amdgpu.ids" | cut -c1-300; if grep -q "GPU core dump" gpurun_out/exp15.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp15.raw
TTN_PROF_STEP=40 TTN_WG512=1 run timeout -k 10 200 python tools/diag_fine.py 512 || exit 1
TTN_PROF_STEP=10 TTN_WG512=1 run timeout -k 10 200 python tools/diag_fine.py 512 || exit 1
TTN_PROF_STEP=40 run timeout -k 10 200 python tools/diag_fine.py 1 || exit 1
TTN_PROF_STEP=10 run timeout -k 10 200 python tools/diag_fine.py 1 || exit 1
