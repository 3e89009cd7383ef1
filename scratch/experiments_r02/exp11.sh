#!/bin/bash
run() { "$@" 2>&1 | tee -a gpurun_out/exp11.raw | grep -vE "amdgpu.ids" | cut -c1-3000; if grep -q "GPU core dump" gpurun_out/exp11.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp11.raw
echo "== cholqr on, wg512"; TTN_WG512=1 run timeout -k 10 200 python tools/diag_determinism.py 512 || exit 1
echo "== cholqr off, wg512"; TTN_FAST=33 TTN_WG512=1 run timeout -k 10 200 python tools/diag_determinism.py 512 || exit 1
echo "== cholqr on, 1024"; TTN_WG512=0 run timeout -k 10 200 python tools/diag_determinism.py 256 || exit 1
