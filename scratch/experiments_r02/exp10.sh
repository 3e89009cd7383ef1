#!/bin/bash
set -o pipefail
run() { "$@" 2>&1 | tee -a gpurun_out/exp10.raw | grep -E "phase ticks|iter 1|cores/s|passed|failed" | cut -c1-700; if grep -q "GPU core dump" gpurun_out/exp10.raw; then echo "GPU FAULT"; exit 1; fi; }
rm -f gpurun_out/exp10.raw
for st in 5 52; do echo "== step $st"; TTN_PROF=1 TTN_PROF_STEP=$st TTN_WG512=1 run timeout -k 10 120 python tools/diag_batch.py 512 || exit 1; done
echo "== B=1 step 5"; TTN_PROF=1 TTN_PROF_STEP=5 run timeout -k 10 120 python tools/diag_batch.py 1 || exit 1
for i in 1 2; do
run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 4 || exit 1
TTN_FAST=33 run timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 4 || exit 1
done
