#!/bin/bash
# three-launch orthogonalize: parity tests, state diagnostics, bench at B=1024/256
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "orthogonalize" > gpurun_out/o5_tests.log 2>&1 || { tail -30 gpurun_out/o5_tests.log; exit 1; }
tail -3 gpurun_out/o5_tests.log
TTN_ORTHO512=1 timeout -k 10 200 python tools/diag_ortho_prof.py 1024 > gpurun_out/o5_diag.log 2>&1 || true
tail -8 gpurun_out/o5_diag.log
timeout -k 10 200 python bench.py --op orthogonalize --batch 1024 --no-c2 --no-core-sharded 2>&1 | tail -1 > gpurun_out/o5_bench1024.json
cat gpurun_out/o5_bench1024.json
TTN_ORTHO512=0 timeout -k 10 200 python bench.py --op orthogonalize --batch 1024 --no-c2 --no-core-sharded 2>&1 | tail -1 > gpurun_out/o5_bench1024_single.json
cat gpurun_out/o5_bench1024_single.json
