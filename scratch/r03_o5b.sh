#!/bin/bash
# orthogonalize: parity tests (every route), then both forms at B = 1, 8, 256, 1024
set -e
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "orthogonalize" > gpurun_out/o5_tests.log 2>&1 || { tail -30 gpurun_out/o5_tests.log; exit 1; }
tail -2 gpurun_out/o5_tests.log
for B in 1 8 256 1024; do
  for F in 0 1; do
    TTN_ORTHO512=$F timeout -k 10 120 python bench.py --op orthogonalize --batch $B --steps 5 --warmup 2 --no-c2 --no-core-sharded 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('B=$B form=$F', j['ms_per_step'], j['roofline']['frac'])"
  done
done
