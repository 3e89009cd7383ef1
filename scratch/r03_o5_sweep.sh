#!/bin/bash
# orthogonalize: single-launch form vs ramp + 512-thread sweep, over batch sizes
OUT=gpurun_out
for B in 1 16 64 128 256 512 2048; do
  for F in 0 1; do
    TTN_ORTHO512=$F timeout -k 10 120 python bench.py --op orthogonalize --batch $B --steps 5 --warmup 2 --no-c2 --no-core-sharded 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('B=$B form=$F', j['ms_per_step'], j['roofline']['frac'])" || exit 1
  done
done
