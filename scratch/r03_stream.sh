#!/bin/bash
for OP in apply add scale hadamard; do
  for B in 256 1024 4096; do
    timeout -k 10 120 python bench.py --op $OP --batch $B --steps 7 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$OP B=$B', j['ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'])"
  done
done
