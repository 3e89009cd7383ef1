#!/bin/bash
for MODE in 1 2; do for ST in 0 400000 800000 1200000 2000000; do
  echo -n "mode $MODE stagger $ST: "
  TTN_STAGGER=$ST TTN_STAGGER_MODE=$MODE python bench.py --no-cpu --no-single --no-verify --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
