#!/bin/bash
OUT=gpurun_out
rm -f $OUT/r03c_ops.jsonl
for B in 1 8 32 64 256 512; do
  timeout -k 10 120 python bench.py --op dot --batch $B --steps 9 --warmup 3 >> $OUT/r03c_ops.jsonl 2>> $OUT/r03c_ops.err
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03c_ops.jsonl"):
    j = json.loads(ln); print(j["config"]["workload"], j["ms_per_step"], j["roofline"]["frac"])
PY
