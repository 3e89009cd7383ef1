#!/bin/bash
set -o pipefail
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dot or norm or golden or headline_config_properties or batched_handles or krylov" > $OUT/r03b_tests.log 2>&1
echo "tests rc=$?"
tail -n 3 $OUT/r03b_tests.log
rm -f $OUT/r03b_ops.jsonl
for B in 256 1024; do
  timeout -k 10 120 python bench.py --op dot --batch $B --steps 5 --warmup 2 >> $OUT/r03b_ops.jsonl 2>> $OUT/r03b_ops.err
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03b_ops.jsonl"):
    j = json.loads(ln); print(j["config"]["workload"], j["ms_per_step"], j["roofline"]["frac"])
PY
