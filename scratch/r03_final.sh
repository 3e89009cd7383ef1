#!/bin/bash
# round 3, final evidence: default bench line, profiles (stats + PMC + op lines at B = 256), dot / orthogonalize over batch sizes
set -o pipefail
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/r03h_bench.json 2> $OUT/r03h_bench.err
echo "bench rc=$?"
cut -c1-400 $OUT/r03h_bench.json
OPS=1 timeout -k 10 900 bash scratch/collect_profiles.sh r03h > $OUT/r03h_collect.log 2>&1
echo "collect rc=$?"
rm -f $OUT/r03h_ops.jsonl
for OP in dot orthogonalize; do
  for B in 1 16 256 1024 4096; do
    timeout -k 10 120 python bench.py --op $OP --batch $B --steps 5 --warmup 2 >> $OUT/r03h_ops.jsonl 2>> $OUT/r03h_ops.err
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r03h_ops.jsonl"):
    j = json.loads(ln); print(j["config"]["workload"], j["ms_per_step"], j["roofline"]["frac"])
PY
