#!/bin/bash
# round 3, first GPU pass: new parity tests + the bench line with its new sub-records + dot / orthogonalize in both builds
set -o pipefail
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "extended_precision or full_size_vs_oracle or krylov or rk4_and_euler or orthogonalize" > $OUT/r03a_tests1.log 2>&1
echo "tests1 rc=$?" | tee -a $OUT/r03a_summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_c5_laplace.py tests/test_gpu_dmrg.py tests/test_pipeline_gloo.py -x -q -m gpu -s -k "not 128" > $OUT/r03a_tests2.log 2>&1
echo "tests2 rc=$?" | tee -a $OUT/r03a_summary.txt
timeout -k 10 600 python bench.py > $OUT/r03a_bench.json 2> $OUT/r03a_bench.err
echo "bench rc=$?" | tee -a $OUT/r03a_summary.txt
for OP in dot orthogonalize; do
  for B in 256 1024 2048; do
    timeout -k 10 120 python bench.py --op $OP --batch $B --steps 5 --warmup 2 >> $OUT/r03a_ops.jsonl 2>> $OUT/r03a_ops.err
  done
done
echo "ops rc=$?" | tee -a $OUT/r03a_summary.txt
tail -3 $OUT/r03a_tests1.log $OUT/r03a_tests2.log
cut -c1-600 $OUT/r03a_bench.json
python - <<'PY'
import json
for ln in open("gpurun_out/r03a_ops.jsonl"):
    j = json.loads(ln); print(j["config"]["workload"], j["ms_per_step"], j["roofline"]["frac"])
PY
