#!/bin/bash
# experiment batch 1 (GPU box)
for OP in apply hadamard add scale dot orthogonalize; do python bench.py --op $OP --batch 256 --steps 5 --warmup 2; done > gpurun_out/r02_ops.jsonl 2> gpurun_out/r02_ops.err
TTN_LIB=$PWD/tensortrainnumerics.jl_amd/libttn_prio3.so python bench.py --no-cpu --no-single --no-verify > gpurun_out/r02_prio3.json 2>&1
python bench.py --no-cpu --no-single --no-verify --batch 2048 > gpurun_out/r02_b2048.json 2>&1
python bench.py --no-cpu --no-single --no-verify --batch 512 > gpurun_out/r02_b512.json 2>&1
TTN_WG512=1 python bench.py --no-cpu --no-single --no-verify --batch 256 > gpurun_out/r02_b256_wg512.json 2>&1
python - <<'PY'
import json
for f in ["r02_prio3","r02_b2048","r02_b512","r02_b256_wg512"]:
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
for l in open("gpurun_out/r02_ops.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["workload"][:20], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["unit"], d["roofline"]["frac"])
PY
