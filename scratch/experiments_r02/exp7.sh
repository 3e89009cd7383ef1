#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wg512.py -x -q --timeout 300 -k "fused or headline or apply_compress or bench_batch or randomized" 2>&1 | tail -3
echo "v2 on:"; python bench.py --no-cpu --no-single --no-verify --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "v2 off:"; TTN_FAST=129 python bench.py --no-cpu --no-single --no-verify --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "v2 on:"; python bench.py --no-cpu --no-single --no-verify --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
for F in 1 129; do echo "TTN_FAST=$F B=1 1024:"; TTN_FAST=$F TTN_PROF=1 python tools/diag_batch.py 1 2>&1 | grep -E "iter 1|phase ticks"; echo "wg512:"; TTN_FAST=$F TTN_WG512=1 TTN_PROF=1 python tools/diag_batch.py 1 2>&1 | grep -E "iter 1|phase ticks"; done
