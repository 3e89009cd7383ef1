#!/bin/bash
# k_apply: output columns per thread (TTN_APPLY_K), rebuilt on the box for each value
cd tensortrainnumerics.jl_amd/csrc
for K in 1 2 3 4; do
  sed -i "s/^#define TTN_APPLY_K [0-9]*/#define TTN_APPLY_K $K/" ttn_stream_kernels.h
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-pass-failed -o ../libttn_hip.so ttn_api.hip ttn_wg512.hip 2>&1 | grep -E "error" | head -3
  (cd ../.. && timeout -k 10 120 python bench.py --op apply --batch 1024 --steps 7 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('K=$K', j['ms_per_step'], j['roofline']['frac'])")
done
