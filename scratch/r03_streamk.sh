#!/bin/bash
# streaming kernels: fibres per thread, rebuilt on the box for each value:  bash scratch/r03_streamk.sh MACRO op "values"
cd tensortrainnumerics.jl_amd/csrc
M=$1; OP=$2
for K in $3; do
  sed -i "s/^#define $M [0-9]*/#define $M $K/" ttn_stream_kernels.h
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-pass-failed -o ../libttn_hip.so ttn_api.hip ttn_wg512.hip 2>&1 | grep -E "error" | head -3
  (cd ../.. && timeout -k 10 120 python bench.py --op $OP --batch 1024 --steps 7 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$OP $M=$K', j['ms_per_step'], j['roofline']['frac'])")
done
