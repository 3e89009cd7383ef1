#!/bin/bash
# same-box A/B of library builds: bash scratch/ab2.sh NAME1 NAME2 ... (tensortrainnumerics.jl_amd/libttn_NAME.so), 3 interleaved rounds
L=$PWD/tensortrainnumerics.jl_amd
for rep in 1 2 3; do for V in "$@"; do
  echo -n "$V: "; TTN_LIB=$L/libttn_$V.so timeout -k 10 200 python bench.py --no-cpu --no-single --no-verify --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
for V in "$@"; do echo -n "$V B=1: "; TTN_LIB=$L/libttn_$V.so timeout -k 10 100 python tools/diag_batch.py 1 2>&1 | grep "iter 1"; done
for V in "$@"; do echo -n "$V B=512 one train per slot: "; TTN_LIB=$L/libttn_$V.so TTN_WG512=1 timeout -k 10 100 python tools/diag_batch.py 512 2>&1 | grep "iter 1"; done
