#!/bin/bash
# A/B on one box: bash scratch/ab.sh base new [extra bench args]
L=$PWD/tensortrainnumerics.jl_amd; A=$1; B=$2; shift 2
for rep in 1 2; do for V in $A $B; do
  echo -n "$V: "; TTN_LIB=$L/libttn_$V.so python bench.py --no-cpu --no-single --no-verify --steps 4 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
for V in $A $B; do echo -n "$V B=1 (1024-thread build): "; TTN_LIB=$L/libttn_$V.so python tools/diag_batch.py 1 2>&1 | grep "iter 1"; done
for V in $A $B; do echo -n "$V B=1 wg512: "; TTN_LIB=$L/libttn_$V.so TTN_WG512=1 TTN_PROF=1 python tools/diag_batch.py 1 2>&1 | grep -E "iter 1|phase ticks"; done
