#!/bin/bash
L=$PWD/tensortrainnumerics.jl_amd
for V in "" vg256; do
  LIB=$L/libttn_hip.so; [ -n "$V" ] && LIB=$L/libttn_$V.so
  echo "== variant ${V:-default}"
  TTN_LIB=$LIB TTN_WG512=1 TTN_PROF=1 python tools/diag_batch.py 1 2>&1 | grep -E "iter 1|phase ticks"
  TTN_LIB=$LIB TTN_WG512=1 python tools/diag_batch.py 256 2>&1 | grep -E "iter 1"
  TTN_LIB=$LIB TTN_WG512=1 python tools/diag_batch.py 512 2>&1 | grep -E "iter 1"
done
