#!/bin/bash
# Experiment builds: bash scratch/build_variant.sh NAME [-DFLAG ...]  ->  tensortrainnumerics.jl_amd/libttn_NAME.so  (use with TTN_LIB=...)
set -e
NAME=$1; shift
D=tensortrainnumerics.jl_amd
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wno-unused-value "$@" -o $D/libttn_$NAME.so $D/csrc/ttn_api.hip $D/csrc/ttn_wg512.hip 2>&1 | grep -E "error" -A5 || true
ls -la $D/libttn_$NAME.so
