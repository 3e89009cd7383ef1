#!/bin/bash
for st in 0 1 3 28 31 55 57; do echo "== B=1 step $st"; TTN_PROF=1 TTN_PROF_STEP=$st python tools/diag_batch.py 1 2>&1 | grep -E "phase ticks" | cut -c1-260; done
for st in 1 31; do echo "== co-res step $st"; TTN_PROF=1 TTN_PROF_STEP=$st TTN_WG512=1 python tools/diag_batch.py 512 2>&1 | grep -E "phase ticks" | cut -c1-260; done
