#!/bin/bash
for F in 1 513 129; do for W in 0 1; do echo "== TTN_FAST=$F TTN_WG512=$W B=1 step 10"; TTN_FAST=$F TTN_WG512=$W TTN_PROF=1 TTN_PROF_STEP=10 python tools/diag_batch.py 1 2>&1 | grep "phase ticks" | cut -c1-120; done; done
echo "== co-res step 10"; for F in 1 513; do TTN_FAST=$F TTN_WG512=1 TTN_PROF=1 TTN_PROF_STEP=10 python tools/diag_batch.py 512 2>&1 | grep "phase ticks" | cut -c1-120; done
