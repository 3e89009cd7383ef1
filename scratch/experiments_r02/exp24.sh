#!/bin/bash
TTN_PROF=1 TTN_WG512=1 python tools/diag_batch.py 512 2>&1 | grep -E "iter 1|phase ticks|per step" | cut -c1-1500
TTN_PROF_STEP=10 TTN_WG512=1 python tools/diag_fine.py 512 2>&1 | grep -E "step|  G|  eig|  merge"
TTN_PROF_STEP=40 TTN_WG512=1 python tools/diag_fine.py 512 2>&1 | grep -E "step|  F"
