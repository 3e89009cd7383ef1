#!/bin/bash
TTN_PROF_STEP=10 TTN_WG512=0 python tools/diag_fine.py 1 2>&1 | grep -E "step|G 2|merge"
TTN_PROF_STEP=10 TTN_WG512=1 python tools/diag_fine.py 1 2>&1 | grep -E "step|G 2|merge"
TTN_PROF_STEP=10 TTN_WG512=1 python tools/diag_fine.py 512 2>&1 | grep -E "step|G 2|merge"
