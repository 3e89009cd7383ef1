#!/bin/bash
# per-step phase breakdown (512-thread build, two workgroups per CU, B = 512): TTN_PROF_STEP selects the step
for st in 5 10 24 25 34 40 52; do
  echo "== step $st"
  TTN_PROF=1 TTN_PROF_STEP=$st TTN_WG512=1 python tools/diag_batch.py 512 2>&1 | grep -E "phase ticks"
done
echo "== all steps"; TTN_PROF=1 TTN_WG512=1 python tools/diag_batch.py 512 2>&1 | grep -E "iter 1|phase ticks|per step"
echo "== all steps, 1024 threads, B=1"; TTN_PROF=1 python tools/diag_batch.py 1 2>&1 | grep -E "iter 1|phase ticks|per step"
for st in 5 10 34 40; do
  echo "== B=1 step $st"
  TTN_PROF=1 TTN_PROF_STEP=$st python tools/diag_batch.py 1 2>&1 | grep -E "phase ticks"
done
