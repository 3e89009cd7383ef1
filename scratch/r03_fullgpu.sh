#!/bin/bash
# the whole GPU suite, as the driver runs it, plus timing
timeout -k 10 1150 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r03_fullgpu.log 2>&1
echo "rc=$?"
tail -n 25 gpurun_out/r03_fullgpu.log
