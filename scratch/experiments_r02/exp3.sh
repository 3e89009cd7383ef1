#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_pipeline_gloo.py -x -q --timeout 500 > gpurun_out/r02h_pipe.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r02h_pipe.log
python bench.py --shard cores --batch 128 --microbatches 2 --steps 3 --warmup 1 > gpurun_out/r02h_pipe_1rank.json 2>&1; tail -c 400 gpurun_out/r02h_pipe_1rank.json; echo
python bench.py --gpus 2 --backend gloo --shard cores --batch 64 --microbatches 4 --steps 3 --warmup 1 > gpurun_out/r02h_pipe_2rank.json 2>&1; tail -c 400 gpurun_out/r02h_pipe_2rank.json; echo
python bench.py --gpus 2 --backend gloo --shard cores --batch 128 --microbatches 4 --steps 3 --warmup 1 > gpurun_out/r02h_pipe_2rank_b128.json 2>&1; tail -c 400 gpurun_out/r02h_pipe_2rank_b128.json; echo
python bench.py --gpus 2 --backend gloo --no-cpu --no-single --no-verify --batch 512 > gpurun_out/r02h_trains_2rank.json 2>&1; tail -c 300 gpurun_out/r02h_trains_2rank.json
