#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_merge.py -x -q 2>&1 | tail -8 && \
timeout -k 10 600 python bench.py --gpus 2 --rehearse-one-gpu --reads 3000000 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/rehearse.err | tail -2
tail -5 gpurun_out/rehearse.err
