#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_merge.py tests/test_gpu_scale.py -x -q 2>&1 | tail -6 && \
timeout -k 10 600 python bench.py --gpus 2 --rehearse-one-gpu --reads 3000000 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/rehearse.err | cut -c1-200
tail -3 gpurun_out/rehearse.err
