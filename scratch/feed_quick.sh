#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_feeding.py tests/test_gpu_trio_golden.py -x -q 2>&1 | tail -8 && \
timeout -k 10 600 python scratch/bench_e2e_bam.py 2>&1 | tail -5
