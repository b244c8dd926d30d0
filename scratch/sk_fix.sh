#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_superkmer.py -x -q 2>&1 | tail -6 && timeout -k 10 800 python scratch/skew_check.py 2>&1 | grep -v amdgpu.ids | tail -6 && timeout -k 10 800 python scratch/skew_probe.py 2>&1 | grep -v amdgpu.ids | tail -5
