#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_superkmer.py -x -q -m gpu 2>&1 | tail -3
python scratch/sk_probe.py 10000000 268435456 0 | tail -1 | cut -c1-330
python scratch/sk_probe.py 10000000 268435456 32 | tail -1 | cut -c1-330
