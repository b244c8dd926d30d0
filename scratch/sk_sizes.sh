#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_superkmer.py -x -q -m gpu 2>&1 | tail -3
for h in 268435456 536870912; do for f in 0 16; do echo "hint $h flags $f"; python scratch/sk_probe.py 10000000 $h $f | tail -1 | cut -c1-330; done; done
