#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 0 16 64 192 448 960; do echo "flags $f"; python scratch/sk_probe.py 10000000 268435456 $f | tail -1 | cut -c1-330; done
