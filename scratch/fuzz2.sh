#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 420 python scratch/fuzz_round2.py 11 300 1 2>&1 | tail -4 && timeout -k 10 420 python scratch/fuzz_round2.py 12 240 6 2>&1 | tail -4
