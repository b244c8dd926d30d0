#!/bin/bash
set -o pipefail
ulimit -c 0
mkdir -p gpurun_out/r3f
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3f/all.log 2>&1
rc=$?; echo "all rc=$rc" ; tail -6 gpurun_out/r3f/all.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python scratch/fuzz_parity.py 11 150 > gpurun_out/r3f/fuzz1.txt 2>&1; echo "fuzz1 rc=$?"; tail -2 gpurun_out/r3f/fuzz1.txt
timeout -k 10 200 python scratch/fuzz_round2.py 12 150 > gpurun_out/r3f/fuzz2.txt 2>&1; echo "fuzz2 rc=$?"; tail -2 gpurun_out/r3f/fuzz2.txt
timeout -k 10 200 python scratch/fuzz_parity.py 13 120 4 > gpurun_out/r3f/fuzz3.txt 2>&1; echo "fuzz3 rc=$?"; tail -2 gpurun_out/r3f/fuzz3.txt
