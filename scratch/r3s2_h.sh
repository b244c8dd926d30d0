#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_skew.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -6 $O/tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python scratch/skew_probe.py > $O/skew.txt 2>&1; grep -v amdgpu $O/skew.txt | tail -4; K=63 timeout -k 10 300 python scratch/skew_probe.py > $O/skew63.txt 2>&1; grep -v amdgpu $O/skew63.txt | tail -4
