#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2_merge; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_merge.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py tests/test_gpu_sharded_mirrors.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python scratch/merge_probe.py 8 > $O/merge8.txt 2>&1; grep -v amdgpu $O/merge8.txt | tail -3 | cut -c1-330
timeout -k 10 300 python scratch/merge_probe.py 2 > $O/merge2.txt 2>&1; grep -v amdgpu $O/merge2.txt | tail -1 | cut -c1-330
