#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2_hf; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_skew.py tests/test_gpu_parity_basic.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 $O/tests.log
