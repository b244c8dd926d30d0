#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3a
timeout -k 10 600 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_skew.py -x -q -m gpu > gpurun_out/r3a/basic.log 2>&1
rc=$?; echo "basic rc=$rc" ; tail -25 gpurun_out/r3a/basic.log
[ $rc -eq 0 ] && timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; tail -3 gpurun_out/r3a/bench.err; cat gpurun_out/r3a/bench.json
