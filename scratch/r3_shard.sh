#!/bin/bash
set -o pipefail
ulimit -c 0
mkdir -p gpurun_out/r3d
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_mirrors.py tests/test_gpu_trio_golden.py tests/test_gpu_feeding.py -x -q -m gpu > gpurun_out/r3d/shard.log 2>&1
rc=$?; echo "shard rc=$rc"; tail -25 gpurun_out/r3d/shard.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python scratch/bench_e2e_bam.py > gpurun_out/r3d/e2e.txt 2>&1; echo "e2e rc=$?"; grep -v amdgpu.ids gpurun_out/r3d/e2e.txt | tail -16
