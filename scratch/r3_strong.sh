#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3b
timeout -k 10 600 python bench.py --gpus 1 --scaling strong --batches 8 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r3b/strong_n1.json 2> gpurun_out/r3b/strong_n1.err
echo "strong rc=$?"; tail -2 gpurun_out/r3b/strong_n1.err; cat gpurun_out/r3b/strong_n1.json
timeout -k 10 500 python scratch/bigtable_probe.py > gpurun_out/r3b/bigtable.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids gpurun_out/r3b/bigtable.txt | tail -20
