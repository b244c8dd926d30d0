#!/bin/bash
# first GPU run of the super-k-mer path: its parity tests, then the bench with the path auto-selected
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sk1
timeout -k 10 900 python -m pytest tests/test_gpu_superkmer.py -x -q -m gpu > gpurun_out/sk1/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/sk1/tests.log
tail -30 gpurun_out/sk1/tests.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/sk1/bench.json 2> gpurun_out/sk1/bench.err
echo "bench rc=$?"; tail -3 gpurun_out/sk1/bench.err; cat gpurun_out/sk1/bench.json
