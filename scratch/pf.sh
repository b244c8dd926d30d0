#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pf
python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_trio_golden.py tests/test_gpu_scale.py tests/test_gpu_synthetic_scenarios.py -x -q -m gpu 2>&1 | tail -8
python benchmarks/parent_filter.py > gpurun_out/pf/parent_filter.json 2> gpurun_out/pf/err.log; tail -2 gpurun_out/pf/err.log; cat gpurun_out/pf/parent_filter.json
python bench.py --config parent_filter --steps 5 --warmup 1 > gpurun_out/pf/bench_pf.json 2> gpurun_out/pf/bench_pf.err; tail -3 gpurun_out/pf/bench_pf.err; cat gpurun_out/pf/bench_pf.json
