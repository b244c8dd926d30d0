#!/bin/bash
set -o pipefail
ulimit -c 0
mkdir -p gpurun_out/r3w
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_skew.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3w/wide.log 2>&1
rc=$?; echo "tests rc=$rc" ; tail -8 gpurun_out/r3w/wide.log
[ $rc -ne 0 ] && exit 1
for k in 63 47 31; do
timeout -k 10 300 python bench.py --k $k --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r3w/k$k.json 2> gpurun_out/r3w/k$k.err; python3 -c "import json; d=json.load(open('gpurun_out/r3w/k$k.json')); print('k$k', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_avg_ms'], d['config']['kmers_ge3'])"
done
