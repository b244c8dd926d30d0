#!/bin/bash
set -o pipefail
ulimit -c 0
mkdir -p gpurun_out/r3pk
timeout -k 10 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_configs.py tests/test_gpu_parity_basic.py -x -q -m gpu > gpurun_out/r3pk/t.log 2>&1
rc=$?; echo "tests rc=$rc" ; tail -8 gpurun_out/r3pk/t.log
[ $rc -ne 0 ] && exit 1
for fl in 0 64 0 64; do
KDF_DEBUG_FLAGS=$fl timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3pk/b$fl.json 2> gpurun_out/r3pk/b$fl.err; python3 -c "import json; d=json.load(open('gpurun_out/r3pk/b$fl.json')); print('flags $fl', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_avg_ms'], d['config']['kmers_ge3'])"
done
