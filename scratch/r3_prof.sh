#!/bin/bash
# tests (quick subset) + bench + rocprofv3 kernel stats of the bench command
set -o pipefail
mkdir -p gpurun_out/r3p
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_skew.py -x -q -m gpu > gpurun_out/r3p/basic.log 2>&1
rc=$?; echo "basic rc=$rc" ; tail -5 gpurun_out/r3p/basic.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err; cat gpurun_out/r3p/bench.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_avg_ms'])"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r3p/prof -o run -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3p/prof.log 2>&1
f=$(find gpurun_out/r3p/prof -name "*kernel_stats.csv" | head -1); echo $f; head -14 "$f" | cut -c1-200
