#!/bin/bash
# round 3, first GPU contact of the stored-form + deferred-flush engine
set -o pipefail
mkdir -p gpurun_out/r3a
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3a/all.log 2>&1
rc=$?; echo "all rc=$rc" ; tail -15 gpurun_out/r3a/all.log
[ $rc -eq 0 ] && timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; tail -3 gpurun_out/r3a/bench.err; cat gpurun_out/r3a/bench.json
