#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py -x -q 2>&1 | tail -3 && \
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['stage_avg_ms'])" && \
SKEW_PATHS=1 timeout -k 10 600 python scratch/skew_probe.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['genome'], d['wall_ms'], d['stage_ms'], d['distinct'], d.get('heavy_buckets'))" && \
timeout -k 10 600 python scratch/skew_check.py 2>&1 | grep -v amdgpu | tail -2
