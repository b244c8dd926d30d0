#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py -x -q 2>&1 | tail -5 && \
SKEW_PATHS=1 timeout -k 10 600 python scratch/skew_probe.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['genome'], d['wall_ms'], d['stage_ms'], d['distinct'], d['max_count'])" && \
timeout -k 10 600 python scratch/skew_check.py 2>&1 | grep -v amdgpu | tail -3 | cut -c1-150
