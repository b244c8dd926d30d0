#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_scale.py -x -q 2>&1 | tail -3 && \
for k in 63 47 33; do timeout -k 10 300 python bench.py --k $k --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['k'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['stage_avg_ms'])"; done
