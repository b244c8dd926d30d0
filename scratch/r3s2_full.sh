#!/bin/bash
# full -m gpu suite + the bench lines of the evidence set
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2g; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err && show $O/bench_default.json default
timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_k63.json 2> $O/bench_k63.err && show $O/bench_k63.json k63
