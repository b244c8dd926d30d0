#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2h; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_fuzz.py tests/test_gpu_parity_basic.py -x -q -m gpu > $O/tests3.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests3.log
[ $rc -ne 0 ] && exit 1
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31c.json 2> $O/k31c.err && show $O/k31c.json k31
timeout -k 10 600 python bench.py --gpus 1 --scaling strong --batches 3 --steps 3 --warmup 2 --no-cpu-baseline > $O/strong3.json 2> $O/strong3.err && show $O/strong3.json strong3
timeout -k 10 500 python scratch/bigtable_probe.py > $O/bigtable3.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $O/bigtable3.txt | grep deferred
