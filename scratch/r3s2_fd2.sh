#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2_fd2; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_skew.py tests/test_gpu_scale.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
KDF_FUSED_DUMP=1 timeout -k 10 600 python -m pytest tests/test_gpu_trio_golden.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_merge.py -x -q -m gpu > $O/tests2.log 2>&1
rc=$?; echo "tests (fused on) rc=$rc"; tail -3 $O/tests2.log
[ $rc -ne 0 ] && exit 1
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'))"; }
for rep in 1 2; do
  KDF_FUSED_DUMP=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31_fused.json 2> $O/k31_fused.err && show $O/k31_fused.json "k31 fused"
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31_default.json 2> $O/k31_default.err && show $O/k31_default.json "k31 default"
done
KDF_FUSED_DUMP=1 timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/k63_fused.json 2> $O/k63_fused.err && show $O/k63_fused.json "k63 fused"
timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/k63_default.json 2> $O/k63_default.err && show $O/k63_default.json "k63 default"
