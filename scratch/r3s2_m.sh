#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2_m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_fuzz.py tests/test_gpu_skew.py tests/test_gpu_scale.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/new.json 2> $O/new.err && show $O/new.json mirror
cp scratch/variants/libkdf_nomirror.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/old.json 2> $O/old.err && show $O/old.json nomirror
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/new2.json 2> $O/new2.err && show $O/new2.json mirror-again
timeout -k 10 600 python bench.py --gpus 1 --scaling strong --batches 8 --steps 3 --warmup 2 --no-cpu-baseline > $O/strong.json 2> $O/strong.err && show $O/strong.json strong8
