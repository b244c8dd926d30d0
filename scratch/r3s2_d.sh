#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2_d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_fuzz.py tests/test_gpu_trio_golden.py tests/test_gpu_configs.py tests/test_gpu_skew.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so
for k in 31 63; do
timeout -k 10 300 python bench.py --k $k --steps 10 --warmup 2 --no-cpu-baseline > $O/new_k$k.json 2> $O/new_k$k.err && show $O/new_k$k.json lane-dummies-k$k
cp scratch/variants/libkdf_onedummy.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python bench.py --k $k --steps 10 --warmup 2 --no-cpu-baseline > $O/old_k$k.json 2> $O/old_k$k.err && show $O/old_k$k.json one-dummy-k$k
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
done
