#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_scale.py tests/test_gpu_skew.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r['stage_avg_ms'], 'ge3', d['config']['kmers_ge3'])"; }
cp kmer_denovo_filter_amd/libkdf.so /tmp/libkdf_keep.so
for k in 31 63; do
timeout -k 10 300 python bench.py --k $k --steps 10 --warmup 2 --no-cpu-baseline > $O/k$k.json 2> $O/k$k.err && show $O/k$k.json new-k$k
cp scratch/variants/libkdf_a_old.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python bench.py --k $k --steps 10 --warmup 2 --no-cpu-baseline > $O/k${k}_old.json 2> $O/k${k}_old.err && show $O/k${k}_old.json oldA-k$k
cp /tmp/libkdf_keep.so kmer_denovo_filter_amd/libkdf.so
done
