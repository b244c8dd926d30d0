#!/bin/bash
# A/B of the product library against scratch/variants/libkdf_$1.so on the same box: quick parity subset first, then
# the headline bench (k = 31, k = 63) with each library, twice (new, old, new, old)
set -o pipefail
ulimit -c 0
V=${1:-rt_old}
O=gpurun_out/r3s2_rt; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_fuzz.py tests/test_gpu_skew.py tests/test_gpu_scale.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'))"; }
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so
for rep in 1 2; do
  for lib in new $V; do
    if [ $lib = new ]; then cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so; else cp scratch/variants/libkdf_$lib.so kmer_denovo_filter_amd/libkdf.so; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31_$lib.json 2> $O/k31_$lib.err && show $O/k31_$lib.json "k31 $lib"
    timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/k63_$lib.json 2> $O/k63_$lib.err && show $O/k63_$lib.json "k63 $lib"
  done
done
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
