#!/bin/bash
# fused dump: parity subset, then bench with the new library (fused / KDF_FUSED_DUMP=0) and the previous one
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2_fd; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_fuzz.py tests/test_gpu_skew.py tests/test_gpu_scale.py tests/test_gpu_trio_golden.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 1
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31_fused.json 2> $O/k31_fused.err && show $O/k31_fused.json "k31 fused"
  KDF_FUSED_DUMP=0 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31_unfused.json 2> $O/k31_unfused.err && show $O/k31_unfused.json "k31 unfused"
  cp scratch/variants/libkdf_head.so kmer_denovo_filter_amd/libkdf.so
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31_head.json 2> $O/k31_head.err && show $O/k31_head.json "k31 head"
  cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
done
timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/k63_fused.json 2> $O/k63_fused.err && show $O/k63_fused.json "k63 fused"
KDF_FUSED_DUMP=0 timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/k63_unfused.json 2> $O/k63_unfused.err && show $O/k63_unfused.json "k63 unfused"
timeout -k 10 300 python bench.py --scaling strong --batches 8 --steps 5 --warmup 1 --no-cpu-baseline > $O/strong_fused.json 2> $O/strong_fused.err && show $O/strong_fused.json "strong8 fused"
KDF_FUSED_DUMP=0 timeout -k 10 300 python bench.py --scaling strong --batches 8 --steps 5 --warmup 1 --no-cpu-baseline > $O/strong_unfused.json 2> $O/strong_unfused.err && show $O/strong_unfused.json "strong8 unfused"
