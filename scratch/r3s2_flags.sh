#!/bin/bash
# compiler scheduling flags: the k = 31 / k = 63 bench with each variant library, product library first and last
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2_flags; mkdir -p $O
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so
for lib in product "$@" product; do
  if [ $lib = product ]; then cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so; else cp scratch/variants/libkdf_$lib.so kmer_denovo_filter_amd/libkdf.so; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31_$lib.json 2> $O/k31_$lib.err && show $O/k31_$lib.json "k31 $lib"
  timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/k63_$lib.json 2> $O/k63_$lib.err && show $O/k63_$lib.json "k63 $lib"
done
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
