#!/bin/bash
set -o pipefail
O=gpurun_out/r3s2e; mkdir -p $O
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r['stage_avg_ms'], 'ge3', d['config']['kmers_ge3'])"; }
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/v_head.json 2> $O/v_head.err && show $O/v_head.json head
for v in cepb8 c512; do
cp scratch/variants/libkdf_$v.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/v_$v.json 2> $O/v_$v.err && show $O/v_$v.json $v-span
KDF_DEBUG_FLAGS=128 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/v_${v}_one.json 2> $O/v_${v}_one.err && show $O/v_${v}_one.json $v-one
done
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
