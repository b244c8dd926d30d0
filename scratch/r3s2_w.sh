#!/bin/bash
O=gpurun_out/r3s2_w; mkdir -p $O
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so
timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/main.json 2> $O/main.err && show $O/main.json main
for v in w_e12 w_e4 w_la1; do
cp scratch/variants/libkdf_$v.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/$v.json 2> $O/$v.err && show $O/$v.json $v
done
cp scratch/variants/libkdf_w_la1.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/la1_k31.json 2> $O/la1_k31.err && show $O/la1_k31.json la1-k31
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
