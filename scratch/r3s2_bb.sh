#!/bin/bash
set -o pipefail
O=gpurun_out/r3s2_bb; mkdir -p $O
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/main.json 2> $O/main.err && show $O/main.json main
for v in b_t0 b_t0e0; do
cp scratch/variants/libkdf_$v.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/$v.json 2> $O/$v.err && show $O/$v.json $v
timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/${v}_k63.json 2> $O/${v}_k63.err && show $O/${v}_k63.json $v-k63
done
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/main_k63.json 2> $O/main_k63.err && show $O/main_k63.json main-k63
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/main2.json 2> $O/main2.err && show $O/main2.json main-again
