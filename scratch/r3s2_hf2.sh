#!/bin/bash
O=gpurun_out/r3s2_hf; mkdir -p $O
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/k31.json 2> $O/k31.err && show $O/k31.json k31
timeout -k 10 300 python scratch/skew_filtered_probe.py > $O/sf.txt 2>&1; echo split; grep "^{" $O/sf.txt
K=63 timeout -k 10 300 python scratch/skew_filtered_probe.py > $O/sf63.txt 2>&1; grep "^{" $O/sf63.txt
cp kmer_denovo_filter_amd/libkdf.so /tmp/keep.so; cp scratch/variants/libkdf_noheavy.so kmer_denovo_filter_amd/libkdf.so
timeout -k 10 300 python scratch/skew_filtered_probe.py > $O/sf_no.txt 2>&1; echo no-split; grep "^{" $O/sf_no.txt
K=63 timeout -k 10 300 python scratch/skew_filtered_probe.py > $O/sf63_no.txt 2>&1; grep "^{" $O/sf63_no.txt
cp /tmp/keep.so kmer_denovo_filter_amd/libkdf.so
