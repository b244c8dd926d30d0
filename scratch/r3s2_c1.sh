#!/bin/bash
O=gpurun_out/r3s2_c1; mkdir -p $O
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
for c1 in 9 8 10; do
KDF_C1=$c1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/c1_$c1.json 2> $O/c1_$c1.err && show $O/c1_$c1.json c1=$c1
done
for c1 in 9 8; do
KDF_C1=$c1 timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/k63_c1_$c1.json 2> $O/k63_c1_$c1.err && show $O/k63_c1_$c1.json k63-c1=$c1
done
