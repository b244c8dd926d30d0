#!/bin/bash
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2e; mkdir -p $O
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r['stage_avg_ms'], 'ge3', d['config']['kmers_ge3'])"; }
for k in 31; do
timeout -k 10 300 python bench.py --k $k --steps 10 --warmup 2 --no-cpu-baseline > $O/k$k.json 2> $O/k$k.err && show $O/k$k.json span-k$k
KDF_DEBUG_FLAGS=128 timeout -k 10 300 python bench.py --k $k --steps 10 --warmup 2 --no-cpu-baseline > $O/k${k}_old.json 2> $O/k${k}_old.err && show $O/k${k}_old.json one-k$k
done
