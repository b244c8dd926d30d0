#!/bin/bash
set -o pipefail
O=gpurun_out/r3s2f; mkdir -p $O
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r['stage_avg_ms'], 'ge3', d['config']['kmers_ge3'])"; }
for f in 0.93 0.97 1.0 1.03; do
KDF_PIECE_FILL=$f timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/f$f.json 2> $O/f$f.err && show $O/f$f.json fill-$f
done
KDF_PIECE_FILL=1.0 timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/k63f.json 2> $O/k63f.err && show $O/k63f.json k63-fill-1.0
timeout -k 10 300 python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/k63.json 2> $O/k63.err && show $O/k63.json k63-fill-0.93
