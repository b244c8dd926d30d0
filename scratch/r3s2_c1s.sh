#!/bin/bash
set -o pipefail
O=gpurun_out/r3s2_c1s; mkdir -p $O
show() { python3 -c "import json,sys; d=json.load(open('$1')); r=d['roofline']; print('$2', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'], r.get('stage_avg_ms'), 'ge3', d['config'].get('kmers_ge3'))"; }
for c1 in default 9 default 9; do
  if [ $c1 = default ]; then unset KDF_C1; else export KDF_C1=$c1; fi
  timeout -k 10 300 python bench.py --scaling strong --batches 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/s_$c1.json 2> $O/s_$c1.err && show $O/s_$c1.json "strong8 c1=$c1"
done
