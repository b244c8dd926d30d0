#!/bin/bash
cd $GRAFT_REPO_ROOT
for args in "--scaling strong --batches 4 --reads 2000000" "--config parent_filter" "--k 63 --reads 3000000"; do
  echo "== $args"
  timeout -k 10 500 python bench.py --gpus 2 --rehearse-one-gpu $args --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/reh.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['n_gpus'], d['scaling'], d['ms_per_step'], c.get('kmers_ge3'), c.get('multi_gpu'))" || tail -5 gpurun_out/reh.err
done
