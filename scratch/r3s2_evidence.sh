#!/bin/bash
# round 3, session 2: the evidence set at HEAD -- bench lines (default with CPU baseline, k = 63, strong-8, parent filter),
# rocprofv3 kernel stats of the same commands, the big-table probe.  Summaries are copied to profiles/ afterwards.
set -o pipefail
ulimit -c 0
O=gpurun_out/r3s2_ev; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
timeout -k 10 300 python3 bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_k63.json 2> $O/bench_k63.err; echo "k63 rc=$?"
timeout -k 10 600 python3 bench.py --gpus 1 --scaling strong --batches 8 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_strong_n1.json 2> $O/bench_strong_n1.err; echo "strong rc=$?"
timeout -k 10 500 python3 bench.py --config parent_filter --scaling strong --steps 10 --warmup 2 > $O/bench_parent_filter.json 2> $O/bench_parent_filter.err; echo "pf rc=$?"
for cfg in "count:" "k63:--k 63" "strong8:--scaling strong --batches 8 --steps 2 --warmup 1" "parent_filter:--config parent_filter --scaling strong"; do
  name=${cfg%%:*}; args=${cfg#*:}
  case "$args" in *--steps*) st="";; *) st="--steps 5 --warmup 1";; esac
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_$name --output-format csv -- python3 bench.py $st --no-cpu-baseline $args > $O/${name}_bench_under_rocprof.json 2> $O/prof_$name.err
  f=$(ls $O/prof_$name/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv && echo "$name: $(grep -c . $f) kernels"
done
timeout -k 10 500 python3 scratch/bigtable_probe.py > $O/bigtable_probe.txt 2>&1; echo "probe rc=$?"
for f in $O/bench_*.json; do python3 -c "import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], 'step', d['ms_per_step'], 'pass', r['avg_launch_ms'], 'frac', r['frac'])"; done
