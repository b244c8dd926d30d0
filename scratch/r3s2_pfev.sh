#!/bin/bash
set -o pipefail
O=gpurun_out/r3s2_ev; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 bench.py --config parent_filter --scaling strong --steps 10 --warmup 2 > $O/bench_parent_filter.json 2> $O/bench_parent_filter.err; echo "pf rc=$?"
rm -rf $O/prof_parent_filter
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_parent_filter --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --config parent_filter --scaling strong > $O/parent_filter_bench_under_rocprof.json 2> $O/prof_parent_filter.err
f=$(ls $O/prof_parent_filter/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/parent_filter_kernel_stats.csv
bash scratch/r3s2_pfc.sh > /dev/null 2>&1; cat gpurun_out/r03b_parent_filter_counters.txt
python3 -c "import json; d=json.load(open('$O/bench_parent_filter.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['cpu_baseline']['equals_gpu_result'])"
timeout -k 10 300 python3 scratch/sieve_small.py > $O/sieve_small.txt 2>&1
