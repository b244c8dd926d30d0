#!/bin/bash
# round-2 profiles: bench lines, rocprofv3 kernel stats for both bench configs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02a}
out=gpurun_out/prof_$tag; mkdir -p $out
python bench.py --steps 20 --warmup 2 > $out/bench_default.json 2> $out/bench_default.err; tail -2 $out/bench_default.err; cat $out/bench_default.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/count --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/count.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/pf --output-format csv -- python3 bench.py --config parent_filter --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_pf_under_rocprof.json 2> $out/pf.err
for d in count pf; do f=$(ls $out/$d/*/*kernel_stats.csv | head -1); echo "== $d"; head -12 $f | cut -d, -f1-8; cp $f $out/${d}_kernel_stats.csv; done
