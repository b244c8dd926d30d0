#!/bin/bash
# round-3 evidence: bench lines, rocprofv3 kernel stats, SQ counters and TCC traffic for the bench configurations.
# Everything is written under gpurun_out/prof_r03/export/ (gpurun brings gpurun_out/ back); copy it into profiles/ afterwards:
#   cp gpurun_out/prof_r03/export/* profiles/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ulimit -c 0
out=gpurun_out/prof_r03; ex=$out/export; rm -rf $out; mkdir -p $ex
stats() {   # name, bench args...
  n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/$n --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $ex/r03_${n}_bench_under_rocprof.json 2> $out/$n.err
  f=$(ls $out/$n/*/*kernel_stats.csv | head -1); cp $f $ex/r03_${n}_kernel_stats.csv
  echo "== $n"; python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if n.startswith("void k") or n.startswith("k"):
        print("  ", n.replace("void ", "")[:58].ljust(60), r["Calls"].rjust(4), round(float(r["AverageNs"]) / 1e6, 3))
PY
}
# TCC traffic first: bench.py reports profiles/traffic_latest.json of the SAME code
scripts/collect_traffic.sh r03 latest && scripts/collect_traffic.sh r03_k63 --k 63
cp profiles/traffic_r03.json profiles/traffic_r03_k63.json profiles/traffic_latest.json $ex/
stats count
stats k63 --k 63
stats strong8 --scaling strong --batches 8
stats parent_filter --config parent_filter
# the plain bench lines (un-profiled clocks)
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > $ex/r03_bench_default.json 2> $out/default.err
timeout -k 10 300 python3 bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $ex/r03_bench_k63.json 2> $out/k63.err
timeout -k 10 400 python3 bench.py --scaling strong --batches 8 --steps 3 --warmup 2 --no-cpu-baseline > $ex/r03_bench_strong_n1.json 2> $out/strong.err
timeout -k 10 400 python3 bench.py --config parent_filter --steps 10 --warmup 2 > $ex/r03_bench_parent_filter.json 2> $out/pf.err
timeout -k 10 500 python3 scratch/bigtable_probe.py 2>&1 | grep -v amdgpu.ids > $ex/r03_bigtable_probe.txt
bash scratch/sq_counters_r03.sh > /dev/null 2>&1; cp gpurun_out/r03_sq_counters.txt $ex/r03_sq_counters.txt
for f in $ex/r03_bench_default.json $ex/r03_bench_k63.json $ex/r03_bench_strong_n1.json $ex/r03_bench_parent_filter.json; do python3 -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print('$f'.split('/')[-1], d['value'], d['ms_per_step'], r['frac'], r.get('avg_launch_ms'), r.get('traffic'), r.get('stage_avg_ms'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('equals_gpu_result'))"; done
cat $ex/r03_bigtable_probe.txt
