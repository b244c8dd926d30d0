#!/bin/bash
# round-2 evidence: bench lines, rocprofv3 kernel stats and TCC traffic for the bench configurations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r02; mkdir -p $out profiles
stats() {   # name, bench args...
  n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/$n --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $out/${n}_bench_under_rocprof.json 2> $out/$n.err
  f=$(ls $out/$n/*/*kernel_stats.csv | head -1); cp $f profiles/r02_${n}_kernel_stats.csv; cp $out/${n}_bench_under_rocprof.json profiles/r02_${n}_bench_under_rocprof.json
  echo "== $n"; python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if n.startswith("void k") or n.startswith("void sk_"):
        print("  ", n[5:58].ljust(54), r["Calls"].rjust(4), round(float(r["AverageNs"]) / 1e6, 3))
PY
}
python bench.py --steps 20 --warmup 2 > profiles/r02_bench_default.json 2> $out/default.err; tail -1 $out/default.err; cut -c1-400 profiles/r02_bench_default.json
python bench.py --config parent_filter --steps 10 --warmup 2 > profiles/r02_bench_parent_filter.json 2> $out/pf.err; cut -c1-300 profiles/r02_bench_parent_filter.json
python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > profiles/r02_bench_k63.json 2> $out/k63.err; cut -c1-300 profiles/r02_bench_k63.json
python bench.py --path superkmer --steps 10 --warmup 2 --no-cpu-baseline > profiles/r02_bench_superkmer.json 2> $out/sk.err; cut -c1-300 profiles/r02_bench_superkmer.json
stats count
stats parent_filter --config parent_filter
stats k63 --k 63
stats superkmer --path superkmer
scripts/collect_traffic.sh r02 latest
scripts/collect_traffic.sh r02_k63 --k 63
scripts/collect_traffic.sh r02_superkmer --path superkmer
python benchmarks/parent_filter.py > profiles/r02_parent_filter_chain.json 2> $out/chain.err; cat profiles/r02_parent_filter_chain.json
