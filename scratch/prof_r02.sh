#!/bin/bash
# round-2 evidence: bench lines, rocprofv3 kernel stats and TCC traffic for the bench configurations.
# Everything is written under gpurun_out/prof_r02/export/ (the only directory gpurun brings back); copy it into profiles/
# afterwards:  cp gpurun_out/prof_r02/export/* profiles/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r02; ex=$out/export; rm -rf $out; mkdir -p $ex
stats() {   # name, bench args...
  n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/$n --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $ex/r02_${n}_bench_under_rocprof.json 2> $out/$n.err
  f=$(ls $out/$n/*/*kernel_stats.csv | head -1); cp $f $ex/r02_${n}_kernel_stats.csv
  echo "== $n"; python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if n.startswith("void k") or n.startswith("void sk_"):
        print("  ", n[5:58].ljust(54), r["Calls"].rjust(4), round(float(r["AverageNs"]) / 1e6, 3))
PY
}
# TCC traffic first: bench.py reports profiles/traffic_latest.json of the SAME code
scripts/collect_traffic.sh r02 latest && scripts/collect_traffic.sh r02_k63 --k 63 && scripts/collect_traffic.sh r02_superkmer --path superkmer
cp profiles/traffic_r02.json profiles/traffic_r02_k63.json profiles/traffic_r02_superkmer.json profiles/traffic_latest.json $ex/
for t in r02 r02_k63 r02_superkmer; do for c in fetch write; do f=$(ls gpurun_out/traffic_$t/$c/*/*counter_collection.csv | head -1); python3 - $f $ex/${t}_pmc_${c}_size.csv <<'PY'
import csv, sys, collections
acc, n = collections.defaultdict(float), collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["kernel", "counter", "dispatches", "KiB_per_dispatch"])
for (k, c), v in sorted(acc.items()): w.writerow([k, c, n[k], round(v / n[k], 1)])
PY
done; done
python bench.py --steps 20 --warmup 2 > $ex/r02_bench_default.json 2> $out/default.err; tail -1 $out/default.err; cut -c1-400 $ex/r02_bench_default.json
python bench.py --config parent_filter --steps 10 --warmup 2 > $ex/r02_bench_parent_filter.json 2> $out/pf.err; cut -c1-300 $ex/r02_bench_parent_filter.json
python bench.py --k 63 --steps 10 --warmup 2 --no-cpu-baseline > $ex/r02_bench_k63.json 2> $out/k63.err; cut -c1-300 $ex/r02_bench_k63.json
python bench.py --path superkmer --steps 10 --warmup 2 --no-cpu-baseline > $ex/r02_bench_superkmer.json 2> $out/sk.err; cut -c1-300 $ex/r02_bench_superkmer.json
stats count
stats parent_filter --config parent_filter
stats k63 --k 63
stats superkmer --path superkmer
python benchmarks/parent_filter.py > $ex/r02_parent_filter_chain.json 2> $out/chain.err; cat $ex/r02_parent_filter_chain.json
python scratch/merge_probe.py 8 > $ex/r02_merge_probe.txt 2> $out/merge.err; tail -1 $ex/r02_merge_probe.txt
python scratch/sieve_small.py > $ex/r02_sieve_small.txt 2> $out/sieve.err; tail -3 $ex/r02_sieve_small.txt
python scratch/bench_e2e_bam.py > $ex/r02_e2e_bam.txt 2> $out/e2e.err; tail -2 $ex/r02_e2e_bam.txt
python bench.py --gpus 2 --rehearse-one-gpu --steps 2 --warmup 1 --no-cpu-baseline > $ex/r02_rehearse_2ranks_one_gpu.json 2> $out/reh.err; cut -c1-200 $ex/r02_rehearse_2ranks_one_gpu.json
ls $ex
