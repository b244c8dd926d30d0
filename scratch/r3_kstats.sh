#!/bin/bash
# usage: r3_kstats.sh <tag> [KDF_EXTRA_FLAGS...] -- rebuilds when flags are given, runs the bench and rocprofv3 kernel stats
set -o pipefail
tag=$1; shift
out=gpurun_out/r3k_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$*" ]; then KDF_EXTRA_FLAGS="$*" python3 -m kmer_denovo_filter_amd.build --force > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }; fi
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 -c "import json,sys; d=json.load(open('$out/bench.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_avg_ms'], d['config']['kmers_ge3'])"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/prof.json 2> $out/prof.err
f=$(ls $out/prof/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if n.startswith("void k") or n.startswith("k"):
        print("  ", n.replace("void ","")[:60].ljust(62), r["Calls"].rjust(4), round(float(r["AverageNs"]) / 1e6, 4))
PY
