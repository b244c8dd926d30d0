#!/bin/bash
# usage: scratch/prof.sh <tag> [bench args]   -- runs bench under rocprofv3 and prints the kernel table
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof_$tag/bench.json 2> gpurun_out/prof_$tag/err.log
python3 - <<EOF2
import csv,glob,json
f=glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "kb_" in n or "kdf_" in n or "fillBuffer" in n:
        print(n[:44].ljust(46), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e6,3), round(float(r["TotalDurationNs"])/1e6,2))
d=json.load(open("gpurun_out/prof_$tag/bench.json")); print("Gk-mer/s", d["value"], "ms/step", d["ms_per_step"], "pass ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"])
EOF2
