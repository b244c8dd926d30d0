#!/bin/bash
# usage: scratch/pmc.sh <tag> "<counters>"   -- one PMC pass of the bench (2 steps), per-kernel averages
tag=$1; ctrs=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctrs -d gpurun_out/pmc_$tag --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$tag/bench.json 2> gpurun_out/pmc_$tag/err.log
python3 - <<EOF2
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$tag/*/*counter_collection.csv")
if not f: print("no counter file", glob.glob("gpurun_out/pmc_$tag/*/*")); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
seen=set()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"][:40]
    if not ("kb_" in k or "kdf_" in k): continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    key=(k,r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k]+=1
for k in acc:
    print(k, "dispatches", n[k])
    for c,v in sorted(acc[k].items()): print("   ", c.ljust(28), f"{v/n[k]:.4g}")
EOF2
