#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in 0 64 192 448 960; do
  rm -rf gpurun_out/pmc_a$f; mkdir -p gpurun_out/pmc_a$f
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d gpurun_out/pmc_a$f --output-format csv -- python3 scratch/sk_probe.py 10000000 268435456 $f > /dev/null 2>gpurun_out/pmc_a$f/err.log
  python3 - $f <<'PY'
import csv, glob, collections, sys
f = sys.argv[1]
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for fn in glob.glob(f"gpurun_out/pmc_a{f}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if not r["Kernel_Name"].startswith("void sk_bucket_kernel<0>"): continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print("flags", f, {c: round(acc[c] / max(1, len(n[c])) / 1e6, 1) for c in sorted(acc)})
PY
done
