#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_superkmer.py -x -q -m gpu 2>&1 | tail -2
python scratch/sk_probe.py 10000000 268435456 0 | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_q; mkdir -p gpurun_out/pmc_q
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d gpurun_out/pmc_q --output-format csv -- python3 scratch/sk_probe.py 10000000 268435456 0 > /dev/null 2>gpurun_out/pmc_q/err.log
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_q/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not (k.startswith("sk_bucket") or k.startswith("sk_extract")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(k, {c: round(acc[k][c] / max(1, len(n[(k, c)])) / 1e6, 1) for c in sorted(acc[k])})
PY
