#!/bin/bash
# SQ counters of the bench's kernels (run on the GPU box).  A few counters per pass; --kernel-trace only.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY"; do
  i=$((i+1)); out=gpurun_out/pmc_c/p$i; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $out --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/err.log || { echo "set $i failed"; tail -3 $out/err.log; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_c/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not (k.startswith("kb_") or k.startswith("sk_")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print(f"   {c:24s} {acc[k][c] / max(1, len(n[(k, c)])):16.0f} per dispatch")
PY
