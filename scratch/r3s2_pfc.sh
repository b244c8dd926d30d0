#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pf_ctr; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set -d $out/s$i --output-format csv -- python3 bench.py --config parent_filter --scaling strong --steps 2 --warmup 1 --no-cpu-baseline > $out/s$i.json 2> $out/s$i.err
done
python3 - $out <<'PY' > gpurun_out/r03b_parent_filter_counters.txt
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for fn in glob.glob(f"{out}/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("kdf_sieve_count"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("== counters of kdf_sieve_count_kernel (parent filter: 1 489 093 160 windows per dispatch), per dispatch, in millions")
for k in acc:
    d = {c: acc[k][c] / max(1, len(n[(k, c)])) for c in acc[k]}
    print(f"  {k[:50]}")
    for c in sorted(d): print(f"    {c:28s} {d[c] / 1e6:12.1f}")
PY
cat gpurun_out/r03b_parent_filter_counters.txt
